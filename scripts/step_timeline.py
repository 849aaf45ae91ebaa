"""Print the per-dispatch timeline of one bench step from a rocprofv3 kernel trace CSV."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'fps_reg_kernel<16' in r['Kernel_Name'] or 'fps_reg_kernel<8' in r['Kernel_Name']]
a, b = idx[10], idx[11]
t0 = int(rows[a]['Start_Timestamp']); tot = 0
agg = {}
for r in rows[a:b]:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3; tot += d
    name = r['Kernel_Name'].replace('void ', '').replace('g4d::', '').replace('at::native::', '')[:44]
    agg[name.split('(')[0]] = agg.get(name.split('(')[0], 0) + d
    if len(sys.argv) > 2:
        print(f"{(int(r['Start_Timestamp'])-t0)/1e3:8.1f} {d:7.1f}us g={r['Grid_Size_X']}x{r['Grid_Size_Y']}/{r['Workgroup_Size_X']} lds={r['LDS_Block_Size']} v={r['VGPR_Count']} {name}")
print('kernels', b - a, 'sum', round(tot, 1), 'span', (int(rows[b]['Start_Timestamp']) - t0) / 1e3)
for k, v in sorted(agg.items(), key=lambda kv: -kv[1]):
    print(f"  {v:8.1f} us  {k}")
