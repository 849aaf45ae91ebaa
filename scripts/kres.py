#!/usr/bin/env python3
"""Per-kernel register / LDS / occupancy table of one csrc/*.hip file (hipcc -Rpass-analysis=kernel-resource-usage, gfx950).
    python scripts/kres.py garment4d_amd/csrc/mlp_chain.hip [extra hipcc flags]"""
import os
import re
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-munsafe-fp-atomics",
       "-Rpass-analysis=kernel-resource-usage"] + sys.argv[2:] + ["-c", sys.argv[1], "-o", "/dev/null"]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], {}
for line in err.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        if cur:
            rows.append(cur)
        cur = {"name": m.group(1)}
        continue
    for key in ("TotalSGPRs", "VGPRs", "AGPRs", "ScratchSize", "Occupancy", "LDS Size"):
        mm = re.search(r"remark:\s+" + key + r"[^:]*: (\d+)", line)
        if mm and key not in cur:
            cur[key] = int(mm.group(1))
if cur:
    rows.append(cur)
if not rows:
    print(err[-3000:])
names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
for r, n in zip(rows, names):
    n = re.sub(r"\(.*", "", n).replace("void g4d::", "")
    v, a = r.get("VGPRs", 0), r.get("AGPRs", 0)
    print(f"{n:64s} v={v:3d} a={a:3d} tot={v + a:3d} s={r.get('TotalSGPRs', 0):3d} scratch={r.get('ScratchSize', 0)} lds={r.get('LDS Size', 0)} occ={r.get('Occupancy', 0)}")
