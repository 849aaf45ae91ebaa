"""Condense a rocprofv3 *_kernel_stats.csv: g4d kernels only, short names, calls, avg us, total share.  python scripts/kstats.py file.csv [steps]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else None
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"{'kernel':70s} {'calls':>6s} {'avg_us':>9s} {'share%':>7s}" + ("  us/step" if steps else ""))
for r in rows:
    n = r["Name"]
    if "g4d::" not in n and "rocclr" not in n:
        continue
    short = re.sub(r"\(.*", "", n.replace("void ", "").replace("g4d::", ""))[:70]
    t = float(r["TotalDurationNs"])
    line = f"{short:70s} {int(r['Calls']):6d} {float(r['AverageNs']) / 1e3:9.1f} {100 * t / tot:7.2f}"
    if steps:
        line += f" {t / 1e3 / steps:8.1f}"
    print(line)
