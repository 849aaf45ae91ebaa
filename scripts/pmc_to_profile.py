"""Merge rocprofv3 --pmc counter_collection CSVs (one pass per counter) into the per-kernel table bench.py reads from profiles/:
kernel,counter,launches,avg_per_launch_KB.   python scripts/pmc_to_profile.py out.csv pass1_counter_collection.csv [pass2 ...]
FETCH_SIZE / WRITE_SIZE are reported in KB by the counter definition on gfx950 (MI355X_MICROARCH.md, HBM section)."""
import collections
import csv
import sys

acc = collections.defaultdict(list)
for path in sys.argv[2:]:
    for r in csv.DictReader(open(path)):
        if "g4d::" in r["Kernel_Name"]:
            acc[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
with open(sys.argv[1], "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "counter", "launches", "avg_per_launch_KB"])
    for (k, c), v in sorted(acc.items(), key=lambda kv: (kv[0][1], -sum(kv[1]))):
        w.writerow([k, c, len(v), round(sum(v) / len(v), 1)])
