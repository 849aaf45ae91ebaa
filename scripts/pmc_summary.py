"""Summarise a rocprofv3 --pmc *_counter_collection.csv per kernel: mean counter value per dispatch.  python scripts/pmc_summary.py file.csv [substr]"""
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
sub = sys.argv[2] if len(sys.argv) > 2 else "g4d::"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    n = r["Kernel_Name"]
    if sub not in n:
        continue
    short = re.sub(r"\(.*", "", n.replace("void ", "").replace("g4d::", ""))[:60]
    acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k, " dispatches:", len(next(iter(cs.values()))))
    for c, v in sorted(cs.items()):
        print(f"    {c:32s} {sum(v) / len(v):16.1f}")
