"""Aggregate a rocprofv3 --pmc counter_collection.csv per kernel: python tools/pmc_summary.py <csv> [filter]"""
import csv, sys, collections
rows = csv.DictReader(open(sys.argv[1]))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
for r in rows:
    k = r["Kernel_Name"].replace("void ", "")[:48]
    if flt and flt not in k:
        continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[k].add(r["Dispatch_Id"])
for k, c in agg.items():
    n = len(cnt[k])
    print(f"{k:48s} dispatches={n}")
    for name, v in sorted(c.items()):
        print(f"    {name:32s} {v/n:16.1f} per dispatch")
