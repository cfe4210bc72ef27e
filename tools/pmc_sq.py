"""Print per-kernel averages of the SQ counters in a rocprofv3 --pmc counter_collection CSV.  usage: python tools/pmc_sq.py <csv> <kernel substring>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
want = sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if want in r["Kernel_Name"]:
        agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:32s} {sum(v) / len(v):16.0f}  (n={len(v)})")
