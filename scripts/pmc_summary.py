#!/usr/bin/env python
"""Average rocprofv3 --pmc counters per kernel (substring filter) from a counter_collection.csv dir."""
import csv, glob, collections, sys
d = sys.argv[1]; filt = sys.argv[2] if len(sys.argv) > 2 else ""
for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if filt not in k: continue
        k = k.split("(")[0].split("::")[-1][:40]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    for k, v in agg.items():
        print(k, {c: round(x / cnt[(k, c)]) for c, x in sorted(v.items())})
