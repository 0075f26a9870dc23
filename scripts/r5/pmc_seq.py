#!/usr/bin/env python
"""Per-DISPATCH counter values in launch order (rocprofv3 --pmc ... --output-format csv): which launch of a registration fetches what.
    python scripts/r5/pmc_seq.py <dir> [first_dispatch_of_interest] [count]
Prints, per dispatch id: kernel, every collected counter; then per kernel the mean over the window."""
import collections, csv, glob, os, sys

d = sys.argv[1]
rows = collections.defaultdict(dict); names = {}
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        i = int(r["Dispatch_Id"])
        names[i] = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].strip()[:44]
        rows[i][r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(rows)
if not ids:
    print("no counter rows under", d); sys.exit(1)
tail = int(sys.argv[3]) if len(sys.argv) > 3 else 130
first = int(sys.argv[2]) if len(sys.argv) > 2 else max(ids[0], ids[-1] - tail)
ctrs = sorted({c for v in rows.values() for c in v})
print("dispatch kernel " + " ".join(ctrs))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for i in ids:
    if i < first or i >= first + tail:
        continue
    print(i, names[i], " ".join("%.0f" % rows[i].get(c, float("nan")) for c in ctrs))
    for c in ctrs:
        if c in rows[i]:
            agg[names[i]][c].append(rows[i][c])
print("--- means over the window")
for k, v in agg.items():
    print(k, {c: round(sum(x) / len(x), 1) for c, x in v.items()}, "n=%d" % len(next(iter(v.values()))))
