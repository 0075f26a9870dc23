#!/usr/bin/env python
"""Every kernel between the heads of the last two registrations (rocprofv3 --kernel-trace CSV of scripts/r2_chain_bench.py), in launch order:
start offset, idle gap in front, duration.  usage: chain_timeline_all.py <dir> [which from the end, default 1]"""
import csv, glob, os, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]))
rows.sort()
marks = [i for i, r in enumerate(rows) if "qfirst_kernel" in r[2]]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
a, b = marks[-1 - back], marks[-back]
seg = rows[a:b]
t0 = seg[0][0]; prev_end = t0; busy = 0
for s, e, k in seg:
    print(f"{(s - t0) / 1e3:9.1f} us  +{(s - prev_end) / 1e3:6.1f} gap  {(e - s) / 1e3:7.1f} us  {k[:70]}")
    busy += e - s; prev_end = max(prev_end, e)
print(f"segment: {len(seg)} kernels, span {(seg[-1][1] - t0) / 1e3:.1f} us, busy {busy / 1e3:.1f} us")
