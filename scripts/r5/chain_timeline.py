#!/usr/bin/env python
"""Every kernel of ONE map update of the shipped chain in launch order (rocprofv3 --kernel-trace CSV of scripts/r2_chain_bench.py): start offset,
duration, the idle gap in front of it.  The last update = the kernels between the last registration's final solve and the next head.
usage: chain_timeline.py <dir> [which update from the end, default 1]"""
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
last_solve = max(i for i, r in enumerate(seg) if "solve_kernel" in r[2])
upd = seg[last_solve + 1:]
t0 = upd[0][0]
busy = 0; prev_end = t0
for s, e, k in upd:
    print(f"{(s - t0) / 1e3:9.1f} us  +{(s - prev_end) / 1e3:6.1f} gap  {(e - s) / 1e3:7.1f} us  {k[:60]}")
    busy += e - s; prev_end = e
span = upd[-1][1] - t0
print(f"update: {len(upd)} kernels, span {span / 1e3:.1f} us, busy {busy / 1e3:.1f} us, idle {(span - busy) / 1e3:.1f} us")
