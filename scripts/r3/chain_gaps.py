#!/usr/bin/env python
"""Idle gaps of the GPU inside one map update of the shipped chain (rocprofv3 --kernel-trace CSV of scripts/r2_chain_bench.py): the
last update = the kernels between the last two registrations' init kernels; prints busy / idle time and the largest gaps with the kernels
on either side (a gap is a host read-back, an allocation or launch latency).  usage: chain_gaps.py <dir>"""
import csv, glob, os, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]))
rows.sort()
marks = [i for i, r in enumerate(rows) if "qfirst_kernel" in r[2]]       # head of a registration
if len(marks) < 3: sys.exit("not enough registrations")
# update k sits between the end of registration k (its last solve) and the head of registration k + 1
a, b = marks[-2], marks[-1]
seg = rows[a:b]
last_solve = max(i for i, r in enumerate(seg) if "solve_kernel" in r[2])
upd = seg[last_solve + 1:]
busy = sum(e - s for s, e, _ in upd)
span = upd[-1][1] - upd[0][0]
print(f"update: {len(upd)} kernels, span {span/1e3:.1f} us, busy {busy/1e3:.1f} us, idle {(span-busy)/1e3:.1f} us")
gaps = [(upd[i + 1][0] - upd[i][1], upd[i][2], upd[i + 1][2]) for i in range(len(upd) - 1)]
small = sum(g for g, _, _ in gaps if g < 8000)
print(f"gaps < 8 us: {sum(1 for g,_,_ in gaps if g < 8000)} totalling {small/1e3:.1f} us; larger:")
for g, x, y in sorted(gaps, reverse=True)[:18]:
    if g >= 8000: print(f"  {g/1e3:7.1f} us  after {x[:40]:40s} before {y[:40]}")
