#!/usr/bin/env python
"""Per-launch durations from a rocprofv3 --kernel-trace CSV: the last `count` dispatches of every kernel whose name contains one of
the given substrings, in launch order (one registration of bench.py = 20 iterations: shows iteration 0 / 1 against the steady state).
usage: ktrace_series.py <dir or csv> <count> <substr> [<substr> ...]"""
import csv, glob, os, sys
path, count, subs = sys.argv[1], int(sys.argv[2]), sys.argv[3:]
files = [path] if path.endswith(".csv") else glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
for sub in subs:
    sel = [(s, e, k) for s, e, k in rows if sub in k]
    tail = sel[-count:]
    durs = [(e - s) / 1e3 for s, e, k in tail]
    if not durs:
        print(sub, ": no dispatches"); continue
    print(f"{sub}: {len(sel)} dispatches; last {len(durs)} (us): " + " ".join(f"{d:.1f}" for d in durs) + f" | mean {sum(durs)/len(durs):.2f} steady(4..) {sum(durs[4:])/max(len(durs[4:]),1):.2f}")
# whole last iteration windows: gap between consecutive solve kernels
sol = [(s, e) for s, e, k in rows if "solve_kernel" in k][-count:]
if len(sol) > 2:
    gaps = [(sol[i + 1][1] - sol[i][1]) / 1e3 for i in range(len(sol) - 1)]
    print("iteration period (solve end -> solve end, us): " + " ".join(f"{g:.1f}" for g in gaps))
