#!/usr/bin/env python
"""Per-dispatch durations (us) of the NN-side kernels of the last registration from a rocprofv3 --kernel-trace csv directory."""
import csv, glob, sys
d = sys.argv[1]
f = sorted(glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True))[-1]
rows = [r for r in csv.DictReader(open(f)) if "nn1_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tag = lambda r: "S" if "survive" in r["Kernel_Name"] else ("L" if "ELb1ELb1" in r["Kernel_Name"] or "true, true" in r["Kernel_Name"] else "N")
out = [(tag(r), round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, 1)) for r in rows]
# last registration = last 20 iterations: count back 20 non-S kernels
cnt = 0; i = len(out)
while i > 0 and cnt < 20:
    i -= 1
    if out[i][0] != "S": cnt += 1
print(" ".join(f"{t}{v}" for t, v in out[i:]))
