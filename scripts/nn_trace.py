#!/usr/bin/env python
"""Per-dispatch durations of the NN kernels of the last registration from a rocprofv3 --kernel-trace csv directory."""
import csv, glob, sys
d = sys.argv[1]
f = sorted(glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True))[-1]
rows = [r for r in csv.DictReader(open(f)) if "nn1_ml_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = rows[-20:]
print([round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, 1) for r in last])
