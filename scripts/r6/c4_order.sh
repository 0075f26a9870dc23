#!/bin/bash
# the kernels of ONE map update of the config-4 replay in launch order (rocprofv3 kernel trace), with their durations and the gaps between them
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6_c4order; rm -rf $O; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python $R/scripts/r5/config4_scans.py > /dev/null 2>&1
f=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY' | tee $O/order.txt
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    m = re.search(r"([A-Za-z0-9_]+(<[^>(]*>)?)\(", n); return m.group(1) if m else n[:40]
names = [short(r["Kernel_Name"]) for r in rows]
# the last update: from the last 'dyn_beams_kernel' back to ... print the kernels between the last solve_kernel of the previous registration and the next qfirst
idx = [i for i, n in enumerate(names) if n.startswith("qfirst_kernel")]
a, b = idx[-2], idx[-1]
# registration = a .. last solve before update; print everything between the last solve_kernel in [a, b) and b
last_solve = max(i for i in range(a, b) if names[i] == "solve_kernel")
prev_end = None
tot = 0.0
for i in range(last_solve + 1, b):
    s, e = int(rows[i]["Start_Timestamp"]), int(rows[i]["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print(f"{i - last_solve:3d} {names[i]:34s} {(e - s) / 1e3:8.2f} us   gap {gap:7.2f} us")
    tot += (e - s) / 1e3; prev_end = e
print("kernels", b - last_solve - 1, "busy us", round(tot, 1), "span us", round((int(rows[b - 1]["End_Timestamp"]) - int(rows[last_solve + 1]["Start_Timestamp"])) / 1e3, 1))
PY
rm -rf $O/trace
