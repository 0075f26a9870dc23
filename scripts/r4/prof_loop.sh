#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r4loop; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for r in 1 8; do
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$r -o t -- python $R/scripts/r4/loopback_bench.py $r 8 2>/dev/null | grep epoch
python - <<PY
import csv, glob
f = glob.glob("$R/$O/prof_$r/**/*kernel_trace.csv", recursive=True)[0]
import re
def short(n):
    n = n.replace("(anonymous namespace)::", "")
    m = re.search(r"([A-Za-z0-9_]+)(<[^(]*>)?\(", n)
    return (m.group(1) if m else n)[:48]
rows = sorted(((int(x["Start_Timestamp"]), int(x["End_Timestamp"]), short(x["Kernel_Name"])) for x in csv.DictReader(open(f))))
# the last epoch: from the last transform_kernel... find last 'keep_flag_kernel' and print kernels from a bit before to the end
idx = max(i for i, x in enumerate(rows) if "keep_flag" in x[2])
start = idx - 12
t0 = rows[start][0]
busy = 0
for s, e, k in rows[start:]:
    busy += e - s
print("R=$r last epoch: kernels", len(rows) - start, "busy us", busy / 1e3, "span us", (rows[-1][1] - t0) / 1e3)
import collections
agg = collections.OrderedDict()
for s, e, k in rows[start:]:
    a = agg.setdefault(k, [0, 0]); a[0] += 1; a[1] += (e - s) / 1e3
for k, (n, t) in agg.items(): print(f"   {k:50s} x{n:3d} {t:8.1f} us")
PY
done
find $R/$O -name "*kernel_trace.csv" -delete; find $R/$O -name "*agent_info.csv" -delete
