#!/bin/bash
# config 4 replay: HIP runtime API totals (which host calls the 3 passes x 14 scans spend their time in)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6_c4api; rm -rf $O; mkdir -p $O
timeout 900 rocprofv3 --hip-runtime-trace --stats --output-format csv -d $O/trace -o t -- python $R/scripts/r5/config4_scans.py > $O/run.txt 2>&1
f=$(find $O/trace -name "*hip_api_stats.csv" | head -1)
python - "$f" <<'PY' | tee $O/hip_api_stats.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:22]:
    print(f"{r['Name']:36s} calls={int(r['Calls']):6d} total={float(r['TotalDurationNs'])/1e6:9.2f}ms avg={float(r['AverageNs'])/1e3:9.1f}us max={float(r['MaxNs'])/1e3:9.1f}us pct={float(r['Percentage']):5.1f}")
PY
find $O -name "*.csv" -delete; rm -rf $O/trace
tail -3 $O/run.txt
