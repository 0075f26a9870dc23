#!/bin/bash
# per-kernel totals of the BASELINE config 4 replay (rocprofv3 --kernel-trace --stats): which kernels a real lidar map's update spends its time in
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c4kstats; rm -rf $O; mkdir -p $O
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python $R/scripts/r5/config4_scans.py > $O/run.txt 2>&1
f=$(find $O/trace -name "*kernel_stats.csv" | head -1); python $R/scripts/kstats.py $f 2>/dev/null | head -${1:-22}
find $O -name "*.csv" -delete
