#!/bin/bash
# HIP API time of the BASELINE config 4 replay (three passes of 14 scans): which runtime calls the first scans of a fresh mapper pay for
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c4api; mkdir -p $O
timeout 900 rocprofv3 --hip-runtime-trace --stats --output-format csv -d $O/trace -o t -- python $R/scripts/r5/config4_scans.py > $O/run.txt 2>&1
f=$(find $O/trace -name "*hip_api_stats.csv" | head -1)
head -25 "$f" | cut -c1-150 | tee $O/api_stats.txt
find $O -name "*.csv" -size +1M -delete
