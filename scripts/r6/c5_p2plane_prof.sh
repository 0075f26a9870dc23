#!/bin/bash
# kernel table of the config-5 point-to-plane epoch (SurfaceNormal knn 10 over the grown 10 M-point map inside every epoch)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6_c5p; rm -rf $O; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python $R/bench.py --workload config5 --scans 6 --chain p2plane --epoch-normals-knn 10 > $O/run.txt 2>&1
f=$(find $O/trace -name "*kernel_stats.csv" | head -1); python $R/scripts/kstats.py $f 2>/dev/null | head -44 | tee $O/kernel_stats.txt
find $O -name "*.csv" -delete; rm -rf $O/trace
