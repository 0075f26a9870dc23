#!/bin/bash
# per-kernel totals (rocprofv3 --kernel-trace --stats) of: config 4 replay; the octree round-trip chain
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6_kstats; rm -rf $O; mkdir -p $O
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python $R/scripts/r5/config4_scans.py > $O/run_c4.txt 2>&1
f=$(find $O/trace -name "*kernel_stats.csv" | head -1); python $R/scripts/kstats.py $f 2>/dev/null | head -${1:-26} | tee $O/config4_kernel_stats.txt
find $O -name "*.csv" -delete; rm -rf $O/trace
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python $R/scripts/r2_chain_bench.py 1000000 100000 12 octree > $O/run_chain.txt 2>&1
f=$(find $O/trace -name "*kernel_stats.csv" | head -1); python $R/scripts/kstats.py $f 2>/dev/null | head -${1:-26} | tee $O/chain_kernel_stats.txt
find $O -name "*.csv" -delete; rm -rf $O/trace
