#!/bin/bash
# r5: map-update chain after the first batch of changes (dyn_update row runs, matrices as kernel arguments, fused move, two-kernel scans)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/${1:-r5c2}; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_map_chain.py tests/test_gpu_octree.py tests/test_gpu_insert.py tests/test_gpu_recalled.py tests/test_gpu_fuzz.py tests/test_gpu_planar.py tests/test_gpu_merge_loopback.py -m gpu -q -x 2>&1 | grep -a "passed\|failed\|Error\|assert" | tail -8 | tee $O/tests.txt
python scripts/r2_chain_bench.py 1000000 100000 12 2>&1 | grep update | tee $O/chain_bench.txt
ICPMI_SCAN2=0 python scripts/r2_chain_bench.py 1000000 100000 12 "octree, sensor" 2>&1 | grep update | sed 's/^/SCAN2=0 /' | tee -a $O/chain_bench.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace -o t -- python $R/scripts/r2_chain_bench.py 1000000 100000 12 "octree, sensor" > /dev/null 2>&1
python $R/scripts/r5/chain_timeline.py $R/$O/trace > $R/$O/timeline.txt 2>&1; tail -1 $R/$O/timeline.txt
f=$(find $R/$O/trace -name "*kernel_stats.csv" | head -1); python $R/scripts/kstats.py $f 2>/dev/null | head -24 | tee $R/$O/kstats.txt
find $R/$O -name "*.csv" -delete
