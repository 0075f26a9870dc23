#!/bin/bash
# the shipped chain at the size of the bundled data (BASELINE config 4): 37 k-point scans into a ~100 k-point map
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5chain_small; mkdir -p $O
python scripts/r2_chain_bench.py 200000 37000 12 "octree, sensor" 2>&1 | grep update | tee $O/chain_bench.txt
ICPMI_CHAIN_TIMING=1 python scripts/r2_chain_bench.py 200000 37000 5 "octree, sensor" 2>&1 | tail -13 | tee $O/chain_steps.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace -o t -- python $R/scripts/r2_chain_bench.py 200000 37000 12 "octree, sensor" > /dev/null 2>&1
python $R/scripts/r5/chain_timeline.py $R/$O/trace > $R/$O/timeline.txt 2>&1; tail -1 $R/$O/timeline.txt
f=$(find $R/$O/trace -name "*kernel_stats.csv" | head -1); python $R/scripts/kstats.py $f 2>/dev/null | head -30 | tee $R/$O/kstats.txt
find $R/$O -name "*.csv" -delete
