#!/bin/bash
# one update of the shipped chain as a kernel timeline (launch order, durations, gaps), the step clocks of ICPMI_CHAIN_TIMING, and the per-update wall time
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/${1:-r5chain}; mkdir -p $O
python scripts/r2_chain_bench.py 1000000 100000 12 2>&1 | grep update | tee $O/chain_bench.txt
ICPMI_CHAIN_TIMING=1 python scripts/r2_chain_bench.py 1000000 100000 5 "octree, sensor" 2>&1 | tail -14 | tee $O/chain_steps.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace -o t -- python $R/scripts/r2_chain_bench.py 1000000 100000 6 "octree, sensor" > /dev/null 2>&1
python $R/scripts/r5/chain_timeline.py $R/$O/trace > $R/$O/timeline.txt 2>&1; tail -3 $R/$O/timeline.txt
find $R/$O -name "*.csv" -delete
