#!/bin/bash
# one update of the PointDistance + normals chain as a kernel timeline (launch order, durations, gaps) and the step clocks of ICPMI_CHAIN_TIMING
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/${1:-r5chainpd}; mkdir -p $O
ICPMI_CHAIN_TIMING=1 python scripts/r2_chain_bench.py 1000000 100000 5 "point_distance" 2>&1 | tail -16 | tee $O/chain_steps.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace -o t -- python $R/scripts/r2_chain_bench.py 1000000 100000 6 "point_distance" > /dev/null 2>&1
python $R/scripts/r5/chain_timeline_all.py $R/$O/trace 2 > $R/$O/timeline.txt 2>&1; cat $R/$O/timeline.txt
find $R/$O -name "*.csv" -delete
