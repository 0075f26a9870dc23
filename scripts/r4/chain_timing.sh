#!/bin/bash
cd "$GRAFT_REPO_ROOT"
ICPMI_CHAIN_TIMING=1 python scripts/r2_chain_bench.py 1000000 100000 6 "octree, sensor" 2>&1 | tail -16
ICPMI_CHAIN_TIMING=1 python scripts/r2_chain_bench.py 1000000 100000 6 "point_distance" 2>&1 | tail -12
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=gpurun_out/r4chain; mkdir -p $R/$O
timeout 600 rocprofv3 --hip-trace --stats --output-format csv -d $R/$O/api -o t -- python $R/scripts/r2_chain_bench.py 1000000 100000 12 "octree, sensor" > /dev/null 2>&1
f=$(find $R/$O/api -name "*hip_api_stats.csv" | head -1); head -14 $f | cut -c1-120
find $R/$O -name "*trace.csv" -delete
