#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r2c6; mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed" | tail -2
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_chain -- python $GRAFT_REPO_ROOT/scripts/r2_chain_bench.py 1000000 100000 12 "octree, sensor" > /dev/null 2>&1
python $GRAFT_REPO_ROOT/scripts/kstats.py $GRAFT_REPO_ROOT/$O/prof_chain | head -16
