#!/bin/bash
# the map-update chain alone: kernel stats under the tracer, then the clocks without it, then the parity tests of what the chain touches
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/${1:-chainq}; mkdir -p $O
R=$GRAFT_REPO_ROOT
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_chain -o t -- python $R/scripts/r2_chain_bench.py 1000000 100000 12 "octree, sensor" > /dev/null 2>&1 )
f=$(find $O/prof_chain -name "*kernel_stats.csv" | head -1); python scripts/kstats.py $f 2>/dev/null | head -24 | tee $O/chain_kstats.txt
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
python scripts/r2_chain_bench.py 1000000 100000 12 2>&1 | grep update | tee $O/chain_bench.txt
python scripts/r3/checked_loop_bench.py 2>/dev/null | grep "ms per" | tee $O/checked_loop.txt
python scripts/r5/config4.py 2>/dev/null | tail -1 | tee $O/config4.txt
python -m pytest tests/test_gpu_octree.py tests/test_gpu_map_chain.py tests/test_gpu_fuzz.py tests/test_gpu_runtime.py tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -3 | tee $O/tests.txt
