#!/bin/bash
# speculative first NN launch: parity, then A/B against ICPMI_NN_SPEC=0 in one call, then the launch series
cd "$GRAFT_REPO_ROOT"; R=$GRAFT_REPO_ROOT; O=gpurun_out/${1:-spec}; mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_gpu_ext_filters.py tests/test_gpu_configs.py tests/test_gpu_batch.py tests/test_gpu_fused_solve.py tests/test_gpu_golden.py tests/test_gpu_planar.py -m gpu -q -x 2>&1 | grep -a "passed\|failed\|^FAILED\|^E  " | head -20
ICPMI_FUZZ_N=300 python -m pytest tests/test_gpu_fuzz.py -q -k random_chain 2>&1 | grep -a "passed\|failed\|^FAILED\|^E  " | head -20
for rep in 1 2; do for v in 1 0; do for chain in p2p p2plane; do
  ICPMI_NN_SPEC=$v python $R/bench.py --no-cpu --no-extras --chain $chain 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('spec=$v $chain', round(d['value']), d['ms_per_step'], d['roofline']['avg_launch_us'])"
done; done; done
for v in 1 0; do ICPMI_NN_SPEC=$v python scripts/r3/checked_loop_bench.py 2>/dev/null | grep "ms per" | sed "s/^/spec=$v /"; done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o t -- python $R/bench.py --no-cpu --no-extras --chain p2p > /dev/null 2>&1
cd $R; python scripts/r3/ktrace_series.py $O/prof 20 nn1_ | head -3
find $R/$O -name "*kernel_trace.csv" -delete; find $R/$O -name "*agent_info.csv" -delete
