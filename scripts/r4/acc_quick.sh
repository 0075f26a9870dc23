#!/bin/bash
cd "$GRAFT_REPO_ROOT"; R=$GRAFT_REPO_ROOT; O=gpurun_out/${1:-accq}; mkdir -p $O
python -m pytest tests/test_gpu_knn_wg.py tests/test_gpu_parity.py tests/test_gpu_fused_solve.py tests/test_gpu_batch.py tests/test_gpu_configs.py -m gpu -q -x 2>&1 | tail -2
for chain in p2p p2plane docs_knn6; do
  for rep in 1 2; do python $R/bench.py --no-cpu --no-extras --chain $chain 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$chain', round(d['value']), d['ms_per_step'])"; done
done
cd /tmp && export TMPDIR=/tmp
for chain in p2plane docs_knn6; do
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$chain -o t -- python $R/bench.py --no-cpu --no-extras --chain $chain > /dev/null 2>&1
f=$(find $R/$O/prof_$chain -name "*kernel_stats.csv" | head -1); python $R/scripts/kstats.py $f 2>/dev/null | head -6
done
find $R/$O -name "*kernel_trace.csv" -delete; find $R/$O -name "*agent_info.csv" -delete
