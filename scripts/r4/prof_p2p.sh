#!/bin/bash
# kernel trace of the headline chain (p2p) and the point-to-plane chain: per-kernel averages
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/${1:-r4prof}; mkdir -p $O
if [ -n "$2" ]; then python -m pytest tests/test_gpu_parity.py tests/test_gpu_ext_filters.py tests/test_gpu_batch.py tests/test_gpu_golden.py tests/test_gpu_planar.py tests/test_gpu_knn_wg.py -m gpu -q -x 2>&1 | grep -a "passed\|failed\|Error\|assert" | tail -8; fi
python bench.py --no-cpu --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('p2p', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])"
python bench.py --no-cpu --no-extras --chain p2plane 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('p2plane', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for chain in p2p p2plane; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$chain -o t -- python $R/bench.py --no-cpu --no-extras --chain $chain > /dev/null 2>&1
  f=$(find $R/$O/prof_$chain -name "*kernel_stats.csv" | head -1); echo "== $chain"; python $R/scripts/kstats.py $f 2>/dev/null | head -9
done
find $R/$O -name "*kernel_trace.csv" -delete; find $R/$O -name "*agent_info.csv" -delete
