#!/bin/bash
# round-2 GPU check 3: config 3/4/5 tests, NN lanes-per-query in the throughput-bound (batched) regime
set -x
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2c3; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_configs.py -x -q -s > $O/pytest_configs.log 2>&1; echo "rc=$?" >> $O/pytest_configs.log; tail -30 $O/pytest_configs.log
for g in 8 4; do for b in 1 8; do
  ICPMI_NN_G=$g timeout 300 python bench.py --no-cpu --no-extras --batch $b 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('G', $g, 'batch', $b, round(d['value']), d['step_ms']['median'])"
  ICPMI_NN_G=$g timeout 300 python bench.py --no-cpu --no-extras --batch $b --chain p2plane 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('G', $g, 'batch', $b, 'p2plane', round(d['value']), d['step_ms']['median'])"
done; done | tee $O/g_sweep.txt
