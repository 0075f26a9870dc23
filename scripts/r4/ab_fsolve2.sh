#!/bin/bash
cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_parity.py tests/test_gpu_ext_filters.py tests/test_gpu_golden.py tests/test_gpu_configs.py -m gpu -q -x 2>&1 | grep -a "passed\|failed\|Error\|assert" | tail -8
for rep in 1 2; do for fs in 0 1; do
  for chain in p2p p2plane; do
    ICPMI_FUSE_SOLVE=$fs python bench.py --no-cpu --no-extras --chain $chain 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fsolve=$fs $chain', round(d['value']), d['ms_per_step'], d['roofline']['avg_launch_us'], d['pose_err_vs_ground_truth']['m'])"
  done
  ICPMI_FUSE_SOLVE=$fs python scripts/r3/checked_loop_bench.py 2>/dev/null | grep "ms per" | sed "s/^/fsolve=$fs /"
done; done
