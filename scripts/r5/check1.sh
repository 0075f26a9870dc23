#!/bin/bash
# r5 first combined check: L2 micro-benchmark (fixed clocks), the GPU tests touched so far, nt streams A/B, fused solve A/B (query loads hoisted above
# the solve), BASELINE config 4 on the clock.  Output: gpurun_out/r5c1/
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5c1; mkdir -p $O
timeout 120 scripts/r5/l2_survive.bin > $O/l2_survive.txt 2>&1; head -12 $O/l2_survive.txt
timeout 1500 python -m pytest tests/test_pins.py tests/test_gpu_insert.py tests/test_gpu_parity.py tests/test_gpu_map_chain.py -m gpu -q -x 2>&1 | grep -a "passed\|failed\|Error\|assert" | tail -8 | tee $O/tests.txt
REPS=2 bash scripts/r5/ab_lib.sh "p2p p2plane" prod nt 2>&1 | tee $O/ab_nt.txt
for rep in 1 2; do for fs in 0 1; do
  for chain in p2p p2plane; do
    ICPMI_FUSE_SOLVE=$fs python bench.py --no-cpu --no-extras --chain $chain 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fsolve=$fs $chain', round(d['value']), d['ms_per_step'], d['roofline']['avg_launch_us'], d['pose_err_vs_ground_truth']['m'])"
  done
  ICPMI_FUSE_SOLVE=$fs python scripts/r3/checked_loop_bench.py 2>/dev/null | grep "ms per" | sed "s/^/fsolve=$fs /"
done; done 2>&1 | tee $O/ab_fsolve.txt
python - <<'PY' 2>&1 | tee $O/config4.txt
import json, sys, numpy as np
sys.path.insert(0, ".")
import norlab_icp_mapper_amd as pkg
import bench
print(json.dumps(bench.config4_replay(np, pkg, True), indent=1))
PY
