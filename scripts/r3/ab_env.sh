#!/bin/bash
# A/B of kernel variants selected by environment variables, inside ONE gpurun call (box-to-box variance is large).
# usage: ab_env.sh "<chains>" "VAR=val VAR2=val" "VAR=val" ...   (each quoted group is one configuration)
cd "$GRAFT_REPO_ROOT"
chains="$1"; shift
for rep in $(seq 1 ${REPS:-2}); do
  for cfg in "$@"; do
    for ch in $chains; do
      echo "$cfg | $ch | $(env $cfg timeout 300 python bench.py --no-extras --no-cpu --chain $ch 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), 'it/s  nn', round(d['roofline']['avg_launch_us'],2), 'us  step', round(d['step_ms']['median'],4), 'ms  err_gt', d['pose_err_vs_ground_truth']['m'])")"
    done
  done
done
