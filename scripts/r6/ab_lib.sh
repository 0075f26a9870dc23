#!/bin/bash
# A/B of library variants (scripts/r3/libicpmi_<tag>.bin; "prod" = the tree's library) x environment settings, in ONE gpurun call.
# usage: ab_lib.sh "<chains>" "<tag> ENV=val ..." ...
cd "$GRAFT_REPO_ROOT"
cp norlab_icp_mapper_amd/libicpmi.so /tmp/prod.so
chains="$1"; shift
for rep in $(seq 1 ${REPS:-2}); do
  for cfg in "$@"; do
    tag=${cfg%% *}; envs=${cfg#* }; [ "$envs" = "$cfg" ] && envs=""
    if [ "$tag" = prod ]; then cp /tmp/prod.so norlab_icp_mapper_amd/libicpmi.so; else cp scripts/r6/libicpmi_$tag.bin norlab_icp_mapper_amd/libicpmi.so; fi
    for ch in $chains; do
      echo "$cfg | $ch | $(env $envs timeout 300 python bench.py --no-extras --no-cpu --chain $ch 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), 'it/s  nn', round(d['roofline']['avg_launch_us'],2), 'us  step', round(d['step_ms']['median'],4), 'ms  err_gt', d['pose_err_vs_ground_truth']['m'])")"
    done
  done
done
cp /tmp/prod.so norlab_icp_mapper_amd/libicpmi.so
