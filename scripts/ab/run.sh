#!/bin/bash
# A/B of two builds of libicpmi.so inside one gpurun call: scripts/ab/libicpmi_old.bin against the tree's library
cd "$GRAFT_REPO_ROOT"
cp norlab_icp_mapper_amd/libicpmi.so /tmp/new.so
for rep in 1 2; do
  for which in old new; do
    if [ $which = old ]; then cp scripts/ab/libicpmi_old.bin norlab_icp_mapper_amd/libicpmi.so; else cp /tmp/new.so norlab_icp_mapper_amd/libicpmi.so; fi
    for ch in "$@"; do echo "$which $ch $(timeout 200 python bench.py --no-extras --no-cpu --chain $ch 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']))")"; done
  done
done
cp /tmp/new.so norlab_icp_mapper_amd/libicpmi.so
