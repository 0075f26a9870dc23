#!/bin/bash
# A/B of library variants on the map-update chain bench (scripts/r2_chain_bench.py), in ONE gpurun call.  usage: ab_chain.sh "<tag> ENV=val ..." ...
cd "$GRAFT_REPO_ROOT"
cp norlab_icp_mapper_amd/libicpmi.so /tmp/prod.so
for rep in $(seq 1 ${REPS:-2}); do
  for cfg in "$@"; do
    tag=${cfg%% *}; envs=${cfg#* }; [ "$envs" = "$cfg" ] && envs=""
    if [ "$tag" = prod ]; then cp /tmp/prod.so norlab_icp_mapper_amd/libicpmi.so; else cp scripts/r5/libicpmi_$tag.bin norlab_icp_mapper_amd/libicpmi.so; fi
    echo "$cfg | $(env $envs timeout 300 python scripts/r2_chain_bench.py 1000000 100000 12 "${CHAIN:-octree, sensor}" 2>&1 | grep update | tail -1)"
  done
done
cp /tmp/prod.so norlab_icp_mapper_amd/libicpmi.so
