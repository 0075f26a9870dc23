#!/bin/bash
# map-update chain + checked loop + config 4 with library variants (scripts/r4/libicpmi_<tag>.bin; "prod" = the tree's), alternating in ONE call
cd "$GRAFT_REPO_ROOT"
cp norlab_icp_mapper_amd/libicpmi.so /tmp/prod.so
for rep in 1 2 3; do
  for tag in "$@"; do
    if [ "$tag" = prod ]; then cp /tmp/prod.so norlab_icp_mapper_amd/libicpmi.so; else cp scripts/r4/libicpmi_$tag.bin norlab_icp_mapper_amd/libicpmi.so; fi
    echo "== $tag (rep $rep)"
    python scripts/r2_chain_bench.py 1000000 100000 12 2>&1 | grep update
    python scripts/r5/config4.py 2>/dev/null | tail -1
  done
done
cp /tmp/prod.so norlab_icp_mapper_amd/libicpmi.so
