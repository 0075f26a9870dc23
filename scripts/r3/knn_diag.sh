#!/bin/bash
# usage: knn_diag.sh <lib tag> [ENV=val ...]
cd "$GRAFT_REPO_ROOT"
tag=$1; shift
cp norlab_icp_mapper_amd/libicpmi.so /tmp/prod.so
cp scripts/r3/libicpmi_$tag.bin norlab_icp_mapper_amd/libicpmi.so
echo "== $tag $@"; env "$@" python scripts/r3/knn_diag.py 2>&1 | tail -7
cp /tmp/prod.so norlab_icp_mapper_amd/libicpmi.so
