#!/bin/bash
# usage: knn_phase.sh [ENV=val ...] -- scripts/r3/knn_phase.py against the timing build
cd "$GRAFT_REPO_ROOT"
cp norlab_icp_mapper_amd/libicpmi.so /tmp/prod.so
cp scripts/r3/libicpmi_timing.bin norlab_icp_mapper_amd/libicpmi.so
env "$@" python scripts/r3/knn_phase.py 100000 2>&1 | tail -5
cp /tmp/prod.so norlab_icp_mapper_amd/libicpmi.so
