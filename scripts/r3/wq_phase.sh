#!/bin/bash
# usage: wq_phase.sh [ENV=val ...] -- runs scripts/r3/wq_phase.py against the timing build, for p2p and p2plane
cd "$GRAFT_REPO_ROOT"
cp norlab_icp_mapper_amd/libicpmi.so /tmp/prod.so
cp scripts/r3/libicpmi_timing.bin norlab_icp_mapper_amd/libicpmi.so
for m in 1 2; do echo "minimizer $m $@"; env "$@" python scripts/r3/wq_phase.py 100000 $m 2>&1 | tail -3; done
cp /tmp/prod.so norlab_icp_mapper_amd/libicpmi.so
