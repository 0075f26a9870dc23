#!/bin/bash
# nnk_hard_kernel bounded by the k-th key of the pass that queued the query (the tree) against the unbounded scan (scripts/r5/libicpmi_hu1.bin: the previous loop): BASELINE config 4, one call
cd "$GRAFT_REPO_ROOT"
cp norlab_icp_mapper_amd/libicpmi.so /tmp/prod.so
for rep in 1 2 3; do
  cp scripts/r5/libicpmi_hu1.bin norlab_icp_mapper_amd/libicpmi.so; echo "unbounded | $(python scripts/r5/config4.py 2>/dev/null | tail -1)"
  cp /tmp/prod.so norlab_icp_mapper_amd/libicpmi.so;              echo "bounded   | $(python scripts/r5/config4.py 2>/dev/null | tail -1)"
done
cp /tmp/prod.so norlab_icp_mapper_amd/libicpmi.so
bash scripts/r5/config4_kern.sh 2>&1 | grep "nnk_hard\|nnk_wave" | tail -6
