#!/bin/bash
# A/B of library variants over scan sizes around the residency boundary of nn1_wg_kernel (1 536 workgroups at 6 per CU, 1 792 at 7)
cd "$GRAFT_REPO_ROOT"; cp norlab_icp_mapper_amd/libicpmi.so /tmp/prod.so
chain=$1; shift
for tag in "$@"; do
  if [ "$tag" = prod ]; then cp /tmp/prod.so norlab_icp_mapper_amd/libicpmi.so; else cp scripts/r6/libicpmi_$tag.bin norlab_icp_mapper_amd/libicpmi.so; fi
  for n in 65536 98304 100000 114688 131072; do
    echo "$tag | $chain | $n | $(python bench.py --no-extras --no-cpu --chain $chain --scan-points $n 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), 'it/s nn', round(d['roofline']['avg_launch_us'],2), 'us')")"
  done
done
cp /tmp/prod.so norlab_icp_mapper_amd/libicpmi.so
