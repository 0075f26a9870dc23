#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do for f in 2 1 0; do
  echo "ICPMI_NNK_WG_FROM=$f | $(ICPMI_NNK_WG_FROM=$f timeout 300 python bench.py --no-extras --no-cpu --chain docs_knn6 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), 'it/s  nn', round(d['roofline']['avg_launch_us'],2), 'us  step', round(d['step_ms']['median'],4), 'ms  err_gt', d['pose_err_vs_ground_truth']['m'])")"
done; done
