#!/bin/bash
# the end of a registration: copy + drain (prev / ICPMI_FAST_FINISH=0) against the mirrored state (prod), headline + checked loop + config 4, one call
cd "$GRAFT_REPO_ROOT"
cp norlab_icp_mapper_amd/libicpmi.so /tmp/prod.so
run() { echo "== $1"; shift; env "$@" python bench.py --no-cpu --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('headline', round(d['value']), 'it/s  step', round(d['step_ms']['median'],4), 'ms')"
  env "$@" python scripts/r3/checked_loop_bench.py 2>/dev/null | grep "ms per"
  env "$@" python scripts/r5/config4.py 2>/dev/null | tail -1; }
for rep in 1 2 3; do
  cp scripts/r4/libicpmi_prev.bin norlab_icp_mapper_amd/libicpmi.so; run "prev (rep $rep)" A=1
  cp /tmp/prod.so norlab_icp_mapper_amd/libicpmi.so; run "prod (rep $rep)" A=1
  run "prod ICPMI_FAST_FINISH=0 (rep $rep)" ICPMI_FAST_FINISH=0
done
cp /tmp/prod.so norlab_icp_mapper_amd/libicpmi.so
