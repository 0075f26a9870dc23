#!/bin/bash
# the end of a registration: copy + drain behind an event pair (ICPMI_FAST_FINISH=0) against the mirrored state (default), one call
cd "$GRAFT_REPO_ROOT"
run() { echo "== $1"; shift; env "$@" python bench.py --no-cpu --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('headline', round(d['value']), 'it/s  ms_per_step', round(d['ms_per_step'],4), ' device loop', round(d.get('device_loop_ms_per_step',0),4))"
  env "$@" python scripts/r3/checked_loop_bench.py 2>/dev/null | grep "ms per"
  env "$@" python bench.py --no-cpu --no-extras --chain p2plane 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('p2plane', round(d['value']), 'it/s')"
  env "$@" python bench.py --no-cpu --no-extras --chain docs_knn6 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('knn6', round(d['value']), 'it/s')"; }
for rep in 1 2 3; do
  run "FAST_FINISH=0 (rep $rep)" ICPMI_FAST_FINISH=0
  run "default (rep $rep)" A=1
done
