#!/bin/bash
# the coalesced piece phase of the first NN launches (nn1_wg_kernel<.., COAL>): ICPMI_NN_COAL_UNTIL = 0 (off) / 1 / 2 / 3, headline + checked loop + series of launch times
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
  for v in 0 1 2 3; do
    echo "== ICPMI_NN_COAL_UNTIL=$v (rep $rep)"
    ICPMI_NN_COAL_UNTIL=$v python bench.py --no-cpu --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('headline', round(d['value']), 'it/s  ms_per_step', round(d['ms_per_step'],4))"
    ICPMI_NN_COAL_UNTIL=$v python bench.py --no-cpu --no-extras --chain p2plane 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('p2plane', round(d['value']), 'it/s')"
    ICPMI_NN_COAL_UNTIL=$v python scripts/r3/checked_loop_bench.py 2>/dev/null | grep "ms per"
  done
done
