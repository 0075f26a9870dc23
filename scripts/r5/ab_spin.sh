#!/bin/bash
# counts through a tagged host-mapped word the host spins on (default) against a drained stream (ICPMI_SPIN_COUNTS=0): map-update chain + config 4, one call
cd "$GRAFT_REPO_ROOT"
for rep in 1 2 3; do
  for v in 0 1; do
    echo "== ICPMI_SPIN_COUNTS=$v (rep $rep)"
    ICPMI_SPIN_COUNTS=$v python scripts/r2_chain_bench.py 1000000 100000 12 2>&1 | grep update
    ICPMI_SPIN_COUNTS=$v python scripts/r5/config4.py 2>/dev/null | tail -1
  done
done
