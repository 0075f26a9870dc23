#!/bin/bash
# map_build's statistics through the host-mapped page + a tagged spin (default) against the copy + drained stream (ICPMI_SPIN_STATS=0), one call
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do for R in 0 1; do
  echo "== ICPMI_SPIN_STATS=$R (rep $rep)"
  ICPMI_SPIN_STATS=$R python scripts/r2_chain_bench.py 1000000 100000 12 2>&1 | grep update
  echo "   config 4: $(ICPMI_SPIN_STATS=$R python scripts/r5/config4.py 2>/dev/null | tail -1)"
done; done
