#!/bin/bash
# the left-overs of the tiled self search on BASELINE config 4 (real lidar scans): how many reach the ring kernel / the brute pass, and what the
# brute pass's grid and the ring bound do to the replay
cd "$GRAFT_REPO_ROOT"
ICPMI_SELF_DIAG=1 python scripts/r5/config4.py 2>&1 | grep "self-knn" | tail -6
for rm in 6 10 16; do ICPMI_SELF_RING_MAX=$rm ICPMI_SELF_DIAG=1 python scripts/r5/config4.py 2>&1 | grep "self-knn" | tail -1; done
for rep in 1 2; do
 for cfg in "ICPMI_SELF_HARD_GRID=512" "ICPMI_SELF_HARD_GRID=2048" "ICPMI_SELF_HARD_GRID=8192" "ICPMI_SELF_HARD_GRID=2048 ICPMI_SELF_RING_MAX=10" "ICPMI_SELF_HARD_GRID=2048 ICPMI_SELF_RING_MAX=16" "ICPMI_SELF_HARD_GRID=512 ICPMI_SELF_RING_MAX=12"; do
  echo "$cfg | $(env $cfg python scripts/r5/config4.py 2>/dev/null | tail -1)"
 done
done
