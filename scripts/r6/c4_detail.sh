#!/bin/bash
# config 4: per-scan register / update clocks of the last pass; one replay with ICPMI_CHAIN_TIMING=1 (per-op wall time, synchronised: perturbs overlap)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6_c4detail; mkdir -p $O
timeout 300 python scripts/r5/config4_scans.py 2>/dev/null | tail -16 | tee $O/scans.txt
ICPMI_CHAIN_TIMING=1 timeout 300 python scripts/r5/config4_scans.py 2>&1 | grep 'icpmi chain' | tail -14 | tee $O/chain_timing.txt
