#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6_diag; mkdir -p $O
ICPMI_SELF_DIAG=1 timeout 300 python scripts/r5/config4_scans.py 2>&1 | grep 'self-knn' | tail -4 | tee $O/diag_c4.txt
ICPMI_SELF_DIAG=1 timeout 300 python scripts/r2_chain_bench.py 1000000 100000 4 octree 2>&1 | grep 'self-knn' | tail -4 | tee $O/diag_chain.txt
