#!/bin/bash
# r6: first run of the sparse block grid self search: parity, then config 4 and the synthetic chains, A/B against the r5 path (ICPMI_SELF_GRID=0) in one call
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6_first; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_pins.py tests/test_gpu_map_chain.py tests/test_gpu_planar.py tests/test_host_filters.py -m gpu -x -q 2>&1 | tail -15 > $O/tests.txt
cat $O/tests.txt
for g in 1 0 1 0; do
  echo "== ICPMI_SELF_GRID=$g config4" | tee -a $O/ab.txt
  ICPMI_SELF_GRID=$g timeout 300 python scripts/r5/config4.py 2>&1 | tail -2 | tee -a $O/ab.txt
done
ICPMI_SELF_DIAG=1 timeout 300 python scripts/r5/config4_scans.py 2>&1 | grep 'self-knn' | tail -16 | tee $O/diag_c4.txt
for g in 1 0; do
  echo "== ICPMI_SELF_GRID=$g chains" | tee -a $O/ab.txt
  ICPMI_SELF_GRID=$g timeout 600 python scripts/r2_chain_bench.py 2>&1 | tail -5 | tee -a $O/ab.txt
done
ICPMI_SELF_DIAG=1 timeout 300 python scripts/r2_chain_bench.py 1000000 100000 5 octree 2>&1 | grep 'self-knn' | tail -6 | tee $O/diag_chain.txt
