#!/bin/bash
# r6 iteration loop: parity of the self-search paths, config 4 + chains A/B (ICPMI_SELF_GRID=1 | 0), per-kernel table
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6_iter; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_pins.py tests/test_gpu_map_chain.py tests/test_gpu_planar.py tests/test_host_filters.py -m gpu -x -q 2>&1 | tail -8 | tee $O/tests.txt
for g in 1 0; do
  echo "== ICPMI_SELF_GRID=$g config4" | tee -a $O/ab.txt
  ICPMI_SELF_GRID=$g timeout 300 python scripts/r5/config4.py 2>&1 | tail -1 | tee -a $O/ab.txt
  ICPMI_SELF_GRID=$g timeout 600 python scripts/r2_chain_bench.py 2>&1 | tail -4 | tee -a $O/ab.txt
done
ICPMI_SELF_DIAG=1 timeout 300 python scripts/r5/config4_scans.py 2>&1 | grep 'self-knn' | tail -3 | tee $O/diag_c4.txt
bash scripts/r6/kstats.sh 16
