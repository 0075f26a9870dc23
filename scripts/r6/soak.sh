#!/bin/bash
# soak on the r6 code (block-grid self search, incremental normals, cell log, epsilon plumbing): fresh seeds
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6soak; mkdir -p $O
ICPMI_FUZZ_N=6000 ICPMI_FUZZ_SEED=${FUZZ_SEED:-211000} timeout 3000 python -m pytest tests/test_gpu_fuzz.py -m gpu -q 2>&1 | grep -a "passed\|failed\|Error\|assert" | tail -5 | tee $O/fuzz.txt
timeout 900 python tests/tools/soak_chain.py 2000 ${SOAK_SEED:-197} 2>&1 | tail -2 | tee $O/soak_chain.txt
timeout 900 python tests/tools/soak_ops.py 1500 ${SOAK_SEED:-197} 2>&1 | tail -2 | tee $O/soak_ops.txt
timeout 900 python tests/tools/soak.py 2000 ${SOAK_SEED:-197} 2>&1 | tail -2 | tee $O/soak.txt
