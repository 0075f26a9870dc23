#!/bin/bash
# soak on the FINAL r5 code (fast finish, capture gate, tagged counts, radix look-back ...): fresh seeds
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5soak_final; mkdir -p $O
ICPMI_FUZZ_N=6000 ICPMI_FUZZ_SEED=131000 timeout 3000 python -m pytest tests/test_gpu_fuzz.py -m gpu -q 2>&1 | grep -a "passed\|failed\|Error\|assert" | tail -5 | tee $O/fuzz.txt
timeout 900 python tests/tools/soak_chain.py 2000 137 2>&1 | tail -2 | tee $O/soak_chain.txt
timeout 900 python tests/tools/soak_ops.py 1500 137 2>&1 | tail -2 | tee $O/soak_ops.txt
timeout 900 python tests/tools/soak.py 2000 137 2>&1 | tail -2 | tee $O/soak.txt
