#!/bin/bash
# r5 long soak: 10 000 registration + 5 000 chain draws of the randomised differential tests, 3 000 / 2 000 / 3 000 operator-soak cases
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5soak_long; mkdir -p $O
ICPMI_FUZZ_N=10000 ICPMI_FUZZ_SEED=77000 timeout 3000 python -m pytest tests/test_gpu_fuzz.py -m gpu -q 2>&1 | grep -a "passed\|failed\|Error\|assert" | tail -5 | tee $O/fuzz.txt
timeout 900 python tests/tools/soak_chain.py 3000 91 2>&1 | tail -2 | tee $O/soak_chain.txt
timeout 900 python tests/tools/soak_ops.py 2000 91 2>&1 | tail -2 | tee $O/soak_ops.txt
timeout 900 python tests/tools/soak.py 3000 91 2>&1 | tail -2 | tee $O/soak.txt
