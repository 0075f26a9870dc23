#!/bin/bash
# r5 soak after the chain / scan / merge changes: the randomised differential tests widened, the operator soaks, both on fresh seeds
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5soak; mkdir -p $O
ICPMI_FUZZ_N=1500 ICPMI_FUZZ_SEED=5000 timeout 2400 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -x 2>&1 | grep -a "passed\|failed\|Error\|assert" | tail -5 | tee $O/fuzz.txt
timeout 900 python tests/tools/soak_chain.py 600 55 2>&1 | tail -3 | tee $O/soak_chain.txt
timeout 900 python tests/tools/soak_ops.py 400 55 2>&1 | tail -3 | tee $O/soak_ops.txt
timeout 900 python tests/tools/soak.py 600 55 2>&1 | tail -3 | tee $O/soak.txt
