#!/bin/bash
cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_merge_loopback.py -m gpu -q 2>&1 | grep -a "passed\|failed"
python scripts/r4/loopback_bench.py 1 8 2>/dev/null | grep epoch
python scripts/r4/loopback_bench.py 8 8 2>/dev/null | grep epoch
bash scripts/r4/prof_loop.sh 2>/dev/null | grep "merge_flag\|last epoch"
