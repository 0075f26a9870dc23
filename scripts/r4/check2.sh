#!/bin/bash
# r4 check 2: the new parity tests (recalled vectors, full-size configs 2 / 3, ADVICE fixes)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r4c2; mkdir -p $O
python -m pytest tests/test_gpu_recalled.py tests/test_gpu_ext_filters.py -m gpu -q 2>&1 | tail -40 > $O/pytest_a.txt; tail -5 $O/pytest_a.txt
python -m pytest tests/test_gpu_configs.py -m gpu -q -k "full_size" --durations=5 2>&1 | tail -40 > $O/pytest_b.txt; tail -12 $O/pytest_b.txt
