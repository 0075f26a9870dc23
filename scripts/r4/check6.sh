#!/bin/bash
cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_insert.py -m gpu -q 2>&1 | grep -a "passed\|failed\|Error\|assert\|^E " | tail -20
python -m pytest tests -m gpu -q 2>&1 | grep -a "passed\|failed\|^FAILED" | tail -8
