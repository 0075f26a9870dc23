#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_merge_loopback.py tests/test_gpu_configs.py -m gpu -q -x 2>&1 | grep -a "passed\|failed\|Error\|assert" | tail -5
for fa in 1 0; do echo "ICPMI_MERGE_FLAG_ALL=$fa"; ICPMI_MERGE_FLAG_ALL=$fa timeout 300 python scripts/r5/ab_epoch.py 6 2>/dev/null | grep "block 32768" | head -4; done
