#!/bin/bash
# r4 check 1: GPU tests after the ADVICE fixes (graph invalidation, sensor-noise sentinel, planar guards)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r4c1; mkdir -p $O
python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > $O/pytest.txt; tail -5 $O/pytest.txt
