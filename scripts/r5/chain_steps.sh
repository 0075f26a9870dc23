#!/bin/bash
# step clocks of the shipped chain (ICPMI_CHAIN_TIMING: a stream sync behind every step), last three of 8 updates
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5steps
ICPMI_CHAIN_TIMING=1 python scripts/r2_chain_bench.py 1000000 100000 8 "octree, sensor" 2>&1 | tail -34 | tee gpurun_out/r5steps/steps.txt
