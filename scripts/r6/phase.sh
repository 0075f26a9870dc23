#!/bin/bash
# phase clocks of nn1_wg_kernel (library built with -DICPMI_NN_TIMING: s_waitcnt 0 + clock64 at the phase borders, serialised)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6_phase; mkdir -p $O
cp norlab_icp_mapper_amd/libicpmi.so /tmp/prod.so
cp scripts/r6/libicpmi_timing.bin norlab_icp_mapper_amd/libicpmi.so
for mz in 1 2; do echo "== minimizer $mz"; timeout 300 python scripts/nn_phase.py 100000 $mz 2>&1 | tail -6; done | tee $O/phase.txt
cp /tmp/prod.so norlab_icp_mapper_amd/libicpmi.so
