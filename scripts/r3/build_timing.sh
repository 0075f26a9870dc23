#!/bin/bash
# builds scripts/r3/libicpmi_timing.bin: the library with the NN phase clocks compiled in (-DICPMI_NN_TIMING)
set -e
cd "$(dirname "$0")/../../norlab_icp_mapper_amd/csrc"
make -s -j8
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DICPMI_NN_TIMING -c nn.hip -o /tmp/nn_timing.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC api.o map_build.o /tmp/nn_timing.o loop.o ops.o octree.o comm.o ssn.o -o ../../scripts/r3/libicpmi_timing.bin -ldl
