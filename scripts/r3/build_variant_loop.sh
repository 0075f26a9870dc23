#!/bin/bash
# builds scripts/r3/libicpmi_<tag>.bin: the library with loop.hip compiled with extra flags.  usage: build_variant_loop.sh <tag> <flags...>
set -e
tag=$1; shift
cd "$(dirname "$0")/../../norlab_icp_mapper_amd/csrc"
make -s -j8
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off "$@" -c loop.hip -o /tmp/loop_$tag.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC api.o map_build.o nn.o /tmp/loop_$tag.o ops.o octree.o comm.o ssn.o -o ../../scripts/r3/libicpmi_$tag.bin -ldl
