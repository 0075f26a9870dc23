#!/bin/bash
# builds scripts/r3/libicpmi_<tag>.bin: the WHOLE library compiled with extra flags (objects under /tmp).  usage: build_variant_all.sh <tag> <flags...>
set -e
tag=$1; shift
cd "$(dirname "$0")/../../norlab_icp_mapper_amd/csrc"
mkdir -p /tmp/var_$tag
for f in api map_build nn loop ops octree comm ssn; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off "$@" -c $f.hip -o /tmp/var_$tag/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/var_$tag/*.o -o ../../scripts/r3/libicpmi_$tag.bin -ldl
