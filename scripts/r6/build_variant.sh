#!/bin/bash
# builds scripts/r6/libicpmi_<tag>.bin: the whole library compiled with extra flags (fresh objects under /tmp).  usage: build_variant.sh <tag> <flags...>
tag=$1; shift
D=/root/repo/norlab_icp_mapper_amd/csrc
rm -rf /tmp/var6_$tag; mkdir -p /tmp/var6_$tag
for f in api map_build nn selfgrid loop ops octree comm ssn cells; do
  ( cd $D && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off "$@" -c $f.hip -o /tmp/var6_$tag/$f.o 2>/tmp/var6_$tag/$f.log || echo "FAILED $f" ) &
done
wait
grep -l "error:" /tmp/var6_$tag/*.log
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/var6_$tag/*.o -o /root/repo/scripts/r6/libicpmi_$tag.bin -ldl && nm -D /root/repo/scripts/r5/libicpmi_$tag.bin | grep -c " T icpmi_"
