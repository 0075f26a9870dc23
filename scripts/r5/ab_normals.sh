#!/bin/bash
# normals_kernel with the neighbourhood in registers (default) against the generic walk (ICPMI_NORMALS_REG=0): map-update chains, one call
cd "$GRAFT_REPO_ROOT"
for rep in 1 2 3; do for R in 0 1; do
  echo "== ICPMI_NORMALS_REG=$R (rep $rep)"
  ICPMI_NORMALS_REG=$R python scripts/r2_chain_bench.py 1000000 100000 12 2>&1 | grep update
done; done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5normals; mkdir -p $O
for V in 0 1; do
  ICPMI_NORMALS_REG=$V timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/pd_$V -o t -- python $R/scripts/r2_chain_bench.py 1000000 100000 8 "point_distance" > /dev/null 2>&1
  ICPMI_NORMALS_REG=$V timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/oct_$V -o t -- python $R/scripts/r2_chain_bench.py 1000000 100000 8 "octree, sensor" > /dev/null 2>&1
  for c in pd oct; do f=$(find $O/${c}_$V -name "*kernel_stats.csv" | head -1); echo "== $c ICPMI_NORMALS_REG=$V"; python $R/scripts/kstats.py $f 2>/dev/null | grep -i "normals_kernel\|nnk_self"; done
done
find $O -name "*.csv" -delete
