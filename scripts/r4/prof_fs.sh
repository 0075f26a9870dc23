#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r4fsprof; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for fs in 0 1; do for chain in p2p p2plane; do
  ICPMI_FUSE_SOLVE=$fs timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_${chain}_$fs -o t -- python $R/bench.py --no-cpu --no-extras --chain $chain > /dev/null 2>&1
  f=$(find $R/$O/prof_${chain}_$fs -name "*kernel_stats.csv" | head -1); echo "== $chain fsolve=$fs"; python $R/scripts/kstats.py $f 2>/dev/null | head -5
done; done
cd $R; python scripts/r3/ktrace_series.py $O/prof_p2p_1 12 nn1_ sel2_scan accumulate_kernel solve_kernel 2>/dev/null | head -30
find $R/$O -name "*kernel_trace.csv" -delete; find $R/$O -name "*agent_info.csv" -delete
