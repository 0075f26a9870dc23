#!/bin/bash
# per-iteration NN kernel durations (rocprofv3 kernel trace) of the p2p chain at two scan sizes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6_series; rm -rf $O; mkdir -p $O
for n in 98304 100000; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p$n -o t -- python $R/bench.py --no-cpu --no-extras --chain ${1:-p2p} --scan-points $n --min-seconds 0 > /dev/null 2>&1
  echo "== $n"; python $R/scripts/r3/ktrace_series.py $O/p$n 20 nn1_ sel2_scan accumulate_kernel solve_kernel | head -2
done
rm -rf $O
