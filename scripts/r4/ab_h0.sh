#!/bin/bash
# does the level-0 histogram in the NN kernel's tail (100 k fine atomics) cost the kernel?  ICPMI_NN_FUSE_HIST0=0 moves it to the stand-alone builder
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/${1:-abh0}; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for h0 in 1 0; do
  for chain in p2p p2plane; do
    export ICPMI_NN_FUSE_HIST0=$h0
    python $R/bench.py --no-cpu --no-extras --chain $chain 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fuse_h0=$h0 $chain', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])"
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_${chain}_$h0 -o t -- python $R/bench.py --no-cpu --no-extras --chain $chain > /dev/null 2>&1
    f=$(find $R/$O/prof_${chain}_$h0 -name "*kernel_stats.csv" | head -1); python $R/scripts/kstats.py $f 2>/dev/null | head -7
  done
done
find $R/$O -name "*kernel_trace.csv" -delete; find $R/$O -name "*agent_info.csv" -delete
