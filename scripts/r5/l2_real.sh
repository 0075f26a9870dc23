#!/bin/bash
# r5 item 1: (1) the L2-survival micro-benchmark; (2) per-dispatch FETCH_SIZE / TCC_HIT / TCC_MISS of one registration's launches in order,
# normal and with every NN launch issued twice back to back (ICPMI_NN_TWICE=1).  Output: gpurun_out/r5l2/
cd "$GRAFT_REPO_ROOT"; O=$GRAFT_REPO_ROOT/gpurun_out/r5l2; rm -rf $O; mkdir -p $O
timeout 120 scripts/r5/l2_survive.bin > $O/l2_survive.txt 2>&1; cat $O/l2_survive.txt
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu --no-extras --steps 2 --warmup 1 --chain p2p"
for mode in normal twice; do
  if [ $mode = twice ]; then export ICPMI_NN_TWICE=1; else unset ICPMI_NN_TWICE; fi
  timeout 300 rocprofv3 --pmc FETCH_SIZE WRITE_SIZE --output-format csv -d $O/$mode/fetch -- $B > /dev/null 2>$O/${mode}_fetch.err
  timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/$mode/tcc -- $B > /dev/null 2>$O/${mode}_tcc.err
  timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $O/$mode/ea -- $B > /dev/null 2>$O/${mode}_ea.err
  python $GRAFT_REPO_ROOT/scripts/r5/pmc_seq.py $O/$mode > $O/${mode}_seq.txt 2>&1
  tail -12 $O/${mode}_seq.txt
done
unset ICPMI_NN_TWICE
find $O -name "*.csv" -delete; find $O -name "*.err" -size 0 -delete
