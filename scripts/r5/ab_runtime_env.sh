#!/bin/bash
# HIP runtime switches against the headline and the checked loop: one call, alternating (boxes differ by ~4 %)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/abenv; mkdir -p $O
one() { # label, env...
  local lab=$1; shift
  local v=$(env "$@" python bench.py --no-cpu --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['roofline']['avg_launch_us'],2))")
  local c=$(env "$@" python scripts/r3/checked_loop_bench.py 2>/dev/null | grep "p2plane" | sed 's/.*p2plane: \([0-9.]*\) ms.*/\1/')
  echo "$lab : headline it/s, NN us = $v ; checked p2plane ms = $c" | tee -a $O/ab.txt
}
for rep in 1 2; do
one "default            " A=1
one "DEV_KERNARG=1      " HIP_FORCE_DEV_KERNARG=1
one "DEV_KERNARG=0      " HIP_FORCE_DEV_KERNARG=0
one "GRAPH_PKT_CAPTURE=0" DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
one "GRAPH_PKT_CAPTURE=1" DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
one "KERNARG_COPY_OPT=0 " DEBUG_HIP_KERNARG_COPY_OPT=0
one "HDP_FLUSH_WA=0     " DEBUG_CLR_KERNARG_HDP_FLUSH_WA=0
one "FGS_KERNARG=0      " ROC_USE_FGS_KERNARG=0
one "MAX_HW_QUEUES=1    " GPU_MAX_HW_QUEUES=1
one "OPT_FLUSH=0        " AMD_OPT_FLUSH=0
one "ACTIVE_WAIT=100000 " ROC_ACTIVE_WAIT_TIMEOUT=100000
done
