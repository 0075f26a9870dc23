#!/bin/bash
# r5 item 1, second look WITHOUT the profiler (its per-dispatch counter collection may itself flush the caches between dispatches):
# (1) micro-benchmark with a writing warm-up launch, the one-workgroup launch in between, the XCC ids;
# (2) the NN launch timed by HIP events (bench.py's roofline leg, eager launches): normal | L2 thrashed in front of every launch | every launch
#     twice inside the event pair | both.  Output: gpurun_out/r5l2b/
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5l2b; mkdir -p $O
true
for rep in 1 2; do for ch in p2p p2plane; do for mode in "" "ICPMI_NN_FLUSH_L2=1" "ICPMI_NN_TWICE=1" "ICPMI_NN_TWICE=1 ICPMI_NN_FLUSH_L2=1"; do
  echo "[$mode] $ch | $(env $mode timeout 300 python bench.py --no-extras --no-cpu --chain $ch 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), 'it/s  nn events', round(d['roofline']['avg_launch_us'],2), 'us  step', round(d['step_ms']['median'],4), 'ms')")"
done; done; done 2>&1 | tee $O/nn_events.txt
