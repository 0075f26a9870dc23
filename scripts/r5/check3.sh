#!/bin/bash
# r5: full GPU suite after the chain + one-collective epoch changes; epoch timing one-collective vs three-collective on the loopback communicator
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/${1:-r5c4}; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | grep -a "passed\|failed\|Error\|assert" | tail -8 | tee $O/tests.txt
for blk in 32768 0; do for R in 1 2 4 8; do
  echo "ICPMI_MERGE_BLOCK=$blk $(ICPMI_MERGE_BLOCK=$blk python scripts/r4/loopback_bench.py $R 6 2>/dev/null | tail -1 | cut -c1-200)"
done; done | tee $O/loopback.txt
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --workload config5 --scans 8 > $O/bench_config5_torchrun1.json 2> $O/bench_config5.err; tail -c 900 $O/bench_config5_torchrun1.json; echo
ICPMI_SELF_DIAG=1 python scripts/r2_chain_bench.py 1000000 100000 4 "octree, sensor" 2>&1 | tail -4
python scripts/r2_chain_bench.py 1000000 100000 12 2>&1 | grep update | tee $O/chain_bench.txt
