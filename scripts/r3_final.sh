#!/bin/bash
# Round-3 measurement (run mid-round and again at the end): GPU tests, the default bench line (with chains + cpu_baseline), kernel traces
# of the bench chains, the batch, the 10 M map, the map-update chain, the production (checked) loop and the end-to-end loops.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r3final; mkdir -p $O
python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed" | tail -2
python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 400 $O/bench_default.json; echo
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --no-extras --no-cpu > $O/bench_torchrun1.json 2> $O/bench_torchrun1.err; tail -c 300 $O/bench_torchrun1.json; echo
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for chain in p2p p2plane docs_knn6; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$chain -o t -- python $R/bench.py --no-cpu --no-extras --chain $chain > /dev/null 2>&1
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_batch8 -o t -- python $R/bench.py --no-cpu --no-extras --batch 8 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_map10m -o t -- python $R/bench.py --no-cpu --no-extras --map-points 10000000 --scale 3.16 > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_chain -o t -- python $R/scripts/r2_chain_bench.py 1000000 100000 12 "octree, sensor" > /dev/null 2>&1
cd $R
for d in p2p p2plane docs_knn6 batch8 map10m chain; do f=$(find $O/prof_$d -name "*kernel_stats.csv" | head -1); echo "== $d: $f"; python scripts/kstats.py $f 2>/dev/null | head -8; done
python scripts/r3/ktrace_series.py $O/prof_p2p 20 nn1_ sel2_scan accumulate_kernel solve_kernel > $O/series_p2p.txt; cat $O/series_p2p.txt
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
python scripts/r2_chain_bench.py 1000000 100000 12 2>&1 | grep update | tee $O/chain_bench.txt
python scripts/e2e_bench.py 2>&1 | grep scans | tee $O/e2e_bench.txt
python scripts/r3/checked_loop_bench.py 2>/dev/null | grep "ms per" | tee $O/checked_loop.txt
