#!/bin/bash
# Round-6 measurement: GPU tests, the default bench line (headline + chains + cpu_baseline), the config-5 workload under torchrun with one rank,
# kernel traces of the bench chains, the batch, the 10 M map, the config-5 stream, the map-update chain, config 4 (real lidar) replay.
# Output: gpurun_out/<name> (copied into profiles/ as r6_<name>_* by scripts/r6/collect.sh).
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/${1:-r6final}; mkdir -p $O
python -m pytest tests -m gpu -q 2>&1 | grep -a "passed\|failed\|^FAILED" | tail -4 | tee $O/gpu_tests.txt
python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 300 $O/bench_default.json; echo
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29521 bench.py --gpus 1 --no-extras --no-cpu > $O/bench_torchrun1.json 2> $O/bench_torchrun1.err; tail -c 300 $O/bench_torchrun1.json; echo
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29522 bench.py --gpus 1 --workload config5 --scans 8 > $O/bench_config5_torchrun1.json 2> $O/bench_config5.err; tail -c 600 $O/bench_config5_torchrun1.json; echo
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for chain in p2p p2plane docs_knn6; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$chain -o t -- python $R/bench.py --no-cpu --no-extras --chain $chain > /dev/null 2>&1
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_batch8 -o t -- python $R/bench.py --no-cpu --no-extras --batch 8 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_map10m -o t -- python $R/bench.py --no-cpu --no-extras --map-points 10000000 --scale 3.16 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_config5 -o t -- python $R/bench.py --workload config5 --scans 8 > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_chain -o t -- python $R/scripts/r2_chain_bench.py 1000000 100000 12 "octree, sensor" > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_config4 -o t -- python $R/scripts/r5/config4_scans.py > $R/$O/run_config4_traced.txt 2>&1
cd $R
for d in p2p p2plane docs_knn6 batch8 map10m config5 chain config4; do f=$(find $O/prof_$d -name "*kernel_stats.csv" | head -1); echo "== $d: $f"; python scripts/kstats.py $f 2>/dev/null | head -12; done
f=$(find $O/prof_config4 -name "*kernel_stats.csv" | head -1); python scripts/kstats.py $f 2>/dev/null | head -40 > $O/config4_kernel_stats.txt
python scripts/r3/ktrace_series.py $O/prof_p2p 20 nn1_ sel2_scan accumulate_kernel solve_kernel > $O/series_p2p.txt; cat $O/series_p2p.txt
python scripts/r3/ktrace_series.py $O/prof_docs_knn6 20 nnk_ sel2_ accumulate_kernel solve_kernel > $O/series_knn6.txt 2>/dev/null; head -30 $O/series_knn6.txt
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
python scripts/r2_chain_bench.py 1000000 100000 12 2>&1 | grep update | tee $O/chain_bench.txt
python scripts/e2e_bench.py 2>&1 | grep scans | tee $O/e2e_bench.txt
python scripts/r3/checked_loop_bench.py 2>/dev/null | grep "ms per" | tee $O/checked_loop.txt
timeout 300 python scripts/r5/config4_scans.py 2>/dev/null | tail -18 | tee $O/config4_scans.txt
