#!/bin/bash
# Round-2 final measurement: GPU tests, the default bench line (with chains + cpu_baseline), kernel traces of the bench chains,
# the batch, the map-update chain, a marker trace with the roctx ranges, and the bench under torch.distributed.run (1 rank).
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r2final; mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed" | tail -2
python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 600 $O/bench_default.json; echo
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --no-extras --no-cpu > $O/bench_torchrun1.json 2> $O/bench_torchrun1.err; tail -c 300 $O/bench_torchrun1.json; echo
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for chain in p2p p2plane docs_knn6; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$chain -- python $R/bench.py --no-cpu --no-extras --chain $chain > /dev/null 2>&1
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_batch8 -- python $R/bench.py --no-cpu --no-extras --batch 8 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_map10m -- python $R/bench.py --no-cpu --no-extras --map-points 10000000 --scale 3.16 > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_chain -- python $R/scripts/r2_chain_bench.py 1000000 100000 12 "octree, sensor" > /dev/null 2>&1
ICPMI_ROCTX=1 timeout 600 rocprofv3 --kernel-trace --marker-trace --stats --output-format csv -d $R/$O/prof_markers -- python $R/scripts/r2_chain_bench.py 1000000 100000 8 "octree, sensor" > /dev/null 2>&1
cd $R
for d in p2p p2plane docs_knn6 batch8 map10m chain; do f=$(find $O/prof_$d -name "*kernel_stats.csv" | head -1); echo "== $d: $f"; python scripts/kstats.py $f 2>/dev/null | head -8; done
find $O/prof_markers -name "*stats.csv" | head
python scripts/r2_chain_bench.py 1000000 100000 12 2>&1 | tail -4
python scripts/e2e_bench.py 2>&1 | tail -3
