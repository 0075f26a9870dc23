#!/bin/bash
# round-2 GPU check 1: parity suite with the 3-launch iteration, A/B bench against the 4-launch chain, kernel trace
set -x
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r2c1
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r2c1/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2c1/pytest.log
tail -5 gpurun_out/r2c1/pytest.log
for chain in p2p p2plane docs_knn6; do
  for it3 in 1 0; do
    ICPMI_ITER3=$it3 timeout 300 python bench.py --no-cpu --chain $chain > gpurun_out/r2c1/bench_${chain}_iter3_${it3}.json 2>gpurun_out/r2c1/bench_${chain}_iter3_${it3}.err
    echo "$chain iter3=$it3: $(cut -c1-400 gpurun_out/r2c1/bench_${chain}_iter3_${it3}.json)"
  done
done
python scripts/solve_cycles.py > gpurun_out/r2c1/solve_cycles.txt 2>&1; cat gpurun_out/r2c1/solve_cycles.txt
cd /tmp && export TMPDIR=/tmp
for chain in p2p p2plane; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2c1/prof_$chain -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --chain $chain > /dev/null 2>&1
  python $GRAFT_REPO_ROOT/scripts/kstats.py $GRAFT_REPO_ROOT/gpurun_out/r2c1/prof_$chain | head -12
done
