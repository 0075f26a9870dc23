#!/bin/bash
# tile-cooperative seed pass in front of the first NN launch (ICPMI_NN_TILE_SEED=1) -- bits, then the clocks
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5tile; mkdir -p $O
ICPMI_NN_TILE_SEED=1 timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_golden.py tests/test_gpu_ext_filters.py tests/test_gpu_fuzz.py tests/test_gpu_planar.py -m gpu -q -x 2>&1 | grep -a "passed\|failed\|Error\|assert" | tail -5 | tee $O/tests.txt
for rep in 1 2; do for ts in 0 1; do
  for chain in p2p p2plane; do
    ICPMI_NN_TILE_SEED=$ts python bench.py --no-cpu --no-extras --chain $chain 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('tile_seed=$ts $chain', round(d['value']), d['ms_per_step'], d['roofline']['avg_launch_us'], d['pose_err_vs_ground_truth']['m'])"
  done
  ICPMI_NN_TILE_SEED=$ts python scripts/r3/checked_loop_bench.py 2>/dev/null | grep "ms per" | sed "s/^/tile_seed=$ts /"
done; done 2>&1 | tee $O/ab.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for ts in 0 1; do
ICPMI_NN_TILE_SEED=$ts timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/$O/p2p_$ts -o t -- python $R/bench.py --no-cpu --no-extras --chain p2p --steps 10 --warmup 3 > /dev/null 2>&1
echo "== tile_seed=$ts"; python $R/scripts/r3/ktrace_series.py $R/$O/p2p_$ts 20 nn1_wg nn1_tile | cut -c1-200
done | tee $O/series.txt
find $R/$O -name "*.csv" -delete
