#!/bin/bash
# wide (13 + 19 bit) selection for k > 1: parity tests, then A/B against ICPMI_SEL13=0 in one call, then the kernel table
cd "$GRAFT_REPO_ROOT"; R=$GRAFT_REPO_ROOT; O=gpurun_out/${1:-sel13}; mkdir -p $O
python -m pytest tests/test_gpu_knn_wg.py tests/test_gpu_parity.py tests/test_gpu_ext_filters.py tests/test_gpu_configs.py tests/test_gpu_batch.py -m gpu -q -x 2>&1 | grep -a "passed\|failed\|^FAILED\|^E  " | head -20
ICPMI_FUZZ_N=300 python -m pytest tests/test_gpu_fuzz.py -q -k random_chain 2>&1 | grep -a "passed\|failed\|^FAILED\|^E  " | head -20
for rep in 1 2; do for v in 1 0; do
  ICPMI_SEL13=$v python $R/bench.py --no-cpu --no-extras --chain docs_knn6 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('sel13=$v docs_knn6', round(d['value']), d['ms_per_step'], d['pose_err_vs_ground_truth'])"
done; done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o t -- python $R/bench.py --no-cpu --no-extras --chain docs_knn6 > /dev/null 2>&1
f=$(find $R/$O/prof -name "*kernel_stats.csv" | head -1); python $R/scripts/kstats.py $f 2>/dev/null | head -8
find $R/$O -name "*kernel_trace.csv" -delete; find $R/$O -name "*agent_info.csv" -delete
