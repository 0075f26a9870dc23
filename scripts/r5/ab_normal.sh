#!/bin/bash
# the matched normal stored next to match_pt by the NN kernel (ICPMI_NN_KEEP_NORMAL=1) vs gathered by the pair-sum kernel
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5nrm; mkdir -p $O
R=$GRAFT_REPO_ROOT
for rep in 1 2; do for k in 0 1; do
  echo "KEEP_NORMAL=$k | $(ICPMI_NN_KEEP_NORMAL=$k python bench.py --no-cpu --no-extras --chain p2plane 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), 'it/s  nn', round(d['roofline']['avg_launch_us'],2), 'us  step', round(d['step_ms']['median'],4), 'ms  err_gt', d['pose_err_vs_ground_truth']['m'])")"
done; done | tee $O/ab.txt
cd /tmp && export TMPDIR=/tmp
for k in 0 1; do
  ICPMI_NN_KEEP_NORMAL=$k timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$k -o t -- python $R/bench.py --no-cpu --no-extras --chain p2plane > /dev/null 2>&1
  f=$(find $R/$O/prof_$k -name "*kernel_stats.csv" | head -1); echo "== KEEP_NORMAL=$k"; python $R/scripts/kstats.py $f 2>/dev/null | head -5
done | tee $O/kstats.txt
for k in 0 1; do
  ICPMI_NN_KEEP_NORMAL=$k timeout 300 rocprofv3 --pmc FETCH_SIZE WRITE_SIZE --output-format csv -d $R/$O/pmc_$k -- python $R/bench.py --no-cpu --no-extras --steps 3 --warmup 1 --chain p2plane > /dev/null 2>&1
  echo "== KEEP_NORMAL=$k"; python $R/scripts/r5/pmc_seq.py $R/$O/pmc_$k 2>/dev/null | grep -a "^accumulate\|^nn1_wg" 
done | tee $O/pmc.txt
find $R/$O -name "*.csv" -delete
