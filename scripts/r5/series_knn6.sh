#!/bin/bash
# per-launch kernel durations of one knn-6 registration with and without the speculative window
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5series6; mkdir -p $O
for W in 1 0; do
  ICPMI_SEL_WIN=$W timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/knn6_w$W -o t -- python $R/bench.py --no-cpu --no-extras --chain docs_knn6 --steps 10 --warmup 3 > /dev/null 2>&1
  echo "== ICPMI_SEL_WIN=$W" >> $O/series_knn6.txt
  python $R/scripts/r3/ktrace_series.py $O/knn6_w$W 2 nnk_ml >> $O/series_knn6.txt; python $R/scripts/r3/ktrace_series.py $O/knn6_w$W 18 nnk_wg sel2_hist0 sel2_scan accumulate solve_kernel >> $O/series_knn6.txt
done
cat $O/series_knn6.txt
find $O -name "*.csv" -delete
