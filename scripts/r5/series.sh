#!/bin/bash
# per-launch kernel durations of one registration: knn-6 chain, p2p, checked p2plane
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5series; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/knn6 -o t -- python $R/bench.py --no-cpu --no-extras --chain docs_knn6 --steps 10 --warmup 3 > /dev/null 2>&1
python $R/scripts/r3/ktrace_series.py $O/knn6 2 nnk_ml > $O/series_knn6.txt; python $R/scripts/r3/ktrace_series.py $O/knn6 18 nnk_wg sel2_hist0 sel2_scan accumulate solve_kernel >> $O/series_knn6.txt; cat $O/series_knn6.txt
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/p2p -o t -- python $R/bench.py --no-cpu --no-extras --chain p2p --steps 10 --warmup 3 > /dev/null 2>&1
python $R/scripts/r3/ktrace_series.py $O/p2p 20 nn1_ sel2_scan accumulate solve_kernel qfirst qpass qcount > $O/series_p2p.txt; cat $O/series_p2p.txt
find $O -name "*.csv" -delete
