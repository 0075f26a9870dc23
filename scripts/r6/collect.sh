#!/bin/bash
# copy the summaries of a scripts/r6_final.sh run (gpurun_out/<name>) into profiles/ as r6_<tag>_*
# usage: scripts/r6/collect.sh r6mid mid
S=gpurun_out/$1; T=profiles/r6_$2
for d in p2p p2plane docs_knn6 batch8 map10m config5 chain config4; do
  f=$(find $S/prof_$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f ${T}_${d}_kernel_stats.csv
done
cp $S/config4_kernel_stats.txt ${T}_config4_kernel_stats.txt
cp $S/bench_default.json ${T}_bench_default.json
cp $S/bench_torchrun1.json ${T}_bench_torchrun_1rank.json
cp $S/bench_config5_torchrun1.json ${T}_bench_config5_torchrun_1rank.json
cp $S/series_p2p.txt ${T}_series_p2p.txt; cp $S/series_knn6.txt ${T}_series_knn6.txt
cat $S/gpu_tests.txt $S/chain_bench.txt $S/e2e_bench.txt $S/checked_loop.txt > ${T}_chain_e2e_checked.txt
cp $S/config4_scans.txt ${T}_config4_scans.txt
