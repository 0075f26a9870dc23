#!/bin/bash
# usage: prof_knn.sh <tag> [ENV=val ...]   -- kernel trace of the knn-6 chain, per-launch series of its loop kernels
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
tag=$1; shift
out=/tmp/prof_$tag; rm -rf $out
env "$@" rocprofv3 --kernel-trace --stats -d $out -o t --output-format csv -- python bench.py --no-extras --no-cpu --chain docs_knn6 --steps 4 --warmup 2 > /tmp/prof_$tag.log 2>&1
echo "== $tag (docs_knn6 $@)"
python scripts/r3/ktrace_series.py $out 20 nnk_ml nnk_wg sel2_hist0 sel2_scan accumulate_kernel solve_kernel
mkdir -p gpurun_out/r3; cp $(find $out -name "*kernel_stats.csv" | head -1) gpurun_out/r3/${tag}_kernel_stats.csv 2>/dev/null
