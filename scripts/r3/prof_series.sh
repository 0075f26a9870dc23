#!/bin/bash
# usage: prof_series.sh <tag> <chain> [ENV=val ...]   -- kernel trace of a short bench run, per-launch series of the loop's kernels
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
tag=$1; chain=$2; shift; shift
out=/tmp/prof_$tag; rm -rf $out
env "$@" rocprofv3 --kernel-trace --stats -d $out -o t --output-format csv -- python bench.py --no-extras --no-cpu --chain $chain --steps 4 --warmup 2 > /tmp/prof_$tag.log 2>&1
echo "== $tag ($chain $@)"
python scripts/r3/ktrace_series.py $out 20 nn1_ sel2_scan accumulate_kernel solve_kernel
mkdir -p gpurun_out/r3; cp $(find $out -name "*kernel_stats.csv" | head -1) gpurun_out/r3/${tag}_kernel_stats.csv 2>/dev/null
