cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for n in 70000 81920 86000 100000; do
  rm -rf /tmp/pk; rocprofv3 --kernel-trace --stats -d /tmp/pk -o t --output-format csv -- python bench.py --no-extras --no-cpu --chain docs_knn6 --scan-points $n --steps 4 --warmup 2 > /dev/null 2>&1
  echo "n=$n $(python scripts/r3/ktrace_series.py /tmp/pk 20 nnk_wg | head -1 | sed 's/.*steady/steady/')"
done
