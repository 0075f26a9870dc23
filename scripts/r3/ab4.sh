cd $GRAFT_REPO_ROOT
cp norlab_icp_mapper_amd/libicpmi.so /tmp/prod.so
for rep in 1 2; do for tag in occ6 prod; do
  if [ $tag = prod ]; then cp /tmp/prod.so norlab_icp_mapper_amd/libicpmi.so; else cp scripts/r3/libicpmi_$tag.bin norlab_icp_mapper_amd/libicpmi.so; fi
  for args in "--chain p2p" "--chain p2plane" "--chain p2p --batch 8" "--chain p2p --map-points 10000000 --scale 3.16"; do
    echo "$tag | $args | $(python bench.py --no-extras --no-cpu $args 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']))")"
  done
  python scripts/r3/checked_loop_bench.py 2>/dev/null | grep "ms per" | cut -c1-60 | sed "s/^/$tag | /"
done; done
cp /tmp/prod.so norlab_icp_mapper_amd/libicpmi.so
