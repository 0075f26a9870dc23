#!/bin/bash
# pair-sum workgroups (ICPMI_ACC_BLOCKS): knn 6 has 600 k pairs -- 256 workgroups of 1024 leave a third, dependent, pair per lane
cd "$GRAFT_REPO_ROOT"; R=$GRAFT_REPO_ROOT
for rep in 1 2; do
for cap in 256 300 400 600; do
  for chain in docs_knn6 p2p; do
    ICPMI_ACC_BLOCKS=$cap python $R/bench.py --no-cpu --no-extras --chain $chain 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cap=$cap $chain', round(d['value']), d['ms_per_step'])"
  done
done
done
