#!/bin/bash
# the speculative level 0 of the k > 1 loop (ICPMI_SEL_WIN) and the pair sums' skipped gathers (ICPMI_ACC_SKIP: built for this A/B, no gain, removed afterwards -- the switch is a no-op in the tree), one call
cd "$GRAFT_REPO_ROOT"
one() { env "$@" python bench.py --no-cpu --no-extras --chain "$CH" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('   $CH', round(d['value']), 'it/s  ms_per_step', round(d['ms_per_step'],4))"; }
run() { echo "== $1"; shift; for CH in docs_knn6 p2plane; do one "$@"; done; }
for rep in 1 2 3; do
  run "both off (rep $rep)" ICPMI_SEL_WIN=0 ICPMI_ACC_SKIP=0
  run "window only (rep $rep)" ICPMI_ACC_SKIP=0
  run "skip only (rep $rep)" ICPMI_SEL_WIN=0
  run "default: both (rep $rep)" A=1
done
echo "== headline (k = 1 point-to-point: neither applies)"
python bench.py --no-cpu --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('   headline', round(d['value']), 'it/s')"
python - <<'P'
import numpy as np, norlab_icp_mapper_amd as pkg
sc = pkg.synth.make_scene(m=1_000_000, n=100_000)
icp = pkg.ICPSequence(minimizer=2, knn=6, max_dist=2.0, outliers=[(4, 0.85)], max_iterations=20, use_differential=0)
icp.setMap(sc["map"], sc["normals"])
for rep in range(2):
    icp(sc["scan"]); d = icp.debugCounters()
    print("knn 6, 100 k x 1 M, 20 iterations: window served", int(d[12]), "iterations, missed", int(d[13]))
P
