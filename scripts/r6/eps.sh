#!/bin/bash
# r6: epsilon-approximate matcher -- property tests, the exact path's bits (parity + knn_wg + sel_window tests), bench legs
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6_eps; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_epsilon.py -m gpu -x -q 2>&1 | tail -25 | tee $O/tests.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_knn_wg.py tests/test_gpu_sel_window.py tests/test_pins.py -m gpu -x -q 2>&1 | tail -8 | tee -a $O/tests.txt
timeout 900 python bench.py --no-cpu > $O/bench.json 2> $O/bench.err
python - <<'PY' | tee $O/summary.txt
import json
d = json.load(open("gpurun_out/r6_eps/bench.json"))
print("headline", d["value"], d["roofline"]["avg_launch_us"])
c = d["chains"]
for k in ("docs_knn6", "docs_knn6_epsilon1"):
    print(k, {x: c[k].get(x) for x in ("value", "step_ms", "pose_err_vs_ground_truth", "pose_diff_vs_exact_search", "error")})
for k in ("config4_replay", "config4_replay_epsilon1"):
    print(k, {x: c[k].get(x) for x in ("value", "register_ms_per_scan", "update_ms_per_scan", "icp_iterations", "error")})
PY
