#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r4c5; mkdir -p $O
python -m pytest tests/test_gpu_insert.py -m gpu -q -x 2>&1 | grep -a "passed\|failed\|Error\|assert\|^E " | tail -20
python -m pytest tests -m gpu -q 2>&1 | grep -a "passed\|failed" | tail -5
python bench.py --no-cpu > $O/bench.json 2>/dev/null; python - <<'PY'
import json
d=json.load(open("gpurun_out/r4c5/bench.json"))
print("headline", round(d["value"]), d["set_map_warm_ms"])
c=d["chains"]
print("config5", c["config5_stream_1gpu"].get("scans_per_s"), c["config5_stream_1gpu"].get("merge_epoch_ms"), c["config5_stream_1gpu"].get("register_ms"))
for r in ("R1","R2","R4","R8"): print(r, c["merge_loopback"][r]["merge_epoch_ms"])
print(c["merge_loopback"].get("epoch_ms_R8_over_R1"))
PY
python scripts/r2_chain_bench.py 1000000 100000 12 2>&1 | grep update
python scripts/e2e_bench.py 2>&1 | grep scans
