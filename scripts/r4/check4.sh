#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r4c4; mkdir -p $O
python -m pytest tests -m gpu -q 2>&1 | grep -a "passed\|failed\|Error\|assert" | tail -8
python bench.py --no-cpu > $O/bench.json 2>/dev/null; python - <<'PY'
import json
d=json.load(open("gpurun_out/r4c4/bench.json"))
print("headline", round(d["value"]), d["roofline"]["avg_launch_us"], d["roofline"]["frac"])
for k,v in d["chains"].items(): print(k, round(v["value"]) if "value" in v else v.get("error"), v.get("same_bits_as_headline"), v.get("ms_per_registration"))
PY
