#!/bin/bash
# round-2 GPU check 2: batch + run-ahead tests, full parity suite, bench with the `chains` object, e2e A/B of the run-ahead loop
set -x
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2c2; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_batch.py -x -q > $O/pytest_batch.log 2>&1; echo "rc=$?" >> $O/pytest_batch.log; tail -15 $O/pytest_batch.log
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -5 $O/pytest.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/r2c2/bench.json"))
print("value", d["value"], d["step_ms"], "roofline", d["roofline"]["frac"], d["roofline"]["avg_launch_us"])
for k,v in d.get("chains",{}).items(): print(k, {kk:v[kk] for kk in v if kk in ("value","step_ms","error","first_reading_equals_single_registration","errors")}, v.get("roofline",{}).get("frac"), v.get("roofline",{}).get("avg_launch_us"))
print("cpu", d.get("cpu_baseline",{}).get("value"), d.get("pose_err_vs_cpu"))
PY
tail -3 $O/bench.err
for ra in 0 2 3; do echo "RUN_AHEAD=$ra"; ICPMI_RUN_AHEAD=$ra timeout 300 python scripts/e2e_bench.py 1000000 100000 12 2>&1 | tail -2; done | tee $O/e2e.txt
for b in 2 4 8 16; do timeout 300 python bench.py --no-cpu --no-extras --batch $b 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('batch', $b, d['value'], d['step_ms'])"; done | tee $O/batch_sweep.txt
timeout 300 python bench.py --no-cpu --no-extras --batch 8 --chain p2plane 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('batch 8 p2plane', d['value'], d['step_ms'])" | tee -a $O/batch_sweep.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_batch8 -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --no-extras --batch 8 > /dev/null 2>&1
python $GRAFT_REPO_ROOT/scripts/kstats.py $GRAFT_REPO_ROOT/$O/prof_batch8 | head -8
