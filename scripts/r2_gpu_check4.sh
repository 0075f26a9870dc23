#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r2c4; mkdir -p $O
run() { timeout 300 python bench.py --no-cpu --no-extras "$@" 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(round(d['value']), round(d['step_ms']['median'],3))"; }
for ch in p2p p2plane; do
  echo "$ch single default: $(run --chain $ch)"
  echo "$ch batch8 default(G4 narrow, G8 wide): $(run --chain $ch --batch 8)"
  echo "$ch batch8 wide16: $(ICPMI_NN_WIDE16=1 run --chain $ch --batch 8)"
  echo "$ch batch8 G2: $(ICPMI_NN_G=2 run --chain $ch --batch 8)"
  echo "$ch batch16 default: $(run --chain $ch --batch 16)"
  echo "$ch batch16 G2: $(ICPMI_NN_G=2 run --chain $ch --batch 16)"
done | tee $O/g_sweep2.txt
python -m pytest tests/test_gpu_batch.py -q 2>&1 | tail -2
