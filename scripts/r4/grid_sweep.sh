#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do for t in 4 6 8 12 16; do for ch in p2p p2plane; do
echo "target $t | $ch | $(ICPMI_GRID_TARGET=$t timeout 300 python bench.py --no-extras --no-cpu --chain $ch 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), 'it/s  nn', round(d['roofline']['avg_launch_us'],2), 'us cell', round(d['grid']['cell'],3) if isinstance(d['grid'],dict) else d['grid'])")"
done; done; done
