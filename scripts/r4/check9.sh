#!/bin/bash
cd "$GRAFT_REPO_ROOT"
python -m pytest tests -m gpu -q 2>&1 | grep -a "passed\|failed\|^FAILED" | tail -8
python scripts/r4/loopback_bench.py 1 8 2>/dev/null | grep epoch
python scripts/r4/loopback_bench.py 8 8 2>/dev/null | grep epoch
python scripts/r2_chain_bench.py 1000000 100000 12 2>&1 | grep update
python scripts/e2e_bench.py 2>&1 | grep scans
python bench.py --workload config5 --scans 8 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['rank0']; print('config5 10M: scans/s', round(d['scans_per_s'],1), 'register', r['register_ms']['median'], 'epoch', r['merge_epoch_ms'])"
