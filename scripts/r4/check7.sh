#!/bin/bash
cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_insert.py tests/test_gpu_merge_loopback.py tests/test_gpu_map_chain.py -m gpu -q 2>&1 | grep -a "passed\|failed\|Error\|assert\|^E " | tail -20
python -m pytest tests -m gpu -q 2>&1 | grep -a "passed\|failed\|^FAILED" | tail -8
python bench.py --workload config5 --scans 8 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['rank0']; print('config5 10M: scans/s', round(d['scans_per_s'],1), 'register', r['register_ms']['median'], 'epoch', r['merge_epoch_ms'])"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=gpurun_out/r4c7
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o t -- python $R/bench.py --workload config5 --scans 8 > /dev/null 2>&1
f=$(find $R/$O/prof -name "*kernel_stats.csv" | head -1); python $R/scripts/kstats.py $f 2>/dev/null | head -12
find $R/$O -name "*kernel_trace.csv" -delete; find $R/$O -name "*agent_info.csv" -delete
