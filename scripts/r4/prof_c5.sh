#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r4c5prof; mkdir -p $O
python bench.py --workload config5 --scans 8 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['rank0']; print('config5 10M: scans/s', round(d['scans_per_s'],1), 'register', r['register_ms']['median'], 'epoch', r['merge_epoch_ms'], 'accepted', r['accepted_per_scan_this_rank'][:4])"
python bench.py --workload config5 --scans 8 --map-points 1000000 --scale 1.0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['rank0']; print('config5 1M: scans/s', round(d['scans_per_s'],1), 'register', r['register_ms']['median'], 'epoch', r['merge_epoch_ms'], 'accepted', r['accepted_per_scan_this_rank'][:4])"
ICPMI_INSERT=0 python bench.py --workload config5 --scans 8 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['rank0']; print('config5 10M INSERT=0: scans/s', round(d['scans_per_s'],1), 'register', r['register_ms']['median'], 'epoch', r['merge_epoch_ms'])"
ICPMI_INSERT=0 python bench.py --workload config5 --scans 8 --map-points 1000000 --scale 1.0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['rank0']; print('config5 1M INSERT=0: scans/s', round(d['scans_per_s'],1), 'register', r['register_ms']['median'], 'epoch', r['merge_epoch_ms'])"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o t -- python $R/bench.py --workload config5 --scans 8 > /dev/null 2>&1
f=$(find $R/$O/prof -name "*kernel_stats.csv" | head -1); python $R/scripts/kstats.py $f 2>/dev/null | head -40
find $R/$O -name "*kernel_trace.csv" -delete; find $R/$O -name "*agent_info.csv" -delete
