#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6_incn; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_incremental_normals.py -m gpu -x -q 2>&1 | tail -30 | tee $O/tests.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_pins.py tests/test_gpu_map_chain.py tests/test_gpu_merge_loopback.py tests/test_gpu_insert.py -m gpu -x -q 2>&1 | tail -6 | tee -a $O/tests.txt
timeout 600 python bench.py --workload config5 --scans 6 --chain p2plane --epoch-normals-knn 10 > $O/config5_p2plane.json 2> $O/config5_p2plane.err; python -c "
import json; d=json.load(open('$O/config5_p2plane.json')); r=d['rank0']; print('p2plane epoch ms', r['merge_epoch_ms'], 'register', r['register_ms'], 'scans/s', d['scans_per_s'])"
ICPMI_SELF_DIAG=1 timeout 600 python bench.py --workload config5 --scans 3 --chain p2plane --epoch-normals-knn 10 2>&1 >/dev/null | grep "self-knn" | tail -8 | tee $O/diag.txt
