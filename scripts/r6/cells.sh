#!/bin/bash
# r6: device cell binning -- parity tests, the sharded example tests, config 5 with the binning (and the p2plane variant) inside the epoch
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6_cells; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_cells.py tests/test_gpu_merge_loopback.py -m gpu -x -q 2>&1 | tail -15 | tee $O/tests.txt
timeout 900 python -m pytest tests/test_host_cpp.py -m gpu -x -q -k sharded 2>&1 | tail -15 | tee -a $O/tests.txt
timeout 600 python bench.py --workload config5 --scans 8 > $O/config5.json 2> $O/config5.err; tail -c 1500 $O/config5.json; echo
timeout 600 python bench.py --workload config5 --scans 3 --chain p2plane --epoch-normals-knn 10 > $O/config5_p2plane.json 2> $O/config5_p2plane.err; tail -c 1500 $O/config5_p2plane.json; echo
ICPMI_EPOCH_TIMING=1 timeout 600 python bench.py --workload config5 --scans 3 --chain p2plane --epoch-normals-knn 10 2>&1 >/dev/null | grep "icpmi epoch" | tail -12 | tee $O/epoch_timing_p2plane.txt
