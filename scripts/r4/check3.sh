#!/bin/bash
# r4 check 3: hashed greedy merge against the oracle (loopback), the whole GPU suite, the default bench line with the new chains,
# and --workload config5 under torchrun with one rank
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r4c3; mkdir -p $O
python -m pytest tests/test_gpu_merge_loopback.py -m gpu -q 2>&1 | tail -15 > $O/pytest_merge.txt; tail -4 $O/pytest_merge.txt
python -m pytest tests -m gpu -q 2>&1 | tail -15 > $O/pytest_all.txt; tail -4 $O/pytest_all.txt
python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 600 $O/bench_default.json; echo; tail -3 $O/bench_default.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --workload config5 --scans 6 > $O/bench_config5.json 2> $O/bench_config5.err; tail -c 1500 $O/bench_config5.json; echo; tail -3 $O/bench_config5.err
