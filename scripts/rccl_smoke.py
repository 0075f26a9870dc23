#!/usr/bin/env python
"""RCCL plumbing on whatever GPUs are visible (world size = number of processes torchrun started, 1 is fine):
process group over backend nccl, barrier, max all-reduce (what bench.py does) and allgather_points on device tensors."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local)
dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
from norlab_icp_mapper_amd.dist import allgather_points
dist.barrier()
t = torch.tensor([1.0 + rank], dtype=torch.float64, device="cuda"); dist.all_reduce(t, op=dist.ReduceOp.MAX)
pts = torch.rand((100 + 10 * rank, 4), device="cuda")
merged, counts = allgather_points(pts)
assert merged.shape[0] == sum(counts) and merged.is_cuda
print(f"rank {rank}/{world}: all_reduce max {t.item()}, allgather {counts} -> {tuple(merged.shape)} ok")
dist.barrier(); dist.destroy_process_group()
