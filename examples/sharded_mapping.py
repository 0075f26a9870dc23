#!/usr/bin/env python
"""BASELINE config 5 in miniature: S independent scan streams against a shared, growing map, one stream per GPU.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 examples/sharded_mapping.py \
           [--map-points 1000000] [--scan-points 100000] [--epochs 6]

Every rank holds a replica of the map (norlab_icp_mapper_amd.dist.ShardedMapper).  Per epoch each rank registers one
scan of its own stream, keeps the points farther than minDistNewPoint from the map, the kept points are all-gathered
over RCCL in rank order, merged block by block with the exact PointDistance rule, and every rank appends the identical
set.  --backend device runs the whole epoch inside libicpmi.so (its own RCCL communicator, merge and append on the
device: no accepted point crosses PCIe).  Works with one process as well (no process group)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
import norlab_icp_mapper_amd as pkg
from norlab_icp_mapper_amd.dist import ShardedMapper


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--map-points", type=int, default=1_000_000)
    ap.add_argument("--scan-points", type=int, default=100_000)
    ap.add_argument("--epochs", type=int, default=6)
    ap.add_argument("--min-dist", type=float, default=0.15)
    ap.add_argument("--backend", default="device", choices=["device", "resident", "host"],
                    help="device: exchange + merge + append inside the library (RCCL on the handle's stream); resident: the map stays in HBM, "
                         "torch.distributed moves the accepted points; host: every operator takes host arrays")
    args = ap.parse_args()
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    use_pg = "RANK" in os.environ and "MASTER_PORT" in os.environ
    torch.cuda.set_device(local)
    if use_pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
    scene = pkg.synth.make_scene(m=args.map_points, n=8)
    icp = pkg.ICPSequence(device=local, minimizer=2, max_dist=2.0, outliers=[(4, 0.85)], max_iterations=40, use_differential=1)
    backend = {"device": ShardedMapper.device_backend, "resident": ShardedMapper.resident_backend, "host": ShardedMapper.gpu_backend}[args.backend](icp)
    mapper = ShardedMapper(backend, min_dist_new_point=args.min_dist, normals_knn=10)
    mapper.set_map(scene["map"][::2])                        # start from half of the surface samples: the streams fill it in
    scans = [pkg.synth.make_scene(m=8, n=args.scan_points, seed_scan=100 + 1000 * rank + e) for e in range(args.epochs)]
    if use_pg: dist.barrier()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for e, sc in enumerate(scans):
        pose, mine, appended = mapper.epoch(sc["scan"], np.eye(4))
        dt, dr = pkg.synth.pose_error(pose, sc["T_gt"])
        if rank == 0:
            print(f"epoch {e}: rank 0 pose error {dt:.4f} m / {dr:.5f} rad, {mine} points offered by rank 0, {appended} appended by all ranks, "
                  f"map {mapper._resident_points if args.backend != 'host' else mapper.map.shape[0]}")
    if use_pg: dist.barrier()
    torch.cuda.synchronize(); secs = time.perf_counter() - t0
    if rank == 0:
        print(f"{world} stream(s) x {args.epochs} epochs in {secs:.2f} s = {world * args.epochs / secs:.1f} scans/s ({args.backend} map)")
    if use_pg: dist.destroy_process_group()


if __name__ == "__main__":
    main()
