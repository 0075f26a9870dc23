"""Multi-GPU plumbing for the scan-sharded path.

The ICP path shards across independent scans: one process per GPU, the map replicated, every rank
registering its own scan stream -- no collective on the per-iteration data path.  The only exchange
is the map-growth step: after an update epoch each rank holds the input points it accepted into the
map (``PointDistanceMapperModule`` keep mask) and all ranks must end with the same merged set before
they rebuild their replica (``icp.setMap``) and bin the result into 20 m cells for ``RAMCellManager``
(reference: norlab_icp_mapper/Map.cpp:206-229, RAMCellManager.cpp:13-16).

``allgather_points`` is that exchange: an all-gather of the per-rank counts followed by an all-gather
of max-padded float4 blocks, concatenated in rank order (deterministic, identical on every rank).
Over RCCL (backend "nccl" on ROCm) it is two small collectives per epoch; payloads are a few MB at
most, so the cost is latency, not xGMI bandwidth.  The same code runs over gloo on CPU tensors, which
is how the tests cover it without GPUs.
"""
from typing import Dict, List, Tuple

import numpy as np
import torch
import torch.distributed as dist


def allgather_points(points: torch.Tensor, group=None) -> Tuple[torch.Tensor, List[int]]:
    """points: (n_r, C) tensor of this rank (C columns, any float dtype, CPU for gloo / CUDA for nccl).
    Returns (merged (sum n_r, C) in rank order, per-rank counts)."""
    if points.dim() != 2:
        raise ValueError("points must be (n, C)")
    world = dist.get_world_size(group)
    dev = points.device
    n_local = torch.tensor([points.shape[0]], dtype=torch.int64, device=dev)
    counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(counts, n_local, group=group)
    counts = [int(c.item()) for c in counts]
    nmax = max(counts) if counts else 0
    if nmax == 0:
        return points.new_zeros((0, points.shape[1])), counts
    pad = points.new_zeros((nmax, points.shape[1]))
    pad[: points.shape[0]] = points
    blocks = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(blocks, pad.contiguous(), group=group)
    merged = torch.cat([b[:c] for b, c in zip(blocks, counts)], dim=0)
    return merged, counts


def bin_cells(points: np.ndarray, cell_size: float = 20.0) -> Dict[str, np.ndarray]:
    """Map::unloadCells binning on the host (Map.cpp:202-222): cell id "row_col_aisle" with
    floor(coord / CELL_SIZE) per axis, points kept in input order inside each cell."""
    pts = np.ascontiguousarray(points, dtype=np.float32)
    ijk = np.floor(pts[:, :3] / np.float32(cell_size)).astype(np.int64)
    cells: Dict[str, List[int]] = {}
    for idx, (i, j, k) in enumerate(ijk):
        cells.setdefault(f"{i}_{j}_{k}", []).append(idx)
    return {cid: pts[np.array(rows)] for cid, rows in cells.items()}


class RAMCellManager:
    """norlab_icp_mapper/RAMCellManager.{h,cpp}: cell id -> cloud, saveCell overwrites, retrieveCell of
    an unknown id returns an empty cloud."""

    def __init__(self):
        self.cells: Dict[str, np.ndarray] = {}

    def getAllCellIds(self):
        return list(self.cells.keys())

    def saveCell(self, cell_id: str, cell: np.ndarray):
        self.cells[cell_id] = np.array(cell, dtype=np.float32, copy=True)

    def retrieveCell(self, cell_id: str) -> np.ndarray:
        return self.cells.get(cell_id, np.zeros((0, 4), dtype=np.float32))

    def clearAllCells(self):
        self.cells.clear()

    def merge(self, points: np.ndarray, cell_size: float = 20.0):
        """Append points to their cells (what a map-growth epoch does to cells that are not loaded)."""
        for cid, pts in bin_cells(points, cell_size).items():
            old = self.retrieveCell(cid)
            self.saveCell(cid, np.concatenate([old, pts], axis=0) if old.shape[0] else pts)
