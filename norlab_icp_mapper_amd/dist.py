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


class ShardedMapper:
    """One rank of the scan-sharded mapping loop of SURVEY.md 8(e) / BASELINE config 5.

    Every rank holds a replica of the map and its own scan stream.  One epoch = every rank registers one scan
    against the shared map (`Mapper::processInput`, Mapper.cpp:194-238, with the prior applied first), keeps the
    points that are at least `min_dist_new_point` away from the map (`PointDistanceMapperModule`, .cpp:28-50), the
    accepted points are all-gathered in rank order and merged block by block: block r keeps the points that are at
    least `min_dist_new_point` from the points accepted from ranks < r (the exact PointDistance rule again: what one
    mapper would have appended had it processed the scans in rank order -- identical on all ranks), appended, and
    every rank rebuilds its replica with the identical cloud (`icp.setMap`, Map.cpp:528).  No collective touches the
    per-iteration path.

    `backend` provides the operators; three are built in:
        gpu_backend      host arrays between the C-ABI operators, torch.distributed moves the accepted points
        resident_backend the map stays in HBM, torch.distributed moves the accepted points
        device_backend   the whole epoch inside the library: RCCL all-gather on the handle's stream, merge and append on
                         the device (`icpmi_staged_merge_allgather`) -- no accepted point crosses PCIe
    (and the tests inject the CPU oracle's operators over gloo).
    """

    def __init__(self, backend, min_dist_new_point=0.15, normals_knn=0, group=None):
        self.backend = backend
        self.min_dist = float(min_dist_new_point)
        self.normals_knn = int(normals_knn)
        self.group = group
        self.map = np.zeros((0, 4), dtype=np.float32)
        self.normals = None
        self.pose = np.eye(4, dtype=np.float32)
        self._resident_points = 0

    @staticmethod
    def gpu_backend(icp):
        class _B:
            register = staticmethod(lambda scan: icp(scan))
            set_map = staticmethod(lambda cloud, normals: icp.setMap(cloud, normals))
            keep = staticmethod(lambda m, c, d: icp.pointDistanceKeep(m, c, d))
            normals = staticmethod(lambda cloud, knn: icp.surfaceNormals(cloud, knn))
            transform = staticmethod(lambda T, cloud: icp.transform(T, cloud))
        return _B

    @staticmethod
    def resident_backend(icp):
        """The same epoch with the map resident in HBM (`ICPSequence` over the C ABI): the scan is uploaded once
        (`icpmi_register_prior`), the keep mask is decided against the resident map without touching it
        (`icpmi_staged_point_distance_keep`), and what all ranks accepted is appended on the device
        (`icpmi_map_update_point_distance` with min_dist 0 keeps every point) -- per epoch only the scan and the accepted
        points cross PCIe, not the map (160 MB at 10 M points)."""
        class _R:
            resident = True
            register_prior = staticmethod(lambda scan, prior: icp.registerWithPrior(scan, prior))
            staged_keep = staticmethod(lambda correction, d: icp.stagedPointDistanceKeep(correction, d))
            append = staticmethod(lambda pts, knn: icp.mapUpdatePointDistance(pts, 0.0, normals_knn=knn))
            set_map = staticmethod(lambda cloud, normals: icp.setMap(cloud, normals))
            normals = staticmethod(lambda cloud, knn: icp.surfaceNormals(cloud, knn))
            keep = staticmethod(lambda m, c, d: icp.pointDistanceKeep(m, c, d))
            get_map = staticmethod(lambda: icp.getMap())
        return _R

    @staticmethod
    def device_backend(icp, group=None):
        """The epoch inside libicpmi.so: the library owns an RCCL communicator (created here from an id that rank 0 makes and
        torch.distributed hands round once) and runs the exchange, the rank-ordered merge and the append on the device."""
        world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        rank = dist.get_rank(group) if world > 1 else 0
        if world > 1:
            box = [icp.commUniqueId() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0, group=group)
            icp.commInit(box[0], world, rank)

        class _D:
            resident = True
            device_merge = True
            register_prior = staticmethod(lambda scan, prior: icp.registerWithPrior(scan, prior))
            merge = staticmethod(lambda correction, d, knn: icp.stagedMergeAllGather(correction, d, normals_knn=knn))
            set_map = staticmethod(lambda cloud, normals: icp.setMap(cloud, normals))
            normals = staticmethod(lambda cloud, knn: icp.surfaceNormals(cloud, knn))
            get_map = staticmethod(lambda: icp.getMap())
        return _D

    def _merge_blocks(self, merged, counts):
        """rank-ordered exact merge: block r keeps what is at least min_dist from the points accepted from ranks < r"""
        if len(counts) <= 1 or not (self.min_dist > 0):
            return merged
        out, off = None, 0
        for c in counts:
            blk = merged[off:off + c]; off += c
            if c == 0:
                continue
            if out is None:
                out = blk
            else:
                out = np.concatenate([out, blk[self.backend.keep(out, np.ascontiguousarray(blk), self.min_dist)]], axis=0)
        return merged[:0] if out is None else np.ascontiguousarray(out)

    def get_map(self):
        """The replica's map (downloaded from the device in resident mode)."""
        return self.backend.get_map() if getattr(self.backend, "resident", False) else self.map

    def _epoch_resident(self, scan, prior):
        # The epoch is a collective: a rank with no scan left, or whose registration fails (an ordinary ICP ConvergenceError), must
        # still take part in the exchange -- empty-handed -- or its peers wait in the all-gather forever.  Its own error is raised
        # once every rank is through the epoch.
        failure, correction = None, None
        if scan.shape[0] > 0:
            try:
                correction = self.backend.register_prior(scan, prior)       # identity while there is no map; stages the scan
            except Exception as e:  # noqa: BLE001 -- re-raised below, after the collective
                failure = e
        if correction is not None:
            self.pose = (np.asarray(correction, dtype=np.float64) @ np.asarray(prior, dtype=np.float64)).astype(np.float32)
        if getattr(self.backend, "device_merge", False):
            mine_n, appended, m1 = self.backend.merge(correction, self.min_dist, self.normals_knn)   # None: contributes nothing
            self._resident_points = m1
            if failure is not None:
                raise failure
            return self.pose, mine_n, appended
        if correction is not None:
            mask, placed = self.backend.staged_keep(correction, self.min_dist)
            mine = placed[mask]
        else:
            mine = np.zeros((0, 4), dtype=np.float32)
        if dist.is_available() and dist.is_initialized():
            t = torch.from_numpy(np.ascontiguousarray(mine))
            if dist.get_backend(self.group) == "nccl":
                t = t.cuda()
            merged, counts = allgather_points(t, group=self.group)
            merged = self._merge_blocks(merged.cpu().numpy(), counts)
        else:
            merged = mine
        if merged.shape[0]:
            self.backend.append(np.ascontiguousarray(merged), self.normals_knn)
            self._resident_points += int(merged.shape[0])
        if failure is not None:
            raise failure
        return self.pose, int(mine.shape[0]), int(merged.shape[0])

    def set_map(self, cloud, normals=None):
        self._resident_points = int(np.asarray(cloud).shape[0])
        self.map = np.ascontiguousarray(cloud, dtype=np.float32)
        if normals is None and self.normals_knn > 0 and self.map.shape[0]:
            normals = self.backend.normals(self.map, self.normals_knn)
        self.normals = None if normals is None else np.ascontiguousarray(normals, dtype=np.float32)
        self.backend.set_map(self.map, self.normals)

    @staticmethod
    def _apply(T, cloud):
        T = np.asarray(T, dtype=np.float32)
        out = cloud.copy()
        out[:, :3] = cloud[:, :3] @ T[:3, :3].T + T[:3, 3]
        return out

    def epoch(self, scan, prior):
        """scan: (n, 4) in the sensor frame; prior: 4x4 estimated pose.  Returns (corrected pose, number of points
        this rank contributed, number of points appended to the shared map)."""
        scan = np.ascontiguousarray(scan, dtype=np.float32)
        if getattr(self.backend, "resident", False):
            return self._epoch_resident(scan, np.asarray(prior, dtype=np.float32))
        # RigidTransformation::compute of the backend when it has one (the device's fmaf chain, bit for bit what the resident path
        # applies), numpy otherwise.  The reference moves the scan TWICE -- by the prior (Mapper.cpp:197), then the registered
        # cloud by the correction (:221) -- not once by their product: the same two steps here.
        move = getattr(self.backend, "transform", None) or self._apply
        failure = None
        mine = np.zeros((0, 4), dtype=np.float32)                              # an empty scan / a failed registration contributes nothing
        if scan.shape[0] > 0:
            try:
                in_map = move(np.asarray(prior, dtype=np.float32), scan)             # Mapper.cpp:197
                correction = self.backend.register(in_map) if self.map.shape[0] else np.eye(4, dtype=np.float32)
                self.pose = (np.asarray(correction, dtype=np.float64) @ np.asarray(prior, dtype=np.float64)).astype(np.float32)  # :215
                placed = move(np.asarray(correction, dtype=np.float32), in_map)      # :221
                mask = self.backend.keep(self.map, placed, self.min_dist) if self.map.shape[0] else np.ones(placed.shape[0], bool)
                mine = placed[mask]
            except Exception as e:  # noqa: BLE001 -- the exchange below is a collective: take part, then re-raise
                failure = e
        if dist.is_available() and dist.is_initialized():
            t = torch.from_numpy(np.ascontiguousarray(mine))
            if dist.get_backend(self.group) == "nccl":            # RCCL moves device tensors
                t = t.cuda()
            merged, counts = allgather_points(t, group=self.group)
            merged = self._merge_blocks(merged.cpu().numpy(), counts)      # points of different ranks closer than min_dist
        else:
            merged = mine
        if merged.shape[0]:
            new_map = np.concatenate([self.map, merged], axis=0)
            self.set_map(new_map, None)
        self._resident_points = int(self.map.shape[0])
        if failure is not None:
            raise failure
        return self.pose, int(mine.shape[0]), int(merged.shape[0])
