"""CPU tests of the N > 1 path with the gloo backend, world_size 2: the map-growth
all-gather must leave every rank with the identical merged point set (rank order), and the merged
set must bin into identical 20 m cells (RAMCellManager semantics)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from norlab_icp_mapper_amd.dist import allgather_points, RAMCellManager
    rng = np.random.default_rng(100 + rank)
    n = [137, 0, 64][rank % 3] if rank else 251            # ragged, rank 1 contributes nothing
    pts = np.ones((n, 4), dtype=np.float32)
    pts[:, :3] = rng.uniform(-70, 70, (n, 3)).astype(np.float32)
    merged, counts = allgather_points(torch.from_numpy(pts))
    # second epoch with every rank empty
    empty, counts2 = allgather_points(torch.zeros((0, 4), dtype=torch.float32))
    cm = RAMCellManager()
    cm.merge(merged.numpy())
    cm.merge(merged.numpy()[:10])                           # a later epoch appends to existing cells
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), merged=merged.numpy(), counts=np.array(counts), mine=pts,
             empty=empty.numpy(), counts2=np.array(counts2), ids=np.array(sorted(cm.getAllCellIds())),
             sizes=np.array([cm.retrieveCell(c).shape[0] for c in sorted(cm.getAllCellIds())]))
    dist.barrier()
    dist.destroy_process_group()


def test_allgather_merge_is_identical_on_all_ranks(tmp_path):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r = [np.load(os.path.join(tmp_path, f"rank{k}.npz")) for k in range(world)]
    assert r[0]["counts"].tolist() == [251, 0] and r[1]["counts"].tolist() == [251, 0]
    assert np.array_equal(r[0]["merged"], r[1]["merged"])
    # rank order: rank 0's block first, bit for bit
    assert np.array_equal(r[0]["merged"][:251], r[0]["mine"])
    assert r[0]["empty"].shape == (0, 4) and r[0]["counts2"].tolist() == [0, 0]
    assert np.array_equal(r[0]["ids"], r[1]["ids"]) and np.array_equal(r[0]["sizes"], r[1]["sizes"])
    assert r[0]["sizes"].sum() == 251 + 10


def test_cell_binning_matches_reference_rule():
    from norlab_icp_mapper_amd.dist import bin_cells, RAMCellManager
    pts = np.array([[-0.5, 19.99, 20.0, 1], [0.0, 0.0, 0.0, 1], [-20.0, -20.01, 39.9, 1], [19.9, 0.1, 0.2, 1]], dtype=np.float32)
    cells = bin_cells(pts)
    assert sorted(cells) == ["-1_-2_1", "-1_0_1", "0_0_0"]
    assert cells["0_0_0"].shape[0] == 2 and np.array_equal(cells["0_0_0"][0], pts[1])  # input order kept
    cm = RAMCellManager()
    assert cm.retrieveCell("5_5_5").shape == (0, 4)          # unknown id -> empty cloud
    cm.saveCell("a", pts[:2]); cm.saveCell("a", pts[2:])     # saveCell overwrites
    assert np.array_equal(cm.retrieveCell("a"), pts[2:])
    cm.clearAllCells()
    assert cm.getAllCellIds() == []


def _mapper_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_bindings as ob
    from norlab_icp_mapper_amd import synth
    from norlab_icp_mapper_amd.dist import ShardedMapper

    # CPU backend for the test: the oracle's operators (test infrastructure) behind the same four hooks
    oicp = ob.OracleICP(ob.make_config(minimizer=1, max_dist=2.0, outliers=[(4, 0.85)], max_iterations=15, nthreads=2))

    class Backend:
        @staticmethod
        def register(scan):
            err, T = oicp(scan)
            assert err == 0
            return T
        set_map = staticmethod(lambda cloud, normals: oicp.setMap(cloud, normals))
        keep = staticmethod(lambda m, c, d: ob.point_distance_keep(m, c, d, nthreads=2))
        normals = staticmethod(lambda cloud, knn: ob.surface_normals(cloud, knn))

    sc = synth.make_scene(m=6000, n=1500, seed_scan=43 + 1000 * rank)      # one scan stream per rank, shared map
    mapper = ShardedMapper(Backend, min_dist_new_point=0.5)
    mapper.set_map(sc["map"])
    sizes, poses = [mapper.map.shape[0]], []
    for epoch in range(2):
        pose, mine, appended = mapper.epoch(sc["scan"] if epoch == 0 else sc["scan"][::2], np.eye(4))
        sizes.append(mapper.map.shape[0]); poses.append(pose)
    np.savez(os.path.join(out_dir, f"mapper{rank}.npz"), map=mapper.map, sizes=np.array(sizes), pose=poses[0], T_gt=sc["T_gt"], base=sc["map"].shape[0])
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_mapper_replicas_stay_identical(tmp_path):
    """World size 2 over gloo: after every epoch the ranks hold the identical grown map; each rank's pose
    is its own registration result."""
    world = 2
    mp.spawn(_mapper_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [np.load(os.path.join(str(tmp_path), f"mapper{k}.npz")) for k in range(world)]
    assert np.array_equal(r[0]["map"], r[1]["map"])
    assert np.array_equal(r[0]["sizes"], r[1]["sizes"])
    assert r[0]["sizes"][1] > r[0]["sizes"][0]                 # the map grew in the first epoch
    assert r[0]["sizes"][2] - r[0]["sizes"][1] < r[0]["sizes"][1] - r[0]["sizes"][0]  # the second epoch sees mostly known surface
    # PointDistanceMapperModule's invariant holds on the merged map as it would after one mapper had taken the scans in rank
    # order: every appended point is at least minDistNewPoint from the old map, and a point of a later rank at least that
    # far from the points accepted from the earlier ranks
    import oracle_bindings as ob
    m = r[0]["map"]; base = int(r[0]["base"]); first = int(r[0]["sizes"][1])
    grown = m[base:first]
    assert ob.point_distance_keep(m[:base], grown, 0.5).all()
    ids, d2 = ob.knn(grown, grown, k=2)                          # nearest OTHER appended point
    close = d2[:, 1] < 0.25
    # pairs closer than minDist can only come from ONE rank's scan (a mapper appends all of a scan's accepted points at once)
    assert close.sum() < grown.shape[0]
    for k in range(world):
        assert np.isfinite(r[k]["pose"]).all()
    assert not np.array_equal(r[0]["pose"], r[1]["pose"])      # every rank registered its own scan


def _resident_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_bindings as ob
    from norlab_icp_mapper_amd import synth
    from norlab_icp_mapper_amd.dist import ShardedMapper

    # CPU stand-in of the RESIDENT backend (ShardedMapper.resident_backend): the replica lives behind the backend, the
    # mapper only sees scans, masks and accepted points -- same hooks, the oracle's operators behind them
    oicp = ob.OracleICP(ob.make_config(minimizer=1, max_dist=2.0, outliers=[(4, 0.85)], max_iterations=15, nthreads=2))

    class Resident:
        resident = True
        cloud = np.zeros((0, 4), np.float32)
        staged = None

        @classmethod
        def set_map(cls, cloud, normals):
            cls.cloud = np.ascontiguousarray(cloud, dtype=np.float32)
            oicp.setMap(cls.cloud, normals)

        @classmethod
        def register_prior(cls, scan, prior):
            cls.staged = ob.transform(prior, scan)
            if cls.cloud.shape[0] == 0:
                return np.eye(4, dtype=np.float32)
            err, T = oicp(cls.staged)
            assert err == 0
            return T

        @classmethod
        def staged_keep(cls, correction, d):
            placed = ob.transform(correction, cls.staged)
            if cls.cloud.shape[0] == 0:
                return np.ones(placed.shape[0], bool), placed
            return ob.point_distance_keep(cls.cloud, placed, d, nthreads=2), placed

        @classmethod
        def append(cls, pts, knn):
            cls.set_map(np.concatenate([cls.cloud, pts]), None)

        normals = staticmethod(lambda cloud, knn: ob.surface_normals(cloud, knn))
        keep = staticmethod(lambda m, c, d: ob.point_distance_keep(m, c, d, nthreads=2))
        get_map = classmethod(lambda cls: cls.cloud)

    sc = synth.make_scene(m=6000, n=1500, seed_scan=43 + 1000 * rank)
    mapper = ShardedMapper(Resident, min_dist_new_point=0.5)
    mapper.set_map(sc["map"])
    sizes = [mapper.get_map().shape[0]]
    for epoch in range(2):
        pose, mine, appended = mapper.epoch(sc["scan"] if epoch == 0 else sc["scan"][::2], np.eye(4))
        sizes.append(mapper.get_map().shape[0])
        assert mapper._resident_points == sizes[-1]
    np.savez(os.path.join(out_dir, f"resident{rank}.npz"), map=mapper.get_map(), sizes=np.array(sizes))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_mapper_resident_flow_replicas_stay_identical(tmp_path):
    """The resident flow of the epoch (register_prior -> staged keep -> exchange -> append) over gloo with two ranks and
    a CPU stand-in behind the resident hooks: identical replicas, and the same growth as the host-array flow above."""
    world = 2
    port = _free_port()
    mp.spawn(_resident_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    mp.spawn(_mapper_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [np.load(os.path.join(str(tmp_path), f"resident{k}.npz")) for k in range(world)]
    h = np.load(os.path.join(str(tmp_path), "mapper0.npz"))
    assert np.array_equal(r[0]["map"], r[1]["map"]) and np.array_equal(r[0]["sizes"], r[1]["sizes"])
    assert r[0]["sizes"][1] > r[0]["sizes"][0]
    # same decisions as the host-array flow (the scan is placed by two exact float transforms here, by one numpy product there)
    assert abs(int(r[0]["sizes"][2]) - int(h["sizes"][2])) <= 3


def _uneven_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=__import__("datetime").timedelta(seconds=60))
    import oracle_bindings as ob
    from norlab_icp_mapper_amd import synth
    from norlab_icp_mapper_amd.dist import ShardedMapper

    oicp = ob.OracleICP(ob.make_config(minimizer=1, max_dist=2.0, outliers=[(4, 0.85)], max_iterations=10, nthreads=2))
    fail_next = {"on": False}

    class Backend:
        @staticmethod
        def register(scan):
            if fail_next["on"]:
                raise RuntimeError("ConvergenceError: no point to minimize")   # what an ordinary failed registration raises
            err, T = oicp(scan)
            assert err == 0
            return T
        set_map = staticmethod(lambda cloud, normals: oicp.setMap(cloud, normals))
        keep = staticmethod(lambda m, c, d: ob.point_distance_keep(m, c, d, nthreads=2))
        normals = staticmethod(lambda cloud, knn: ob.surface_normals(cloud, knn))

    sc = synth.make_scene(m=5000, n=1200, seed_scan=43 + 1000 * rank)
    mapper = ShardedMapper(Backend, min_dist_new_point=0.5)
    mapper.set_map(sc["map"])
    sizes, raised = [mapper.map.shape[0]], []
    empty = np.zeros((0, 4), dtype=np.float32)
    # epoch 0: both ranks have a scan; epoch 1: rank 1 has run out of scans (14 scans over 2, 4 or 8 ranks leave such a tail);
    # epoch 2: rank 0's registration fails; epoch 3: both fine again -- nobody may hang, replicas must stay identical
    plan = [(sc["scan"], sc["scan"]), (sc["scan"][::2], empty), (sc["scan"][1::2], sc["scan"][::3]), (sc["scan"][::5], sc["scan"][1::3])]
    for e, scans in enumerate(plan):
        fail_next["on"] = (e == 2 and rank == 0)
        try:
            mapper.epoch(scans[rank], np.eye(4))
            raised.append(0)
        except RuntimeError as ex:
            assert "no point to minimize" in str(ex)
            raised.append(1)
        sizes.append(mapper.map.shape[0])
    np.savez(os.path.join(out_dir, f"uneven{rank}.npz"), map=mapper.map, sizes=np.array(sizes), raised=np.array(raised))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_mapper_uneven_scan_counts_and_failed_registration(tmp_path):
    """a rank with no scan left, or whose registration throws, must still take part in the epoch's exchange;
    its own error surfaces afterwards and the replicas stay identical."""
    world = 2
    mp.spawn(_uneven_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [np.load(os.path.join(str(tmp_path), f"uneven{k}.npz")) for k in range(world)]
    assert np.array_equal(r[0]["map"], r[1]["map"])
    assert np.array_equal(r[0]["sizes"], r[1]["sizes"])
    assert list(r[0]["raised"]) == [0, 0, 1, 0] and list(r[1]["raised"]) == [0, 0, 0, 0]
    assert r[0]["sizes"][2] > r[0]["sizes"][1] or r[0]["sizes"][1] > r[0]["sizes"][0]
