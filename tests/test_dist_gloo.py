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
