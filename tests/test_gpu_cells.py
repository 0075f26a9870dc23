"""icpmi_staged_bin_cells (csrc/cells.hip): the merged set of a map-growth epoch binned into the mapper's cubic cells on the device
and appended to the handle's cell log -- against the reference's per-point loop restated in numpy (Map.cpp:206-229: cell =
floor(coordinate / CELL_SIZE) per axis, Map.cpp:472-480; RAMCellManager.cpp:13-16 keeps id -> cloud): the same cells in the order of
their first point, the same points in merged order inside every cell, bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

MIN_DIST = 0.3


@pytest.fixture(scope="module")
def amd():
    import norlab_icp_mapper_amd as pkg
    return pkg


def reference_binning(points, cell_size):
    """Map::unloadCells' loop: the cell of every point, cells in the order they are first met, merged order inside a cell."""
    ijk = np.floor(points[:, :3] / np.float32(cell_size)).astype(np.int64)   # float32 division, as toGridCoordinate
    order, cells = [], {}
    for i, key in enumerate(map(tuple, ijk)):
        if key not in cells:
            cells[key] = []
            order.append(key)
        cells[key].append(i)
    return order, cells


def _epoch(amd, sc, monkeypatch, ranks, shift):
    monkeypatch.setenv("ICPMI_COMM_LOOPBACK", str(ranks))
    monkeypatch.setenv("ICPMI_COMM_LOOPBACK_SHIFT", repr(shift))
    monkeypatch.setenv("ICPMI_COMM_LOOPBACK_RAGGED", "1")
    icp = amd.ICPSequence(minimizer=2, max_dist=2.0, outliers=[(4, 0.85)], max_iterations=30, use_differential=1)
    assert icp.setMap(sc["map"][::2].copy(), sc["normals"][::2].copy())
    icp.commInit(icp.commUniqueId(), 1, 0)
    return icp


def _one_epoch(icp, sc):
    corr = icp.registerWithPrior(sc["scan"], np.eye(4, dtype=np.float32))
    _, appended, _, merged = icp.stagedMergeAllGather(corr, MIN_DIST, normals_knn=0, return_merged=True)
    assert appended == merged.shape[0]
    return merged


@pytest.mark.parametrize("in_epoch", [False, True])
@pytest.mark.parametrize("cell_size,ranks,shift", [(20.0, 2, 0.5), (5.0, 4, 0.45), (3.0, 3, 0.6)])
def test_cells_of_an_epoch_match_the_reference_loop(amd, mid_scene, monkeypatch, cell_size, ranks, shift, in_epoch):
    icp = _epoch(amd, mid_scene, monkeypatch, ranks, shift)
    if in_epoch:
        icp.cellLogConfigure(cell_size)           # the epoch enqueues the binning behind its merge; stagedBinCells collects the table
    base = 0
    for epoch in range(3):                        # the log grows epoch after epoch; later epochs add little (the map has the points)
        merged = _one_epoch(icp, mid_scene)
        assert epoch > 0 or merged.shape[0] > 1000
        order, cells = reference_binning(merged, cell_size)
        ijk, off, cnt = icp.stagedBinCells(cell_size)
        if merged.shape[0] == 0:
            assert ijk.shape[0] == 0
            continue
        if epoch == 0:
            first_cells = len(order)
        assert [tuple(r) for r in ijk.tolist()] == order
        assert cnt.tolist() == [len(cells[k]) for k in order]
        assert off.tolist() == (base + np.concatenate([[0], np.cumsum(cnt)[:-1]])).tolist()
        for key, o, c in zip(order, off, cnt):
            assert np.array_equal(icp.cellLogRead(o, c), merged[cells[key]])
        base += merged.shape[0]
        assert icp.cellLogSize() == base
        with pytest.raises(Exception, match="already"):
            icp.stagedBinCells(cell_size)         # once per epoch
    if cell_size < 20.0:
        assert first_cells > 64                    # (more cells than the first guess of the sort's key bits: the second attempt ran)
    icp.cellLogClear()
    assert icp.cellLogSize() == 0


def test_capacity_and_cell_limits(amd, mid_scene, monkeypatch):
    icp = _epoch(amd, mid_scene, monkeypatch, 2, 0.5)
    merged = _one_epoch(icp, mid_scene)
    order, cells = reference_binning(merged, 8.0)
    with pytest.raises(Exception, match="capacity"):
        icp.stagedBinCells(8.0, capacity=len(order) - 1)
    assert icp.cellLogSize() == 0                 # nothing appended by the refused call
    ijk, off, cnt = icp.stagedBinCells(8.0, capacity=len(order))
    assert [tuple(r) for r in ijk.tolist()] == order and int(cnt.sum()) == merged.shape[0]
    # an edge that makes one epoch touch more than 4096 cells: refused, the log untouched, the host path (merged points) still there
    icp2 = _epoch(amd, mid_scene, monkeypatch, 2, 0.5)
    merged = _one_epoch(icp2, mid_scene)
    assert len(reference_binning(merged, 0.3)[0]) > 4096
    with pytest.raises(Exception, match="4096"):
        icp2.stagedBinCells(0.3)
    assert icp2.cellLogSize() == 0
    assert np.array_equal(icp2.stagedMergedPoints(), merged)
    ijk, off, cnt = icp2.stagedBinCells(20.0)      # (the refused call left the epoch's set unbinned)
    assert int(cnt.sum()) == merged.shape[0]


def test_no_epoch_no_cells(amd, mid_scene):
    icp = amd.ICPSequence(minimizer=1, max_dist=2.0, outliers=[(4, 0.85)], max_iterations=5)
    assert icp.setMap(mid_scene["map"][::4].copy())
    ijk, off, cnt = icp.stagedBinCells(20.0)
    assert ijk.shape == (0, 3) and icp.cellLogSize() == 0
