"""icpmi_config::fuse_solve (r4): three launches per iteration -- the solve of iteration i rides in the prologue of EVERY workgroup of
iteration i + 1's NN launch (csrc/nn.hip: nn1_wg_kernel<4, true, true>, csrc/solve.h), the loop state ping-pongs between two buffers,
the pair sums are fixed-point device atomics.  The same code computes the same numbers: a registration must end on the SAME BITS with
the knob on and off -- fixed-count graphs, checked loops (segment graphs and the eager run-ahead loop), error exits."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def amd():
    import norlab_icp_mapper_amd as pkg
    return pkg


CHAINS = {
    "p2p_trim": dict(minimizer=1, max_dist=2.0, outliers=[(4, 0.85)]),
    "p2plane_trim": dict(minimizer=2, max_dist=2.0, outliers=[(4, 0.85)]),
    "p2plane_median_maxdist": dict(minimizer=2, max_dist=1.5, outliers=[(1, 1.0), (3, 3.0)]),
    "p2p_no_filter": dict(minimizer=1, max_dist=1.0, outliers=[]),
    "identity": dict(minimizer=0, max_dist=2.0, outliers=[(4, 0.9)]),
}


@pytest.mark.parametrize("name", list(CHAINS))
@pytest.mark.parametrize("mode", ["fixed7", "fixed20", "checked"])
def test_fused_solve_is_bitwise_equal(amd, mid_scene, name, mode):
    import torch
    sc = mid_scene
    d = torch.from_numpy(sc["scan"]).cuda()
    res = []
    for fuse in (0, 1):
        kw = dict(CHAINS[name], fuse_solve=fuse)
        if mode == "checked":
            kw.update(max_iterations=40, use_differential=1)
        icp = amd.ICPSequence(**kw)
        assert icp.setMap(sc["map"], sc["normals"])
        for _ in range(2):   # the second call replays the cached graphs
            T = icp.registerDev(d.data_ptr(), d.shape[0], fixed_iterations={"fixed7": 7, "fixed20": 20, "checked": 0}[mode])
        res.append((T.copy(), icp.stats.iterations, icp.stats.stop_reason, icp.stats.pairs, icp.stats.trimmed_limit, icp.stats.weighted_point_used_ratio))
    a, b = res
    assert np.array_equal(a[0], b[0]), (name, mode, np.abs(a[0] - b[0]).max())
    assert a[1:] == b[1:], (a[1:], b[1:])
    if mode == "checked" and CHAINS[name]["minimizer"] != 0:
        assert a[1] > 2


def test_fused_solve_eager_loop_and_errors(amd, mid_scene, monkeypatch):
    """ICPMI_SEG=0 is read once per process, so the eager run-ahead loop is reached through use_graph = 0; a Bound checker that throws
    and a reading with nothing to match must surface the same error class with the knob on."""
    sc = mid_scene
    outs = []
    for fuse in (0, 1):
        icp = amd.ICPSequence(minimizer=2, max_dist=2.0, outliers=[(4, 0.85)], max_iterations=40, use_differential=1, use_graph=0, fuse_solve=fuse)
        assert icp.setMap(sc["map"], sc["normals"])
        T = icp(sc["scan"])
        outs.append((T.copy(), icp.stats.iterations, icp.stats.pairs))
        icp.setConfig(minimizer=2, max_dist=2.0, outliers=[(4, 0.85)], max_iterations=40, use_bound=1, max_rot_norm=1e-4, max_trans_norm=1e-4, fuse_solve=fuse)
        with pytest.raises(amd.ConvergenceError):
            icp(sc["scan"])
        icp.setConfig(minimizer=1, max_dist=0.5, outliers=[], max_iterations=10, fuse_solve=fuse)
        far = sc["scan"].copy(); far[:, :3] += 300.0
        with pytest.raises(amd.ConvergenceError):
            icp(far)
    assert np.array_equal(outs[0][0], outs[1][0]) and outs[0][1:] == outs[1][1:]
