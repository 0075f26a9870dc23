"""r5, the k > 1 loop (docs/MapperConfiguration.md:174-189, knn 6): the speculative level 0 of the fused quantile selection (common.h:
ICPMI_S2_WIN; nn.hip: nnk_wg_kernel's tail; loop.hip: win_lookup) may only change WHERE a count comes from, never the count.

The NN kernel counts its distances into seven bins around the previous iteration's 16-bit prefix + below + above; when the selected
rank falls inside, the stand-alone level-0 builder has nothing to do.  Hit or miss the counts are exact: trim limit, pair count and pose
of every registration must be the bits of a handle with icpmi_config::sel_window_off = 1 -- and the window must really have served iterations
(icpmi_debug_counters slots 12 / 13: iterations served by the window / by the full histogram behind a window that missed).

The default handle is also held to the oracle here and by the rest of the suite."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def amd():
    import norlab_icp_mapper_amd as pkg
    return pkg

def run_variant(extra, k, m, n, graph, minimizer, quant, iters, checked):
    """three registrations on one handle; `extra`: engine fields of icpmi_config (sel_window_off -- an environment switch read once per process
    through r5, hence this file's subprocesses then; a per-handle field since r6)"""
    import norlab_icp_mapper_amd as pkg
    sc = pkg.synth.make_scene(m=m, n=n)
    out_type = 3 if quant == 0.5 else 4   # MedianDist{factor} / TrimmedDist{ratio}
    icp = pkg.ICPSequence(minimizer=minimizer, knn=k, max_dist=2.0, outliers=[(out_type, 3.0 if out_type == 3 else quant)], max_iterations=iters,
                          use_differential=checked, use_graph=graph, **extra)
    icp.setMap(sc["map"], sc["normals"])
    out = []
    for s in [sc["scan"], sc["scan"][::2].copy(), sc["scan"]]:
        T = np.asarray(icp(s), dtype=np.float64)
        dbg = icp.debugCounters()
        out.append(dict(T=T.tobytes().hex(), it=int(icp.stats.iterations), pairs=int(icp.stats.pairs),
                        limit=float(icp.stats.trimmed_limit).hex(), ratio=float(icp.stats.weighted_point_used_ratio).hex(),
                        win_hit=int(dbg[12]), win_miss=int(dbg[13])))
    return out


def strip(rs):
    return [{k: v for k, v in r.items() if not k.startswith("win_")} for r in rs]


# k, map points, scan points, graph, minimizer (1 = point-to-point, 2 = point-to-plane), quantile (0.5 = MedianDist), iterations, checked
CASES = [(6, 400_000, 50_000, 1, 2, 0.85, 20, 0),   # the documented chain's shape, fixed count, one graph
         (6, 400_000, 50_000, 0, 2, 0.85, 40, 1),   # checked loop, eager
         (6, 150_000, 20_011, 1, 1, 0.7, 16, 0),    # point-to-point (the matched point is gathered inside `pair`)
         (3, 150_000, 20_011, 0, 2, 0.5, 14, 0),    # MedianDist: limit = factor x median
         (16, 100_000, 10_000, 1, 2, 0.9, 12, 0)]   # sixteen distances per lane: the widest packed counts


@pytest.mark.parametrize("case", CASES)
def test_window_changes_no_bit(case):
    ref = run_variant({"sel_window_off": 1}, *case)
    assert all(r["win_hit"] == 0 and r["win_miss"] == 0 for r in ref)
    assert ref[0]["it"] > 5, "the registration must run seeded iterations"
    both = run_variant({}, *case)
    assert strip(both) == strip(ref), f"defaults differ from the switched-off handle: {both} vs {ref}"
    # the window was looked at in every iteration nnk_wg_kernel served (from iteration 2 on) ...
    # (a segment graph of a checked loop may run dead iterations behind the stop: they return before they count)
    for r in both:
        assert r["win_hit"] + r["win_miss"] == max(r["it"] - 2, 0), r
    # ... and served the settled ones (a fixed count runs long past convergence; the checked loop stops early and may see fewer)
    assert sum(r["win_hit"] for r in both) > 0, both


def test_window_is_left_alone_where_it_cannot_be_trusted(amd, small_scene):
    """Chains that may need the brute pass (unbounded maxDist: the brute kernel rewrites d2 after the NN launch) and loops past 2^21
    matches never count the window: slots 12 / 13 stay zero and the registration equals the oracle-checked default in every other test."""
    sc = small_scene
    icp = amd.ICPSequence(minimizer=2, knn=6, max_dist=float("inf"), outliers=[(4, 0.85)], max_iterations=10, use_differential=0)
    icp.setMap(sc["map"], sc["normals"])
    icp(sc["scan"])
    dbg = icp.debugCounters()
    assert int(dbg[12]) == 0 and int(dbg[13]) == 0
    assert icp.stats.iterations == 10


def test_window_registration_against_the_oracle(amd, oracle, mid_scene):
    """knn 6, TrimmedDist 0.85, point-to-plane, 20 fixed iterations: iterations, pair count, overlap and pose of the HIP loop (window on) against the oracle's loop, as everywhere else in the suite."""
    sc = mid_scene
    kw = dict(minimizer=2, knn=6, max_dist=2.0, outliers=[(4, 0.85)], max_iterations=20, use_differential=0)
    icp = amd.ICPSequence(**kw)
    icp.setMap(sc["map"], sc["normals"])
    T = icp(sc["scan"])
    dbg = icp.debugCounters()
    assert int(dbg[12]) > 0, "the window never served an iteration"
    assert int(dbg[12]) + int(dbg[13]) == 18
    oicp = oracle.OracleICP(oracle.make_config(nthreads=8, **kw))
    oicp.setMap(sc["map"], sc["normals"])
    err, T_ref = oicp(sc["scan"])
    assert err == 0
    assert icp.stats.iterations == oicp.stats.iterations == 20
    assert icp.stats.pairs == oicp.stats.pairs
    assert abs(icp.errorMinimizer.getOverlap() - oicp.stats.weighted_point_used_ratio) < 1e-6
    dt, dr = amd.synth.pose_error(T, T_ref)
    assert dt <= 1e-4 and dr <= 1e-4, (dt, dr)
