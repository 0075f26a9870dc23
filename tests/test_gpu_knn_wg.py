"""nnk_wg_kernel (csrc/nn.hip: the loop's k > 1 matcher of the seeded launches) against nnk_ml_kernel: both are EXACT k-nearest
searches with the same (d^2, index) keys, and the loop's sums run in a fixed order -- so a registration must land on the same
bits whichever of the two served an iteration.  icpmi_config::knn_wg_from chooses per handle (an environment switch read once per
process through r5): < 0 = nnk_ml_kernel everywhere, 0 = the default (nnk_wg from iteration 2), 1 = nnk_wg from the first,
UNSEEDED launch (own-row pass, list overflows and repeated passes), 2 = from the launch whose seeds moved far (wide-seed pass at
level 0)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def run_variant(knn_wg_from, k, m, n, graph):
    import norlab_icp_mapper_amd as pkg
    sc = pkg.synth.make_scene(m=m, n=n)
    icp = pkg.ICPSequence(minimizer=2, knn=k, max_dist=2.0, outliers=[(4, 0.85)], max_iterations=12, use_differential=0, use_graph=graph,
                          knn_wg_from=knn_wg_from)
    icp.setMap(sc["map"], sc["normals"])
    out = []
    for rep in range(2):
        T = np.asarray(icp(sc["scan"]), dtype=np.float64)
        out.append(dict(T=T.tobytes().hex(), it=int(icp.stats.iterations), pairs=int(icp.stats.pairs),
                        limit=float(icp.stats.trimmed_limit).hex(), ratio=float(icp.stats.weighted_point_used_ratio).hex()))
    return out


@pytest.mark.parametrize("k,m,n,graph", [(6, 400_000, 50_000, 0), (6, 400_000, 50_000, 1), (3, 150_000, 20_011, 0), (8, 60_000, 5_000, 0), (10, 200_000, 20_000, 0), (16, 100_000, 10_000, 1)])
def test_wg_matcher_lands_on_the_same_bits(k, m, n, graph):
    ref = run_variant(-1, k, m, n, graph)
    assert ref[0]["it"] > 3, "the registration must run seeded iterations"
    for frm in (0, 3, 2, 1):
        got = run_variant(frm, k, m, n, graph)
        assert got == ref, f"knn_wg_from={frm} differs from nnk_ml_kernel: {got} vs {ref}"
