#!/usr/bin/env python
"""Soak test: random maps / readings / chains through the C ABI, looking for hangs, device errors and
non-finite poses.  Every N-th case is cross-checked against the oracle."""
import sys, os, time, math
_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, _ROOT)
sys.path.insert(0, os.path.join(_ROOT, "tests"))
import numpy as np
import norlab_icp_mapper_amd as pkg
import oracle_bindings as ob

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
only = int(sys.argv[3]) if len(sys.argv) > 3 else -1
base = pkg.synth.make_scene(m=400_000, n=60_000)
t0 = time.time(); checked = 0; errors = 0
for case in range(cases):
    rng = np.random.default_rng([seed, case])  # one stream per case: `only` replays exactly the case a full run reported
    m = int(rng.choice([1, 7, 300, 5_000, 60_000, 400_000]))
    n = int(rng.choice([1, 5, 257, 4_000, 60_000]))
    k = int(rng.choice([1, 1, 1, 3, 6, 10, 8, 16, 12]))   # (r3: the cooperative kernels serve k <= 16)
    minimizer = int(rng.choice([1, 2, 2]))
    # (type, param[, iparam, param2, param3]): 6 GenericDescriptor, 7 Robust (fct | scale << 4 | dist << 8), 8 VarTrimmedDist
    rob = lambda fct, tun, sc=0, nb=0, dist=0: (7, tun, fct | (sc << 4) | (dist << 8), float(nb))
    outl = [[], [(4, 0.85)], [(3, 3.0)], [(1, 1.0), (4, 0.7)], [(2, 0.005), (4, 0.95)], [(5, 0.8), (4, 0.9)], [(4, 1.0)], [(3, 1.0), (1, 0.7)],
            [(5, 1.2)], [(8, 0.05, 0, 0.99, 0.95)], [(1, 1.5), (8, 0.3, 0, 0.8, 2.0)], [rob(0, 1.0, 1)], [rob(5, 1.5, 1, 3, 1)], [(4, 0.9), rob(1, 0.3)],
            [rob(4, 0.5, 0, 0, 1)], [rob(7, 1.5, 1)], [(6, 0.3, 4, 0.0), (4, 0.85)], [(6, 0.0, 2, 0.0)]][int(rng.integers(0, 18))]
    force = int(rng.choice([0, 0, 0, 1, 2]))
    md = float(rng.choice([0.5, 2.0, math.inf]))
    kw = dict(minimizer=minimizer, knn=k, max_dist=md if math.isfinite(md) else 1e30, outliers=outl, max_iterations=int(rng.integers(1, 25)),
              use_differential=int(rng.integers(0, 2)), smooth_length=int(rng.integers(1, 9)), min_diff_rot=float(rng.choice([1e-3, 1e-5, 1e-2])),
              min_diff_trans=float(rng.choice([1e-3, 1e-5, 1e-2])), use_bound=int(rng.integers(0, 2)), max_rot_norm=float(rng.choice([0.02, 0.8])),
              max_trans_norm=float(rng.choice([0.05, 5.0])))
    if minimizer == 2 and force: kw["force_4dof" if force == 1 else "force_2d"] = 1
    needs_rn = any(o[0] == 5 for o in outl)
    needs_scalar = any(o[0] == 6 for o in outl)
    plane_residual = any(o[0] == 7 and (o[2] >> 8) & 15 == 1 for o in outl)
    if not math.isfinite(md): kw["max_dist"] = math.inf
    sel = rng.permutation(base["map"].shape[0])[:m]
    mp, nrm = base["map"][sel], base["normals"][sel]
    rd = base["scan"][rng.permutation(base["scan"].shape[0])[:n]].copy()
    rd[:, :3] += rng.normal(0, rng.choice([0.0, 0.01, 0.3]), (n, 3)).astype(np.float32)
    if only >= 0 and case != only: continue
    try:
        icp = pkg.ICPSequence(**kw)
        icp.setMap(mp, nrm)
        scal = rng.random(m).astype(np.float32) if needs_scalar else None
        if needs_scalar: icp.setMapScalar(scal)
        rn = None
        if needs_rn:
            rn = rng.normal(0, 1, (n, 3)).astype(np.float32); rn /= np.maximum(np.linalg.norm(rn, axis=1, keepdims=True), 1e-6)
            rn[: n // 2] = nrm[rng.integers(0, m, n // 2)]
        try:
            T = icp(rd, rn)
            assert np.isfinite(T).all(), "non-finite pose"
            err_gpu = 0
        except pkg.ConvergenceError:
            err_gpu = 1
        if only >= 0:
            o = ob.OracleICP(ob.make_config(nthreads=16, **kw)); o.setMap(mp, nrm)
            if needs_scalar: o.setMapScalar(scal)
            err, T_ref = o(rd, rn)
            print(kw, m, n)
            print("GPU", err_gpu, icp.stats.iterations, icp.stats.pairs, icp.stats.stop_reason, "\n", T if not err_gpu else None)
            print("CPU", err, o.stats.iterations, o.stats.pairs, o.stats.stop_reason, "\n", T_ref)
            for it in (1, 2, 3):
                kw2 = dict(kw); kw2["max_iterations"] = it; kw2["use_differential"] = 0
                a = pkg.ICPSequence(**kw2); a.setMap(mp, nrm)
                if needs_scalar: a.setMapScalar(scal)
                Ta = a(rd, rn)
                b = ob.OracleICP(ob.make_config(nthreads=16, **kw2)); b.setMap(mp, nrm)
                if needs_scalar: b.setMapScalar(scal)
                eb, Tb = b(rd, rn)
                print("iterations", it, "pose diff", pkg.synth.pose_error(Ta, Tb), "pairs", a.stats.pairs, b.stats.pairs)
        # k >= m pairs every query with every map point: H is identically zero up to rounding, the rotation is noise on
        # both sides (ill-posed, not comparable)
        # likewise point-to-point with ONE reading point: all pairs share p, so H = sum w q p^T - (sum w q)(sum w p)^T / sum w
        # cancels to the rounding noise of two different summation orders
        if case % 5 == 0 and m * n <= 400_000 * 4_000 and k < m and not (minimizer == 1 and n == 1):
            o = ob.OracleICP(ob.make_config(nthreads=16, **kw)); o.setMap(mp, nrm)
            if needs_scalar: o.setMapScalar(scal)
            err, T_ref = o(rd, rn)
            # every pair weighted down to nothing (a hard-rejecting M-estimator far from the map: total weight 7e-19 in seed 197 case 1700): the device's
            # fixed-point pair sums resolve 2^-40, see sum w = 0 and report "transformation is not a number" where the oracle's doubles carry on with the
            # rounding noise of a vanishing H -- ill-posed on every side (oracle/DEVIATIONS.md D8), not comparable
            if err == 0 and err_gpu != 0 and icp.stats.pairs > 0 and icp.stats.weighted_point_used_ratio * k * n < 1e-9:   # (the device's statistics of the iteration it gave up in)
                icp.close(); checked += 1; continue
            assert (err != 0) == (err_gpu != 0), ("error mismatch", err, err_gpu, kw, m, n)
            # fewer than six surviving pairs (a hard-rejecting M-estimator on a seven-point map: tests/tools/data/soak_fail_420.npz,
            # scripts/r2_soak_case.py -- bit-identical for five iterations, then three pairs are left): the system is rank
            # deficient, the step is rounding noise on both sides and the two trajectories part -- ill-posed, not comparable
            if err == 0 and min(icp.stats.pairs, o.stats.pairs) >= 6:
                dt, dr = pkg.synth.pose_error(T, T_ref)
                assert dt <= 1e-3 and dr <= 1e-3, ("pose mismatch", dt, dr, kw, m, n)
                assert icp.stats.iterations == o.stats.iterations, ("iterations", icp.stats.iterations, o.stats.iterations, kw, m, n)
            checked += 1
        icp.close()
    except AssertionError as e:
        errors += 1; print("CASE", case, "FAILED:", e)
print(f"soak: {cases} cases, {checked} cross-checked, {errors} failures, {time.time() - t0:.1f} s")
sys.exit(1 if errors else 0)
