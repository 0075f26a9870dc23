import sys, os, math
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import norlab_icp_mapper_amd as pkg
import oracle_bindings as ob
base = pkg.synth.make_scene(m=400_000, n=60_000)
bad = 0
for case in range(300):
    rng = np.random.default_rng([77, case])
    m = int(rng.choice([7, 7, 300])); n = int(rng.choice([5, 257, 4000]))
    fct = int(rng.choice([0, 2, 3, 4, 5])); tun = float(rng.choice([0.5, 1.0, 0.05]))
    kw = dict(minimizer=int(rng.choice([1, 2])), knn=1, max_dist=math.inf, outliers=[(7, tun, fct | (0 << 4) | (1 << 8), 0.0)], max_iterations=11)
    sel = rng.permutation(base["map"].shape[0])[:m]
    mp, nrm = base["map"][sel], base["normals"][sel]
    rd = base["scan"][rng.permutation(base["scan"].shape[0])[:n]].copy()
    rd[:, :3] += rng.normal(0, rng.choice([0.0, 0.01, 0.3]), (n, 3)).astype(np.float32)
    icp = pkg.ICPSequence(**kw); icp.setMap(mp, nrm)
    try: T = icp(rd); eg = 0
    except pkg.ConvergenceError: eg = 1
    o = ob.OracleICP(ob.make_config(nthreads=8, **kw)); o.setMap(mp, nrm)
    err, Tr = o(rd)
    if (err != 0) != (eg != 0): print("case", case, "error mismatch", err, eg, kw, m, n); bad += 1; continue
    if err == 0:
        dt, dr = pkg.synth.pose_error(T, Tr)
        if dt > 1e-3 or dr > 1e-3:
            bad += 1
            print("case", case, "mismatch", dt, dr, kw, m, n, "iters", icp.stats.iterations, o.stats.iterations, "pairs", icp.stats.pairs, o.stats.pairs)
            for it in range(1, 8):
                kw2 = dict(kw); kw2["max_iterations"] = it
                a = pkg.ICPSequence(**kw2); a.setMap(mp, nrm); Ta = a(rd)
                b = ob.OracleICP(ob.make_config(nthreads=8, **kw2)); b.setMap(mp, nrm); eb, Tb = b(rd)
                print("   it", it, pkg.synth.pose_error(Ta, Tb), a.stats.pairs, b.stats.pairs, a.stats.weighted_point_used_ratio, b.stats.weighted_point_used_ratio)
print("bad", bad)
