import sys, os, math
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import norlab_icp_mapper_amd as pkg
import oracle_bindings as ob
d = np.load('/root/repo/tests/tools/data/soak_fail_420.npz', allow_pickle=True)
mp, nrm, rd = d['mp'], d['nrm'], d['rd']
inf = math.inf
kw = eval(str(d['kw']))
print(kw, mp.shape, rd.shape)
for it in range(1, 12):
    kw2 = dict(kw); kw2["max_iterations"] = it
    a = pkg.ICPSequence(**kw2); a.setMap(mp, nrm); Ta = a(rd)
    b = ob.OracleICP(ob.make_config(nthreads=8, **kw2)); b.setMap(mp, nrm); eb, Tb = b(rd)
    print("it", it, pkg.synth.pose_error(Ta, Tb), "pairs", a.stats.pairs, b.stats.pairs, "wratio", a.stats.weighted_point_used_ratio, b.stats.weighted_point_used_ratio)
    if it in (1,2,3): print(Ta - Tb)
