import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import norlab_icp_mapper_amd.synth as synth
import oracle_bindings as ob
sc = synth.make_scene(m=1_000_000, n=100_000)
for nt in (32, 64, 128, 256):
    o = ob.OracleICP(ob.make_config(max_iterations=20, nthreads=nt, minimizer=1, max_dist=2.0, outliers=[(4, 0.85)]))
    o.setMap(sc["map"], sc["normals"])
    o(sc["scan"]); o(sc["scan"])
    print(nt, o.stats.iterations / o.stats.seconds_total)
