import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
order = sys.argv[1] if len(sys.argv) > 1 else "torch_first"
def maps():
    return sorted({l.split()[-1] for l in open("/proc/self/maps") if "hip" in l or "hsa" in l})
if order == "torch_first":
    import torch
    print("torch cuda avail:", torch.cuda.is_available(), torch.cuda.device_count())
    x = torch.ones(4, device="cuda"); print(x.sum().item())
    import norlab_icp_mapper_amd as pkg
    icp = pkg.ICPSequence(minimizer=0)
    print("icp created after torch")
else:
    import norlab_icp_mapper_amd as pkg
    icp = pkg.ICPSequence(minimizer=0)
    print("icp created before torch")
    import torch
    print("torch cuda avail:", torch.cuda.is_available())
print("\n".join(maps()))
sc = pkg.synth.make_scene(m=20000, n=2000)
icp.setMap(sc["map"])
d = torch.from_numpy(sc["scan"]).cuda()
T = icp.registerDev(d.data_ptr(), d.shape[0], fixed_iterations=3)
print(T)
