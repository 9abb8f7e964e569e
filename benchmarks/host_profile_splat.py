"""Host-side profile of the splat frame (cProfile over 200 frames): where the Python / HIP-API time between the kernels goes.
   python3 benchmarks/host_profile_splat.py   (on the GPU box)"""
import cProfile, pstats, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from importlib import import_module
pkg = import_module("uncertainty_nerf_gs_amd") if False else None
import __graft_entry__ as g
g.build()
from uncertainty_nerf_gs_amd import splat, synthetic
dev = torch.device("cuda:0")
gp = {k: v.to(dev) for k, v in synthetic.make_splat_tensors(seed=0, N=1_000_000).items()}
cam = dict(synthetic.CAMERA_1080P)
H, W = cam["H"], cam["W"]
bg = torch.tensor([0.1, 0.2, 0.3], device=dev)
poses = [synthetic.orbit_c2w(0.1 * i, radius=2.5, height=0.5) for i in range(8)]
def frame(i):
    return splat.active_splatfacto_outputs(gp, poses[i % 8], cam["fx"], cam["fy"], cam["cx"], cam["cy"], H, W, bg)
for i in range(10):
    frame(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(200):
    frame(i)
torch.cuda.synchronize()
print("ms per frame", (time.perf_counter() - t0) / 200 * 1e3)
pr = cProfile.Profile()
pr.enable()
for i in range(200):
    frame(i)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue())
