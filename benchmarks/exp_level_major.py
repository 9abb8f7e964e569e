"""Experiment: hash-grid gathers level-major (one 4 MiB level table at a time, L2-resident per XCD)
vs sample-major (all 16 levels per sample, 64 MiB working set).  Uses the stand-alone hashgrid entry point."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uncertainty_nerf_gs_amd import ops, render, synthetic

dev = torch.device("cuda:0")
t = synthetic.make_scene_tensors(seed=0, kind="active")
scene = synthetic.scene_to_device(t, dev)
cam = dict(synthetic.CAMERA_1080P)
H, W = cam["H"], cam["W"]
o, d, _ = ops.generate_rays(synthetic.orbit_c2w(0.4), cam["fx"], cam["fy"], cam["cx"], cam["cy"], H, W, dev, 0, 1 << 18)
sb, _ = render.sample_rays(scene, o, d, None, 0, want_prop_depth=False)
# euclidean mid points -> contracted [0,1] positions (torch, experiment only)
s_near, s_far = 0.025, 1 - 1 / 2000.0
def s2e(b):
    x = b * s_far + (1 - b) * s_near
    return torch.where(x < 0.5, 2 * x, 1 / (2 - 2 * x))
eb = s2e(sb)
mid = (eb[:, :-1] + eb[:, 1:]) / 2
pos = o[:, None, :] + d[:, None, :] * mid[..., None]
mag = pos.abs().amax(-1, keepdim=True)
pos = torch.where(mag < 1, pos, (2 - 1 / mag) * (pos / mag))
xyz = ((pos + 2) / 4).clamp(1e-6, 1 - 1e-6).reshape(-1, 3).contiguous()
N = xyz.shape[0]
table, scal, log2T = scene.field.table, scene.field.scalings, scene.field.log2T
T = 1 << log2T

def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n

def all_levels():
    ops.hashgrid_fwd(xyz, table, scal, log2T)
def level_major():
    for l in range(16):
        ops.hashgrid_fwd(xyz, table[l * T:(l + 1) * T], scal[l:l + 1].contiguous(), log2T)
print(f"N = {N} samples (2^18 rays x 48)")
print(f"sample-major, 16 levels in one kernel : {timeit(all_levels):8.3f} ms")
print(f"level-major, 16 single-level kernels  : {timeit(level_major):8.3f} ms")
for l in (0, 5, 8, 11, 15):
    tl = timeit(lambda: ops.hashgrid_fwd(xyz, table[l * T:(l + 1) * T], scal[l:l + 1].contiguous(), log2T))
    print(f"  level {l:2d} alone (res {int(scal[l])}) : {tl:8.3f} ms")
