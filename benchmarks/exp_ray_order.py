"""Experiment (not part of the product path): how much does the ORDER in which rays reach the field
kernel matter for its hash-grid gathers?  Row-major launch groups vs BxB pixel blocks."""
import sys, os, math, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uncertainty_nerf_gs_amd import ops, render, synthetic

dev = torch.device("cuda:0")
method = sys.argv[1] if len(sys.argv) > 1 else "active"
kw = dict(K=8, seed=1, p_drop=0.2) if method == "mcdropout" else {}
t = synthetic.make_scene_tensors(seed=0, kind=method)
scene = synthetic.scene_to_device(t, dev, **kw)
cam = dict(synthetic.CAMERA_1080P)
H, W = cam["H"], cam["W"]
c2w = synthetic.orbit_c2w(0.4)
o, d, _ = ops.generate_rays(c2w, cam["fx"], cam["fy"], cam["cx"], cam["cy"], H, W, dev)

def blocked_perm(B):
    ii, jj = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    key = ((ii // B) * ((W + B - 1) // B) + (jj // B)) * (B * B) + (ii % B) * B + (jj % B)
    return torch.argsort(key.reshape(-1)).to(dev)

def run(order, group):
    perm = None if order == 0 else blocked_perm(order)
    oo = o if perm is None else o[perm].contiguous()
    dd = d if perm is None else d[perm].contiguous()
    for rep in range(3):
        if rep == 1:
            ops.TIMER = ops.KernelTimer()
        for s in range(0, H * W, group):
            render.render_rays(scene, oo[s:s + group], dd[s:s + group], ray_offset=s, total_rays=H * W)
    tm = ops.TIMER.summary(); ops.TIMER = None
    per = {k: round(v["total_ms"] / 2, 2) for k, v in sorted(tm.items())}
    print(f"order={order:3d} group={group:8d} total={sum(per.values()):7.2f} ms/frame  {per}", flush=True)

for g in (1 << 15, 1 << 16, 1 << 17, 1 << 18, 1 << 19):
    run(0, g)
    run(16, g)
