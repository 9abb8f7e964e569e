"""Experiment: wall time per frame vs summed kernel time for small frames (host launch overhead share)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import conftest  # noqa: F401
from uncertainty_nerf_gs_amd import ops, render, synthetic
dev = torch.device("cuda:0")
for kind, kw in (("active", {}), ("mcdropout", dict(K=8, seed=1, p_drop=0.2))):
    t = synthetic.make_scene_tensors(seed=0, kind=kind)
    sd = synthetic.scene_to_device(t, dev, **kw)
    for H, W in ((200, 200), (400, 400), (540, 960), (1080, 1920)):
        cam = dict(synthetic.CAMERA_1080P); cam.update(H=H, W=W, cx=W / 2, cy=H / 2, fx=1111.0 * W / 1920, fy=1111.0 * W / 1920)
        c2w = synthetic.orbit_c2w(0.3)
        for _ in range(3): render.render_camera(sd, c2w, **cam)
        torch.cuda.synchronize()
        ops.TIMER = ops.KernelTimer()
        n = 20 if H * W < 1e6 else 5
        t0 = time.perf_counter()
        for _ in range(n): render.render_camera(sd, c2w, **cam)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / n * 1e3
        ks = ops.TIMER.summary(); ops.TIMER = None
        kern = sum(v["total_ms"] for v in ks.values()) / n
        print(f"{kind:9s} {W}x{H}: wall {wall:7.3f} ms/frame, kernels {kern:7.3f} ms, host share {100 * (wall - kern) / wall:5.1f} %  ({H * W / wall / 1e3:.1f} Mrays/s)", flush=True)
