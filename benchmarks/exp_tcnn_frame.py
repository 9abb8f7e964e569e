"""Per-kernel times of a 1080p active-nerfacto frame with tcnn-layout grids (the layout of reference checkpoints)."""
import os, sys, time, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import conftest  # noqa: F401
from uncertainty_nerf_gs_amd import ops, render, synthetic
dev = torch.device("cuda:0")
kinds = sys.argv[1:] or ["active"]
for kind in kinds:
    kw = dict(K=8, seed=1, p_drop=0.2) if kind == "mcdropout" else {}
    for grid in ("tcnn", "torch"):
        t = synthetic.make_scene_tensors(seed=0, kind=kind, grid=grid)
        if kind == "laplace":
            wsd, wsr = synthetic.laplace_weight_samples(t, seed=42, n_samples=100)
            kw = dict(ws_density=wsd.to(dev), ws_rgb=wsr.to(dev))
        sd = synthetic.scene_to_device(t, dev, **kw)
        cam = dict(synthetic.CAMERA_1080P)
        for i in range(2): render.render_camera(sd, synthetic.orbit_c2w(0.3 * i), **cam)
        torch.cuda.synchronize()
        ops.TIMER = ops.KernelTimer()
        n = 4
        t0 = time.perf_counter()
        for i in range(n): render.render_camera(sd, synthetic.orbit_c2w(0.3 * i), **cam)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / n * 1e3
        ks = ops.TIMER.summary(); ops.TIMER = None
        print(kind, grid, "%.2f ms/frame  %.1f Mrays/s" % (wall, 1920 * 1080 / wall / 1e3), {k: round(v["total_ms"] / n, 2) for k, v in sorted(ks.items())}, flush=True)
