"""Experiment: cost of the uncertainty metrics next to the render in the eval harness (1080p, active-nerfacto)."""
import sys, time, torch
sys.path.insert(0, ".")
from uncertainty_nerf_gs_amd import eval as E, render, synthetic
dev = torch.device("cuda:0")
t = synthetic.make_scene_tensors(seed=0, kind="active")
sd = synthetic.scene_to_device(t, dev)
cam = synthetic.CAMERA_1080P
views = [synthetic.orbit_c2w(0.3 * i) for i in range(4)]
gt = [torch.clamp(render.render_camera(sd, v, **cam)["rgb"] + 0.05 * torch.randn(cam["H"], cam["W"], 3, device=dev), 0, 1) for v in views]
fn = lambda c2w: render.render_camera(sd, c2w, **cam)
E.get_average_uncertainty_metrics(fn, list(zip(views[:1], gt[:1])))   # warm-up
torch.cuda.synchronize(); t0 = time.perf_counter()
avg, _ = E.get_average_uncertainty_metrics(fn, list(zip(views, gt)))
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / len(views)
print("per image: %.1f ms total; num_rays_per_sec (render + metrics) %.2f M, render only %.2f M" % (dt * 1e3, avg["num_rays_per_sec"] / 1e6, avg["render_rays_per_sec"] / 1e6))
print({k: round(v, 5) for k, v in avg.items() if k.startswith(("psnr", "rgb_ause", "rgb_nll", "rgb_auc"))})
