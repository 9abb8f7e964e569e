"""Long repeatability soak: N renders of one 1080p frame per method x arithmetic, every one compared with the first
(tests/test_gpu_repeatability.py does four).   python benchmarks/repeat_soak.py [frames]"""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest  # noqa
from uncertainty_nerf_gs_amd import render, synthetic
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device("cuda:0")
cam = dict(synthetic.CAMERA_1080P)
rows = []
for kind, precs, grid in (("active", ("f16", "f16x2"), "torch"), ("mcdropout", ("f16", "f16x2"), "torch"), ("laplace", ("f16x2",), "torch"),
                          ("active", ("f16",), "tcnn-half"), ("mcdropout", ("f16", "f16x2"), "tcnn-half"), ("laplace", ("f16x2",), "tcnn-half")):
    t = synthetic.make_scene_tensors(seed=0, kind=kind, grid="tcnn" if grid != "torch" else "torch")
    if grid != "torch":
        t["grid_precision"] = "f16"
    kw = dict(K=8, seed=1234, p_drop=0.2) if kind == "mcdropout" else {}
    if kind == "laplace":
        wsd, wsr = synthetic.laplace_weight_samples(t, seed=42, n_samples=100)
        kw = dict(ws_density=wsd.to(dev), ws_rgb=wsr.to(dev))
    sd = synthetic.scene_to_device(t, dev, **kw)
    for prec in precs:
        sd.field.precision = prec
        ref, bad = None, 0
        for rep in range(N):
            out = render.render_camera(sd, synthetic.orbit_c2w(0.7), depth_seed=7, **cam)
            if ref is None:
                ref = {k: v.clone() for k, v in out.items()}
                continue
            bad += sum(int((ref[k] != out[k]).sum()) for k in ref)
        rows.append({"method": kind, "precision": prec, "grid": grid, "frames": N, "values_differing_from_first_frame": bad})
        print(json.dumps(rows[-1]))
    del sd

# active-splatfacto, 1 M splats: the frame (staged depth and tile sorts, both rasteriser passes) and the sorted lists themselves
from uncertainty_nerf_gs_amd import ops, splat
gp = {k: v.to(dev) for k, v in synthetic.make_splat_tensors(seed=7, N=1_000_000).items()}
bg = torch.tensor([0.1, 0.2, 0.3], device=dev)
c2w = synthetic.orbit_c2w(0.7, radius=2.5, height=0.5)
H, W = cam["H"], cam["W"]
ref, bad = None, 0
for rep in range(4 * N):
    out = splat.active_splatfacto_outputs(gp, c2w, cam["fx"], cam["fy"], cam["cx"], cam["cy"], H, W, bg)
    out = {k: v for k, v in out.items() if v is not None}
    if ref is None:
        ref = {k: v.clone() for k, v in out.items()}
        continue
    bad += sum(int((ref[k] != out[k]).sum()) for k in ref)
rows.append({"method": "active-splatfacto frame", "frames": 4 * N, "values_differing_from_first_frame": bad})
print(json.dumps(rows[-1]))
V = splat.viewmat_from_c2w(c2w)
pr = ops.splat_project(gp["means"], gp["scales"].contiguous(), 1.0, gp["quats"].contiguous(), V[:3], cam["fx"], cam["fy"], cam["cx"], cam["cy"],
                       H, W, raw=True, opacity_logits=gp["opacities"].reshape(-1).contiguous())
xys, depths, radii, conics, comp, tiles = pr[:6]
ref, bad = None, 0
for rep in range(4 * N):
    I, _, keys, gids, bins = ops.splat_bin_sort(xys, depths, radii, tiles, H, W, tight=(conics, pr[7]))
    cur = (keys.clone(), gids.clone(), bins.clone())
    if ref is None:
        ref = cur
        continue
    bad += sum(int((a != b).sum()) for a, b in zip(ref, cur))
rows.append({"method": "unerf_splat_bin_sort (ids, lists, ranges)", "pairs": int(I), "calls": 4 * N, "values_differing_from_first_call": bad})
print(json.dumps(rows[-1]))
