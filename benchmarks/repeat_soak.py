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
