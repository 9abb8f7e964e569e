"""Round 5, VERDICT item 1: how far is precision "f16" (one f16 product per MAC, fp32 accumulate -- the headline's
arithmetic) from the two oracles on the TRAINED-LIKE scene (density logits +-12, colour-head activations ~1e3)?

For each method, small scene (32x40, tables 2^14) and -- with --full -- the full-table 1080p frame on 4,096 sampled
rays: max |d rgb|, |d rgb_std|, |dPSNR|, |dAUSE| against the fp32 oracle and against the autocast(float16)-emulating
oracle, the gap between the two oracles, and the overflow guard's re-render count.  One JSON line per case to
gpurun_out/exp_trained_like_f16.jsonl."""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import nerf_oracle as O          # noqa: E402
from oracle import sampled_frame as SF       # noqa: E402
from uncertainty_nerf_gs_amd import metrics, render, synthetic   # noqa: E402
from test_gpu_nerf_e2e import _gt_image      # noqa: E402


def deltas(out_rgb, out_std, ref_rgb, ref_std):
    gt = _gt_image(ref_rgb)
    rec = {"psnr_ref": metrics.psnr(ref_rgb, gt), "d_psnr": abs(metrics.psnr(out_rgb, gt) - metrics.psnr(ref_rgb, gt)),
           "max_abs_rgb": (out_rgb - ref_rgb).abs().max().item(), "max_abs_rgb_std": (out_std - ref_std).abs().max().item(),
           "std_range": [ref_std.min().item(), ref_std.median().item(), ref_std.max().item()]}
    for et in ("mse", "mae", "rmse"):
        def a(rgb, std):
            err = torch.sum((rgb - gt) ** 2, -1).flatten() if et != "mae" else torch.sum((rgb - gt).abs(), -1).flatten()
            return metrics.ause((std ** 2).flatten(), err, et)[3]
        rec[f"ause_{et}_ref"] = a(ref_rgb, ref_std)
        rec[f"d_ause_{et}"] = abs(a(out_rgb, out_std) - rec[f"ause_{et}_ref"])
    return rec


def case(dev, kind, precision, full, seed=31, contrast=1.0):
    kw, okw = {}, {}
    if full:
        t = synthetic.make_scene_tensors(seed=seed, kind=kind, sharp=True, color_contrast=contrast)
        cam, c2w = dict(synthetic.CAMERA_1080P), synthetic.orbit_c2w(0.0)
    else:
        t = synthetic.make_scene_tensors(seed=seed, kind=kind, log2T=14, prop_log2T=12, sharp=True, color_contrast=contrast)
        H, W = 32, 40
        cam, c2w = dict(fx=0.9 * W, fy=0.9 * W, cx=W / 2, cy=H / 2, H=H, W=W), synthetic.orbit_c2w(0.8)
    H, W = cam["H"], cam["W"]
    if kind == "mcdropout":
        kw = dict(K=8, seed=1234, p_drop=0.2)
        okw = dict(K=8, mc_seed=1234, p_drop=0.2)
    if kind == "laplace":
        wsd, wsr = synthetic.laplace_weight_samples(t, seed=42, n_samples=100)
        kw = dict(ws_density=wsd.to(dev), ws_rgb=wsr.to(dev))
        okw = dict(ws_density=wsd, ws_rgb=wsr, depth_seed=7, depth_draws=100)
    sd = synthetic.scene_to_device(t, dev, **kw)
    sd.field.precision = precision
    out = render.render_camera(sd, c2w, depth_seed=7, **cam)
    torch.cuda.synchronize()
    total = H * W
    ids = SF.ray_runs(total, 16, 256) if full else np.arange(total)
    o, d, _ = O.generate_rays(c2w, cam["fx"], cam["fy"], cam["cx"], cam["cy"], H, W)
    sc = O.scene_from_tensors(t)
    threads = torch.get_num_threads()
    torch.set_num_threads(min(threads, 16))
    refs = {}
    for name, ac in (("fp32", None), ("autocast16", torch.float16)):
        refs[name] = SF.reference_rays(kind, sc, o.reshape(-1, 3), d.reshape(-1, 3), ids, autocast=ac, **okw)
    torch.set_num_threads(threads)
    sel = torch.from_numpy(ids).to(dev)
    got = {k: v.view(total, -1)[sel].cpu() for k, v in out.items()}
    shp = (16, 256, -1) if full else (H, W, -1)
    v = lambda x: x.view(*shp)
    rec = {"kind": kind, "precision": precision, "full": full, "contrast": contrast, "seed": seed, "overflow_rerenders": sd.overflow_rerenders}
    for name in refs:
        rec["vs_" + name] = deltas(v(got["rgb"]), v(got["rgb_std"]), v(refs[name]["rgb"]), v(refs[name]["rgb_std"]))
        rec["vs_" + name]["max_abs_acc"] = (got["accumulation"] - refs[name]["accumulation"]).abs().max().item()
    rec["oracle_gap"] = deltas(v(refs["autocast16"]["rgb"]), v(refs["autocast16"]["rgb_std"]), v(refs["fp32"]["rgb"]),
                               v(refs["fp32"]["rgb_std"]))
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true")
    ap.add_argument("--kinds", default="active,mcdropout,laplace")
    ap.add_argument("--precisions", default="f16,f16x2")
    ap.add_argument("--contrast", type=float, default=1.0)
    ap.add_argument("--seeds", default="31")
    ap.add_argument("--only-full", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "exp_trained_like_f16.jsonl"), "a") as f:
        for full in ([True] if a.only_full else [False, True] if a.full else [False]):
          for seed in [int(x) for x in a.seeds.split(",")]:
            for kind in a.kinds.split(","):
                for prec in a.precisions.split(","):
                    if full and kind == "laplace":
                        continue
                    rec = case(dev, kind, prec, full, seed=seed, contrast=a.contrast)
                    print(json.dumps(rec))
                    f.write(json.dumps(rec) + "\n")
                    f.flush()


if __name__ == "__main__":
    main()
