#!/usr/bin/env python
"""precision="f16" (one f16 product per MAC, fp32 accumulate -- unerf_field_params.f16_single) measured against
(a) the fp32 oracle and (b) the oracle's autocast(float16)-emulating mode, field level and end to end, plus the 1080p
frame time of each precision.  Writes one JSON (stdout).

    python benchmarks/exp_f16_single.py [--no-time]
"""
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import nerf_oracle as O  # noqa: E402
from uncertainty_nerf_gs_amd import lib, metrics, ops, render, synthetic  # noqa: E402

NEAR, FAR = 0.05, 1000.0


def dev_stats(got, ref):
    got, ref = got.detach().cpu().double(), ref.double()
    d = (got - ref).abs()
    return {"max_abs": float(d.max()), "mean_abs": float(d.mean()), "max_rel": float((d / (ref.abs() + 1e-12)).max()),
            "p999_rel": float(torch.quantile((d / (ref.abs() + 1e-6)).flatten()[:4_000_000], 0.999))}


def gt_image(ref_rgb):
    g = torch.Generator().manual_seed(123)
    noise = torch.randn(ref_rgb.shape, generator=g) * 0.05 * (0.3 + torch.rand(ref_rgb.shape[:2] + (1,), generator=g))
    return torch.clamp(ref_rgb + noise, 0, 1)


def gates(out_rgb, out_std, ref_rgb, ref_std):
    gt = gt_image(ref_rgb)
    rec = {"d_psnr": abs(metrics.psnr(out_rgb, gt) - metrics.psnr(ref_rgb, gt)),
           "max_abs_rgb": float((out_rgb - ref_rgb).abs().max()), "max_abs_rgb_std": float((out_std - ref_std).abs().max())}
    for et in ("mse", "mae", "rmse"):
        def a(rgb, std):
            err = torch.sum((rgb - gt) ** 2, -1).flatten() if et != "mae" else torch.sum((rgb - gt).abs(), -1).flatten()
            return metrics.ause((std ** 2).flatten(), err, et)[3]
        rec[f"d_ause_{et}"] = abs(a(out_rgb, out_std) - a(ref_rgb, ref_std))
    return rec


def main():
    lib.build_library()
    dev = torch.device("cuda:0")
    res = {"field": {}, "e2e": {}, "frame_ms": {}}
    H, W = 36, 48
    cam = dict(fx=0.9 * W, fy=0.9 * W, cx=W / 2, cy=H / 2, H=H, W=W)
    for kind in ("active", "mcdropout", "laplace"):
        t = synthetic.make_scene_tensors(seed=1, kind=kind, log2T=14, prop_log2T=12)
        sc = O.scene_from_tensors(t)
        c2w = synthetic.orbit_c2w(2.1)
        o, d, _ = O.generate_rays(c2w, cam["fx"], cam["fy"], cam["cx"], cam["cy"], H, W)
        o, d = o.reshape(-1, 3), d.reshape(-1, 3)
        kw, K, seed = {}, 8, 1234
        if kind == "mcdropout":
            kw = dict(K=K, seed=seed, p_drop=0.2)
        if kind == "laplace":
            wsd, wsr = synthetic.laplace_weight_samples(t, seed=42, n_samples=100)
            kw = dict(ws_density=wsd.to(dev), ws_rgb=wsr.to(dev))
            noise = torch.randn(100, H * W, 48, generator=torch.Generator().manual_seed(8))
        refs = {}
        for name, ac in (("fp32", None), ("autocast16", torch.float16)):
            if kind == "active":
                refs[name] = O.active_outputs(sc, o, d, autocast=ac)
            elif kind == "mcdropout":
                refs[name] = O.mcdropout_outputs(sc, o, d, K, seed, 0.2, autocast=ac)
            else:
                refs[name] = O.laplace_outputs(sc, o, d, wsd, wsr, noise, autocast=ac)
        res["e2e"][kind] = {"autocast16_vs_fp32_oracle": gates(refs["autocast16"]["rgb"].view(H, W, 3), refs["autocast16"]["rgb_std"].view(H, W, 1),
                                                               refs["fp32"]["rgb"].view(H, W, 3), refs["fp32"]["rgb_std"].view(H, W, 1))}
        for prec in ("f16x2", "f16"):
            sd = synthetic.scene_to_device(t, dev, **kw)
            sd.field.precision = prec
            shade = dict(depth_noise=noise.to(dev)) if kind == "laplace" else {}
            out = render.render_rays(sd, o.to(dev), d.to(dev), **shade)
            for name in refs:
                res["e2e"][kind][f"{prec}_vs_{name}"] = gates(out["rgb"].cpu().view(H, W, 3), out["rgb_std"].cpu().view(H, W, 1),
                                                               refs[name]["rgb"].view(H, W, 3), refs[name]["rgb_std"].view(H, W, 1))
                res["e2e"][kind][f"{prec}_vs_{name}"]["accumulation"] = dev_stats(out["accumulation"], refs[name]["accumulation"])["max_abs"]
    # field level (mcdropout, the headline kernel): per-pass density / rgb against both oracles
    t = synthetic.make_scene_tensors(seed=0, kind="mcdropout", log2T=14, prop_log2T=12)
    sc = O.scene_from_tensors(t)
    o, d, _ = O.generate_rays(synthetic.orbit_c2w(0.3), 30.0, 30.0, 12, 8, 16, 24)
    o, d = o.reshape(-1, 3).contiguous(), d.reshape(-1, 3).contiguous()
    sb, _, _ = O.proposal_sample(o, d, NEAR, FAR, sc.prop_nets, sc.num_prop, sc.num_nerf, 0.01)
    eb = O.spacing_to_euclidean(sb, NEAR, FAR)
    R, S = sb.shape[0], sb.shape[1] - 1
    sidx = ((np.arange(R)[:, None] + 0) * S + np.arange(S)[None]).reshape(-1)
    for prec in ("f16x2", "f16"):
        sd = synthetic.scene_to_device(t, dev, K=3, seed=7, p_drop=0.2)
        sd.field.precision = prec
        dens, rgb, _, _ = ops.field_fwd(o.to(dev), d.to(dev), sb.contiguous().to(dev), sd.field, NEAR, FAR)
        for name, ac in (("fp32", None), ("autocast16", torch.float16)):
            worst_d, worst_c = 0.0, 0.0
            for k in range(3):
                kt = torch.from_numpy(O.mc_keep_mask(7, k, sidx, 0, 64, 0.2))
                kh = torch.from_numpy(O.mc_keep_mask(7, k, sidx, 1, 64, 0.2))
                dr, cr = O.mcdropout_field(o, d, eb, sc.field, kt, kh, 0.2, autocast=ac)
                worst_d = max(worst_d, dev_stats(dens[k], dr)["p999_rel"])
                worst_c = max(worst_c, dev_stats(rgb[k], cr)["max_abs"])
            res["field"][f"mcdropout_{prec}_vs_{name}"] = {"density_p999_rel": worst_d, "rgb_max_abs": worst_c}
    if "--no-time" not in sys.argv:
        camf = dict(synthetic.CAMERA_1080P)
        poses = [synthetic.orbit_c2w(2 * math.pi * i / 24) for i in range(24)]
        for kind in ("mcdropout", "active", "laplace"):
            t = synthetic.make_scene_tensors(seed=0, kind=kind)
            kw = {}
            if kind == "mcdropout":
                kw = dict(K=8, seed=1234, p_drop=0.2)
            if kind == "laplace":
                wsd, wsr = synthetic.laplace_weight_samples(t, seed=42, n_samples=100)
                kw = dict(ws_density=wsd.to(dev), ws_rgb=wsr.to(dev))
            sd = synthetic.scene_to_device(t, dev, **kw)
            for prec in ("f16x2", "f16", "f16x2", "f16"):
                sd.field.precision = prec
                for i in range(2):
                    render.render_camera(sd, poses[i], **camf)
                torch.cuda.synchronize()
                ops.TIMER = ops.KernelTimer()
                t0 = time.perf_counter()
                for i in range(5):
                    out = render.render_camera(sd, poses[2 + i], **camf)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / 5 * 1e3
                ks = ops.TIMER.summary()
                ops.TIMER = None
                assert torch.isfinite(out["rgb"]).all()
                res["frame_ms"].setdefault(f"{kind}_{prec}", []).append(
                    {"frame_ms": dt, "Mrays_s": 1920 * 1080 / dt / 1e3, "field_fwd_ms_per_launch": ks["field_fwd"]["avg_ms"],
                     "field_fwd_ms_per_frame": ks["field_fwd"]["total_ms"] / 5})
            del sd
            torch.cuda.empty_cache()
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
