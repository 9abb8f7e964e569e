#!/usr/bin/env python
"""Experiment (CPU, oracle only): how far is the reference's forced-autocast MC-dropout arithmetic
(mcdropout_models.py:86-92: fp16 Linear layers on a GPU, bf16 on the CPU) from the fp32 semantics of the same graph,
which is what this build matches?  Same scene, same rays, same dropout masks; only the Linear layers change.

    python benchmarks/exp_autocast_divergence.py  ->  one JSON line (committed as profiles/r2_autocast_divergence.json)
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import nerf_oracle as O                      # noqa: E402
from uncertainty_nerf_gs_amd import metrics, synthetic   # noqa: E402


def main():
    t = synthetic.make_scene_tensors(seed=1, kind="mcdropout", log2T=14, prop_log2T=12)
    sc = O.scene_from_tensors(t)
    H, W, K = 40, 56, 8
    c2w = synthetic.orbit_c2w(2.1)
    o, d, _ = O.generate_rays(c2w, 0.9 * W, 0.9 * W, W / 2, H / 2, H, W)
    o, d = o.reshape(-1, 3), d.reshape(-1, 3)
    ref = O.mcdropout_outputs(sc, o, d, K, 1234, 0.2)
    g = torch.Generator().manual_seed(123)
    gt = torch.clamp(ref["rgb"] + torch.randn(ref["rgb"].shape, generator=g) * 0.05, 0, 1)
    out = {"frame": f"{W}x{H}", "K": K, "psnr_fp32_vs_gt": metrics.psnr(ref["rgb"], gt)}
    for name, dt in (("fp16", torch.float16), ("bf16", torch.bfloat16)):
        alt = O.mcdropout_outputs(sc, o, d, K, 1234, 0.2, autocast=dt)
        a = lambda r: metrics.ause((r["rgb_std"] ** 2).flatten(), torch.sum((r["rgb"] - gt) ** 2, -1).flatten(), "mse")[3]
        out[name] = {"max_abs_rgb": float((alt["rgb"] - ref["rgb"]).abs().max()),
                     "max_abs_rgb_std": float((alt["rgb_std"] - ref["rgb_std"]).abs().max()),
                     "d_psnr_dB": abs(metrics.psnr(alt["rgb"], gt) - out["psnr_fp32_vs_gt"]),
                     "d_ause_mse": abs(a(alt) - a(ref))}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
