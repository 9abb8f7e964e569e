"""How well conditioned is |dAUSE| on the synthetic bench scene?  (test infrastructure: runs the CPU oracle)

    python tests/tools/ause_conditioning.py [n_chunks] > profiles/r4_exp_ause_conditioning.json

Renders `n_chunks` strided 1024-ray chunks of bench.py's 1080p MC-dropout frame with the oracle, perturbs the oracle's
OWN outputs by uniform noise of the size the f16 kernels differ from it (rgb 1e-5, rgb_std 6e-6) and evaluates
|dPSNR| / |dAUSE| against two synthetic targets:
  * "uninformative": noise amplitude drawn independently per ray (tests/test_gpu_nerf_e2e._gt_image) -- the ranking by
    variance is a random order, AUSE ~ 0.66, and the difference of two such areas follows the order of near-tied rays;
  * "informative": noise amplitude follows the oracle's rgb_std (what bench.parity_record gates on).
The random-init scene's rgb_std all lies within 0.002 .. 0.009, so 6e-6 reorders many rays.
"""
import json
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import conftest  # noqa: F401
    from oracle import nerf_oracle as O, sampled_frame as SF
    from uncertainty_nerf_gs_amd import metrics, synthetic
    n_chunks = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    t = synthetic.make_scene_tensors(seed=0, kind="mcdropout")
    sc = O.scene_from_tensors(t)
    cam = dict(synthetic.CAMERA_1080P)
    c2w = synthetic.orbit_c2w(2 * math.pi * 6 / 24)          # the default bench line's last timed pose
    o, d, _ = O.generate_rays(c2w, cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["H"], cam["W"])
    o, d = o.reshape(-1, 3), d.reshape(-1, 3)
    chunk = 1024
    stride = max(1, o.shape[0] // 64 // chunk) * chunk
    ids = (np.arange(n_chunks, dtype=np.int64)[:, None] * stride + np.arange(chunk, dtype=np.int64)[None]).reshape(-1)
    ref = SF.reference_rays("mcdropout", sc, o, d, ids, K=8, mc_seed=1234, p_drop=0.2, step=chunk)
    rrgb, rstd = ref["rgb"], ref["rgb_std"]

    def target(seed, informative):
        g = torch.Generator().manual_seed(seed)
        if informative:
            amp = 0.3 + (rstd.mean(-1, keepdim=True) / rstd.mean()).clamp(max=4.0)
            return torch.clamp(rrgb + torch.randn(rrgb.shape, generator=g) * 0.05 * amp, 0, 1)
        return torch.clamp(rrgb + torch.randn(rrgb.shape, generator=g) * 0.05 * (0.3 + torch.rand(rrgb.shape[:1] + (1,), generator=g)), 0, 1)

    ause = lambda c, s, gt: metrics.ause((s ** 2).flatten(), torch.sum((c - gt) ** 2, -1).flatten(), "mse")[3]
    out = {"rays": int(len(ids)), "perturbation": {"rgb": 1e-5, "rgb_std": 6e-6},
           "rgb_std_quantiles_0_1_50_99_100": [float(x) for x in torch.quantile(rstd.flatten(), torch.tensor([0., .01, .5, .99, 1.]))]}
    for name, inf in (("uninformative", False), ("informative", True)):
        gt = target(123, inf)
        base = ause(rrgb, rstd, gt)
        rows = []
        for trial in range(8):
            g = torch.Generator().manual_seed(trial)
            rgb = rrgb + (torch.rand(rrgb.shape, generator=g) - 0.5) * 2e-5
            std = (rstd + (torch.rand(rstd.shape, generator=g) - 0.5) * 1.2e-5).clamp(min=0)
            rows.append({"d_ause_mse": abs(ause(rgb, std, gt) - base), "d_psnr": abs(metrics.psnr(rgb, gt) - metrics.psnr(rrgb, gt))})
        out[name] = {"ause_mse": base, "d_ause_mse_mean": float(np.mean([r["d_ause_mse"] for r in rows])),
                     "d_ause_mse_max": float(np.max([r["d_ause_mse"] for r in rows])),
                     "d_psnr_max": float(np.max([r["d_psnr"] for r in rows]))}
    print(json.dumps(out))


def oracle_gap(cache_dir="/tmp"):
    """--oracle-gap: |dPSNR| / |dAUSE| between the REFERENCE'S OWN TWO ARITHMETICS -- the oracle in fp32 and the oracle under
    the autocast(float16) the reference forces at eval (mcdropout_models.py:86-92) -- on 16 strided 1024-ray chunks of
    bench.py's 1080p frame (full tables), for mc-dropout K = 8 and active-nerfacto, random-init and trained-like scene, on
    the plain and on the informative target (oracle/targets.py), over 8 target noise seeds.  CPU only (~10 minutes on 8
    cores; oracle outputs are cached in `cache_dir`).  -> profiles/r5_exp_ause_oracle_gap.json"""
    import conftest  # noqa: F401
    from oracle import nerf_oracle as O, sampled_frame as SF, targets
    from uncertainty_nerf_gs_amd import synthetic
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    cam = dict(synthetic.CAMERA_1080P)
    c2w = synthetic.orbit_c2w(2 * math.pi * 6 / 24)
    o, d, _ = O.generate_rays(c2w, cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["H"], cam["W"])
    o, d = o.reshape(-1, 3), d.reshape(-1, 3)
    chunk, n_chunks = 1024, 16
    stride = max(1, o.shape[0] // 64 // chunk) * chunk
    ids = (np.arange(n_chunks, dtype=np.int64)[:, None] * stride * (64 // n_chunks) + np.arange(chunk, dtype=np.int64)[None]).reshape(-1)
    rows = []
    for kind in ("mcdropout", "active"):
        for sharp in (True, False):
            f = os.path.join(cache_dir, f"oracles_{kind}_{int(sharp)}.pt")
            if os.path.exists(f):
                both = torch.load(f, weights_only=False)
                a, b = both["fp32"], both["ac16"]
            else:
                sc = O.scene_from_tensors(synthetic.make_scene_tensors(seed=0, kind=kind, sharp=sharp))
                kw = dict(K=8, mc_seed=1234, p_drop=0.2) if kind == "mcdropout" else {}
                a = SF.reference_rays(kind, sc, o, d, ids, step=chunk, **kw)
                b = SF.reference_rays(kind, sc, o, d, ids, step=chunk, autocast=torch.float16, **kw)
                torch.save({"fp32": a, "ac16": b, "ids": ids}, f)
            for n in (4096, 16384):
                sel = slice(0, n) if n == 16384 else torch.arange(0, 16384, 4)
                ra, sa, rb, sb = a["rgb"][sel], a["rgb_std"][sel], b["rgb"][sel], b["rgb_std"][sel]
                row = {"method": kind, "scene": "trained-like" if sharp else "random-init", "rays": n,
                       "max_abs_rgb": float((ra - rb).abs().max()), "max_abs_rgb_std": float((sa - sb).abs().max()),
                       "rgb_std_quantiles_1_50_99": [float(x) for x in torch.quantile(sa.flatten(), torch.tensor([.01, .5, .99]))]}
                for name, mk in (("plain", lambda s_: targets.gt_image_plain(ra, s_)), ("informative", lambda s_: targets.gt_image_informative(ra, sa, s_))):
                    recs = [targets.gate_deltas(rb, sb, ra, sa, mk(seed), err_types=("mse", "mae")) for seed in range(123, 131)]
                    row[name] = {"ause_mse": float(np.mean([r["ause_mse_ref"] for r in recs])),
                                 "d_psnr_max": float(np.max([r["d_psnr"] for r in recs])),
                                 **{f"d_ause_{et}": {"seed_123": recs[0][f"d_ause_{et}"], "mean": float(np.mean([r[f"d_ause_{et}"] for r in recs])),
                                                     "max": float(np.max([r[f"d_ause_{et}"] for r in recs]))} for et in ("mse", "mae")}}
                rows.append(row)
                print(json.dumps(row), file=sys.stderr)
    print(json.dumps({"what": "fp32 oracle vs autocast(float16) oracle -- the reference's two arithmetics -- through the north-star "
                              "gate quantities, 8 target seeds; gates: |dPSNR| <= 1e-4 dB, |dAUSE| <= 1e-3", "rows": rows}, indent=1))


if __name__ == "__main__":
    if "--oracle-gap" in sys.argv:
        oracle_gap()
    else:
        main()
