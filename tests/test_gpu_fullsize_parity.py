"""Oracle parity AT THE BASELINE SIZE: the full-table (16 x 2^19 x 2 + 2 x 5 x 2^17 x 2) 1920x1080 frame of every NeRF
method rendered through the C ABI exactly as bench.py renders it (2^20-ray launch groups, 32,768-ray reference chunks,
the per-XCD tile walk, the dense proposal-level copies, the 8x4 patch schedule at W = 1920), compared with the CPU oracle
on 4,096 of its rays: 16 runs of 256 consecutive rays spread over the frame -- both launch groups, 16 different reference
chunks, runs that straddle image rows and patch boundaries.  The oracle renders those rays alone
(oracle/sampled_frame.py: the counter RNGs are keyed by the global ray index, so a ray's masks / depth draws are the
frame's); a few seconds of host time per method.  Same gates as the small-scene tests (|dPSNR| <= 1e-4 dB, |dAUSE| <= 1e-3
for rgb and depth) plus per-key image tolerances."""
import numpy as np
import pytest
import torch

from oracle import nerf_oracle as O
from oracle import sampled_frame as SF
from test_gpu_nerf_e2e import TIE_MARGIN, _gates, _img_close

pytestmark = pytest.mark.gpu

N_RUNS, RUN = 16, 256


def _frame(dev, method, precision, **kw):
    from uncertainty_nerf_gs_amd import render, synthetic
    t = synthetic.make_scene_tensors(seed=0, kind=method)          # full nerfacto tables, as bench.py
    sd = synthetic.scene_to_device(t, dev, **kw)
    sd.field.precision = precision
    cam, c2w = dict(synthetic.CAMERA_1080P), synthetic.orbit_c2w(0.0)
    out = render.render_camera(sd, c2w, depth_seed=7, **cam)
    assert sd.chunk_rays == 1 << 15 and out["rgb"].shape == (1080, 1920, 3)
    return t, sd, cam, c2w, out


@pytest.mark.parametrize("method,precision", [("active", "f16x2"), ("active", "f16"), ("mcdropout", "f16x2"),
                                              ("mcdropout", "f16"), ("laplace", "f16x2")])
def test_full_size_frame_matches_the_oracle_on_sampled_rays(dev, method, precision):
    from uncertainty_nerf_gs_amd import synthetic
    kw, okw = {}, {}
    if method == "mcdropout":
        kw = dict(K=8, seed=1234, p_drop=0.2)
        okw = dict(K=8, mc_seed=1234, p_drop=0.2)
    t0 = synthetic.make_scene_tensors(seed=0, kind=method) if method == "laplace" else None
    if method == "laplace":
        wsd, wsr = synthetic.laplace_weight_samples(t0, seed=42, n_samples=100)
        kw = dict(ws_density=wsd.to(dev), ws_rgb=wsr.to(dev))
        okw = dict(ws_density=wsd, ws_rgb=wsr, depth_seed=7, depth_draws=100)
    t, sd, cam, c2w, out = _frame(dev, method, precision, **kw)
    assert sd.overflow_rerenders == 0
    H, W = cam["H"], cam["W"]
    total = H * W
    ids = SF.ray_runs(total, N_RUNS, RUN)
    assert ids.min() < (1 << 20) <= ids.max() and len(set((ids // (1 << 15)).tolist())) >= N_RUNS - 1
    o, d, _ = O.generate_rays(c2w, cam["fx"], cam["fy"], cam["cx"], cam["cy"], H, W)
    threads = torch.get_num_threads()
    torch.set_num_threads(min(threads, 16))      # torch-CPU oversubscribes on the many-core GPU host
    try:
        diag = {}
        ref = SF.reference_rays(method, O.scene_from_tensors(t), o.reshape(-1, 3), d.reshape(-1, 3), ids, diagnostics=diag, **okw)
    finally:
        torch.set_num_threads(threads)
    sel = torch.from_numpy(ids).to(dev)
    got = {k: v.view(total, -1)[sel].cpu() for k, v in out.items()}
    assert set(ref) - {"density"} <= set(got), set(ref) - set(got)
    v = lambda x: x.view(N_RUNS, RUN, -1)
    _gates(f"fullsize-{method}-{precision}", v(got["rgb"]), v(got["rgb_std"]), v(ref["rgb"]), v(ref["rgb_std"]),
           out={k: v(x) for k, x in got.items()}, ref={k: v(x) for k, x in ref.items()}, diag=diag, precision=precision)
    f16 = precision == "f16"
    _img_close(got["rgb"], ref["rgb"], 1e-4 if f16 else 5e-5, 0, "rgb")
    _img_close(got["rgb_std"], ref["rgb_std"], 3e-4 if f16 else 2e-5, 5e-3, "rgb_std", max_bad_frac=2e-3)
    _img_close(got["accumulation"], ref["accumulation"], 6e-4 if f16 else 2e-4, 0, "accumulation")
    # expected depth: clipped to the CHUNK's sample range upstream, which the sampled rays cannot reproduce (module doc)
    _img_close(got["expected_depth"], ref["expected_depth"], 0, 1e-3, "expected_depth", max_bad_frac=1e-2)
    # median depth: equal except on CDF ties (asserted by the gates: test_gpu_nerf_e2e.TIE_MARGIN)
    off = ((got["depth"] - ref["depth"]).abs() > 1e-3 * ref["depth"].abs()).reshape(-1)
    assert bool((diag["median_margin"].reshape(-1)[off] <= TIE_MARGIN[precision]).all())
    if "depth_std" in ref:
        _img_close(got["depth_std"][~off], ref["depth_std"][~off], 1e-5, 2e-2 if method == "mcdropout" else 5e-3,
                   "depth_std (rays with equal medians)", max_bad_frac=1e-2)


@pytest.mark.parametrize("method", ["mcdropout", "active"])
def test_full_size_trained_like_frame_in_the_reference_arithmetic(dev, method):
    """The headline precision where f16 is hard, at the BASELINE size: the full-table 1080p frame of the TRAINED-LIKE scene
    (make_scene_tensors(sharp=True): density logits +-12, colour-head activations ~1e3) rendered at "f16" -- the
    arithmetic of the bench headline, of nerfacto-mcdropout and of tcnn-configured active-nerfacto models -- against BOTH
    oracles on the same 4,096 sampled rays: the autocast(float16)-emulating one (what the reference computes:
    mcdropout_models.py:86-92) and the fp32 one, gates on the informative target (oracle/targets.py; the plain-target
    numbers and the two oracles' own gap on it go to the parity report).  No overflow re-render on this scene."""
    from uncertainty_nerf_gs_amd import render, synthetic
    from oracle import targets
    from test_gpu_nerf_e2e import _report
    kw, okw = {}, {}
    if method == "mcdropout":
        kw = dict(K=8, seed=1234, p_drop=0.2)
        okw = dict(K=8, mc_seed=1234, p_drop=0.2)
    t = synthetic.make_scene_tensors(seed=0, kind=method, sharp=True)
    sd = synthetic.scene_to_device(t, dev, **kw)
    sd.field.precision = "f16"
    cam, c2w = dict(synthetic.CAMERA_1080P), synthetic.orbit_c2w(0.0)
    out = render.render_camera(sd, c2w, depth_seed=7, **cam)
    assert sd.overflow_rerenders == 0
    H, W = cam["H"], cam["W"]
    total = H * W
    ids = SF.ray_runs(total, N_RUNS, RUN)
    o, d, _ = O.generate_rays(c2w, cam["fx"], cam["fy"], cam["cx"], cam["cy"], H, W)
    sc = O.scene_from_tensors(t)
    threads = torch.get_num_threads()
    torch.set_num_threads(min(threads, 16))
    try:
        diag, diag16 = {}, {}
        ref = SF.reference_rays(method, sc, o.reshape(-1, 3), d.reshape(-1, 3), ids, diagnostics=diag, **okw)
        ref16 = SF.reference_rays(method, sc, o.reshape(-1, 3), d.reshape(-1, 3), ids, diagnostics=diag16, autocast=torch.float16, **okw)
    finally:
        torch.set_num_threads(threads)
    sel = torch.from_numpy(ids).to(dev)
    got = {k: v.view(total, -1)[sel].cpu() for k, v in out.items()}
    v = lambda x: x.view(N_RUNS, RUN, -1)
    vd = lambda dct: {k: v(x) for k, x in dct.items()}
    from test_gpu_nerf_e2e import TIE_MARGIN_TRAINED_LIKE
    tie = TIE_MARGIN_TRAINED_LIKE   # densities up to e^12 (test_gpu_trained_like)
    rec16 = _gates(f"fullsize-trained-like-{method}-f16-vs-autocast", v(got["rgb"]), v(got["rgb_std"]), v(ref16["rgb"]), v(ref16["rgb_std"]),
                   out=vd(got), ref=vd(ref16), diag=diag16, precision="f16", tie_margin=tie, ref_name="autocast(float16) oracle",
                   plain_other=(v(ref["rgb"]), v(ref["rgb_std"])))
    rec = _gates(f"fullsize-trained-like-{method}-f16", v(got["rgb"]), v(got["rgb_std"]), v(ref["rgb"]), v(ref["rgb_std"]),
                 out=vd(got), ref=vd(ref), diag=diag, precision="f16", tie_margin=tie,
                 plain_other=(v(ref16["rgb"]), v(ref16["rgb_std"])))
    _img_close(got["rgb"], ref16["rgb"], 2e-4, 0, "rgb vs the autocast(fp16) oracle")
    _img_close(got["rgb"], ref["rgb"], 2e-4, 0, "rgb vs the fp32 oracle")
    _img_close(got["accumulation"], ref["accumulation"], 6e-4, 0, "accumulation", max_bad_frac=5e-3)   # densities up to e^12
    gap = targets.gate_deltas(v(ref16["rgb"]), v(ref16["rgb_std"]), v(ref["rgb"]), v(ref["rgb_std"]), targets.gt_image_plain(v(ref["rgb"])))
    _report(f"fullsize-trained-like-{method}-oracle-gap-plain-target",
            {"oracles_d_psnr_plain": gap["d_psnr"], "oracles_d_ause_mse_plain": gap["d_ause_mse"],
             "build_vs_fp32_d_ause_mse_plain": rec["d_ause_mse_plain"], "build_vs_autocast_d_ause_mse_plain": rec16["d_ause_mse_plain"]})
