"""A trained-like stress scene: every other parity scene is random-init, with trunk logits and hidden
units of order 1.  `make_scene_tensors(sharp=True)` has density logits spanning about +-12, opaque surfaces and colour-head
activations of order 1e3; `overflow_units` additionally pushes trunk hidden units past 65504, the f16 operand range of
the f16 matrix kernels, whose composite-side NaN flag + fp32 re-render (render.OverflowGuard) must then kick in."""
import numpy as np
import pytest
import torch

from oracle import nerf_oracle as O

pytestmark = pytest.mark.gpu

NEAR, FAR = 0.05, 1000.0
_KERNELS = [(True, "f16"), (True, "f16x2"), (True, "fp32"), (False, "fp32")]
_KERNEL_IDS = ["mfma-f16", "mfma-f16x2", "mfma-fp32", "valu"]


def _scene(kind, dev, overflow_units=(), **kw):
    from uncertainty_nerf_gs_amd import synthetic
    t = synthetic.make_scene_tensors(seed=31, kind=kind, log2T=14, prop_log2T=12, sharp=True, overflow_units=overflow_units)
    return t, O.scene_from_tensors(t), synthetic.scene_to_device(t, dev, **kw)


def _rays(H=16, W=24, theta=0.8):
    from uncertainty_nerf_gs_amd import synthetic
    o, d, _ = O.generate_rays(synthetic.orbit_c2w(theta), 30.0, 30.0, W / 2, H / 2, H, W)
    return o.reshape(-1, 3).contiguous(), d.reshape(-1, 3).contiguous()


def _close(got, ref, rtol, atol, what, max_bad_frac=0.0):
    got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
    bad = (got - ref).abs() > (atol + rtol * ref.abs())
    assert bad.double().mean().item() <= max_bad_frac, f"{what}: {bad.double().mean().item():.3e} off, worst {(got - ref).abs().max().item():.3e}"


def test_the_scene_is_trained_like():
    """the properties the stress scene is there for, measured on the oracle"""
    from uncertainty_nerf_gs_amd import synthetic
    t = synthetic.make_scene_tensors(seed=31, kind="active", log2T=14, prop_log2T=12, sharp=True)
    sc = O.scene_from_tensors(t)
    o, d = _rays()
    sb, _, _ = O.proposal_sample(o, d, NEAR, FAR, sc.prop_nets, sc.num_prop, sc.num_nerf, 0.01)
    eb = O.spacing_to_euclidean(sb, NEAR, FAR)
    dens, rgb, beta = O.active_field(o, d, eb, sc.field)
    logit = torch.log(dens[dens > 0])
    assert logit.max() > 10 and logit.min() < -10, (float(logit.min()), float(logit.max()))
    w = O.get_weights(dens, eb[:, 1:] - eb[:, :-1])
    assert (w.sum(-1) > 0.99).float().mean() > 0.5          # opaque surfaces
    p, _ = O.normalized_positions(O.sample_positions(o, d, eb))
    feat = O.hash_encode(p.reshape(-1, 3), sc.field.grid.table, sc.field.grid.scalings, sc.field.grid.log2_T)
    h = O.mlp_forward(feat, sc.field.grid.weights, sc.field.grid.biases)
    x = O._color_inputs(d, 48, h.view(-1, 48, 17)[..., 1:16], sc.field.appearance)
    c0 = torch.relu(torch.nn.functional.linear(x, sc.field.head_w[0], sc.field.head_b[0]))
    assert c0.max() > 1e3, float(c0.max())                  # colour-head activations of order 1e3


@pytest.mark.parametrize("use_mfma,precision", _KERNELS, ids=_KERNEL_IDS)
def test_field_kernels_on_trained_like_magnitudes(dev, use_mfma, precision):
    """"f16" (the headline's arithmetic, the default of nerfacto-mcdropout and of tcnn-configured models) meets this scene
    here too: against the oracle's autocast(float16) emulation -- the arithmetic it implements: operands of every Linear
    rounded to f16, fp32 accumulate -- with the tolerance of that rounding (relative 5e-4 per operand, logits up to 12,
    activations of 1e3 through a 1e-3 layer), and no further from the FP32 oracle than a small multiple of what the two
    oracles differ by."""
    from uncertainty_nerf_gs_amd import ops
    f16 = precision == "f16"
    for kind, kw in (("active", {}), ("mcdropout", dict(K=3, seed=9, p_drop=0.2))):
        t, sc, sd = _scene(kind, dev, **kw)
        sd.field.use_mfma, sd.field.precision = use_mfma, precision
        o, d = _rays()
        sb, _, _ = O.proposal_sample(o, d, NEAR, FAR, sc.prop_nets, sc.num_prop, sc.num_nerf, 0.01)
        sb = sb.contiguous()
        eb = O.spacing_to_euclidean(sb, NEAR, FAR)
        dens, rgb, aux, _ = ops.field_fwd(o.to(dev), d.to(dev), sb.to(dev), sd.field, NEAR, FAR)
        R, S = sb.shape[0], 48
        ac = torch.float16 if f16 else None
        dtol, ctol = (2e-2, 2e-3) if f16 else (5e-4, 1e-4)   # f16: |d logit| <~ 12 x 1e-3 -> 2e-2 relative in exp
        if kind == "active":
            dr, cr, br = O.active_field(o, d, eb, sc.field, autocast=ac)
            _close(dens[0], dr, dtol, 1e-9, "density", max_bad_frac=1e-3 if f16 else 0.0)
            _close(rgb[0], cr, 0, ctol, "rgb")
            _close(aux, br, dtol, 1e-6, "beta", max_bad_frac=1e-3 if f16 else 0.0)
            if f16:   # ... and against the fp32 oracle: within 3 x the gap of the reference's two arithmetics
                d32, c32, _ = O.active_field(o, d, eb, sc.field)
                gap = (c32 - cr).abs().max().item()
                assert (rgb[0].cpu() - c32).abs().max().item() <= 3 * gap + 1e-4, gap
        else:
            sidx = (np.arange(R)[:, None] * S + np.arange(S)[None]).reshape(-1)
            for k in range(3):
                kt = torch.from_numpy(O.mc_keep_mask(9, k, sidx, 0, 64, 0.2))
                kh = torch.from_numpy(O.mc_keep_mask(9, k, sidx, 1, 64, 0.2))
                dr, cr = O.mcdropout_field(o, d, eb, sc.field, kt, kh, 0.2, autocast=ac)
                _close(dens[k], dr, dtol, 1e-9, f"density pass {k}", max_bad_frac=1e-3 if f16 else 0.0)
                _close(rgb[k], cr, 0, ctol, f"rgb pass {k}")
        assert torch.isfinite(rgb).all()


@pytest.mark.parametrize("precision", ["f16x2", "f16"])
@pytest.mark.parametrize("kind", ["active", "mcdropout", "laplace"])
def test_trained_like_scene_end_to_end(dev, kind, precision):
    """The whole frame path on the stress scene, in the fp32-equivalent arithmetic and in "f16" -- the reference's eval
    arithmetic for mc-dropout (forced autocast, mcdropout_models.py:86-92) and tcnn-configured models, and the bench
    headline's.  "f16" is held against BOTH oracles: the autocast(float16)-emulating one (what the reference computes)
    with the north-star gates and a tight image tolerance, and the fp32 one with the same gates (informative target;
    the plain-target numbers are recorded next to the gap of the two oracles on that target: oracle/targets.py).
    The f16 operand-overflow guard must not fire on this scene (its units stay inside the f16 range)."""
    diag, diag16 = {}, {}   # the oracles' median margins of the renders the gates compare with
    from uncertainty_nerf_gs_amd import metrics, render, synthetic
    t, sc, _ = _scene(kind, dev)
    H, W = 32, 40
    cam = dict(fx=0.9 * W, fy=0.9 * W, cx=W / 2, cy=H / 2, H=H, W=W)
    c2w = synthetic.orbit_c2w(0.8)
    o, d, _ = O.generate_rays(c2w, cam["fx"], cam["fy"], cam["cx"], cam["cy"], H, W)
    o, d = o.reshape(-1, 3), d.reshape(-1, 3)
    f16 = precision == "f16"
    shade = {}
    if kind == "active":
        sd = synthetic.scene_to_device(t, dev)
        oracle = lambda ac, dg: O.active_outputs(sc, o, d, autocast=ac, diagnostics=dg)
    elif kind == "mcdropout":
        sd = synthetic.scene_to_device(t, dev, K=8, seed=2, p_drop=0.2)
        oracle = lambda ac, dg: O.mcdropout_outputs(sc, o, d, 8, 2, 0.2, autocast=ac, diagnostics=dg)
    else:
        wsd, wsr = synthetic.laplace_weight_samples(t, seed=5, n_samples=30)
        sd = synthetic.scene_to_device(t, dev, ws_density=wsd.to(dev), ws_rgb=wsr.to(dev))
        noise = torch.randn(20, H * W, 48, generator=torch.Generator().manual_seed(8))
        shade = dict(depth_noise=noise.to(dev), depth_draws=20)
        oracle = lambda ac, dg: O.laplace_outputs(sc, o, d, wsd, wsr, noise, autocast=ac, diagnostics=dg)
    sd.field.precision = precision
    ref = oracle(None, diag)
    ref16 = oracle(torch.float16, diag16) if f16 else None
    out = render.render_rays(sd, o.to(dev), d.to(dev), **shade)
    assert all(torch.isfinite(v).all() for v in out.values())
    assert sd.overflow_rerenders == 0
    from test_gpu_nerf_e2e import _gates, TIE_MARGIN_TRAINED_LIKE
    v = lambda x: x.view(H, W, -1)
    rec = _gates(f"trained-like-{kind}-{precision}", v(out["rgb"].cpu()), v(out["rgb_std"].cpu()), v(ref["rgb"]), v(ref["rgb_std"]),
                 out=out, ref=ref, diag=diag, precision=precision, tie_margin=TIE_MARGIN_TRAINED_LIKE,     # densities up to e^12: see TIE_MARGIN
                 plain_other=(v(ref16["rgb"]), v(ref16["rgb_std"])) if f16 else None)
    _close(out["rgb"], ref["rgb"], 0, 4e-4 if f16 else 2e-4, "rgb", max_bad_frac=5e-3)
    _close(out["accumulation"], ref["accumulation"], 0, 5e-4, "accumulation", max_bad_frac=5e-3)
    if f16:
        rec16 = _gates(f"trained-like-{kind}-f16-vs-autocast", v(out["rgb"].cpu()), v(out["rgb_std"].cpu()), v(ref16["rgb"]),
                       v(ref16["rgb_std"]), out=out, ref=ref16, diag=diag16, precision="f16", tie_margin=TIE_MARGIN_TRAINED_LIKE,
                       ref_name="autocast(float16) oracle", plain_other=(v(ref["rgb"]), v(ref["rgb_std"])))
        _close(out["rgb"], ref16["rgb"], 0, 4e-4, "rgb vs the autocast(fp16) oracle", max_bad_frac=5e-3)
        # the plain target, for the record: this render against each oracle, and the two oracles against each other
        from oracle import targets
        gap = targets.gate_deltas(v(ref16["rgb"]), v(ref16["rgb_std"]), v(ref["rgb"]), v(ref["rgb_std"]), targets.gt_image_plain(v(ref["rgb"])))
        from test_gpu_nerf_e2e import _report
        _report(f"trained-like-{kind}-oracle-gap-plain-target", {"oracles_d_psnr_plain": gap["d_psnr"], "oracles_d_ause_mse_plain": gap["d_ause_mse"],
                "build_vs_fp32_d_ause_mse_plain": rec["d_ause_mse_plain"], "build_vs_autocast_d_ause_mse_plain": rec16["d_ause_mse_plain"]})


@pytest.mark.parametrize("precision", ["f16x2", "f16"])
@pytest.mark.parametrize("kind", ["active", "mcdropout", "laplace"])
def test_f16_operand_overflow_is_flagged_and_rerendered_in_fp32(dev, kind, precision):
    """hidden units past 65504: the f16 kernels alone give NaN samples (which nan_to_num would hide); the guard reads one
    flag word per launch group at the end of the frame and re-renders flagged groups with the exact-fp32 kernels --
    the frame then EQUALS the fp32-kernel frame, and launch groups that did not overflow are left alone"""
    from uncertainty_nerf_gs_amd import ops, render, synthetic
    t, sc, _ = _scene(kind, dev, overflow_units=(5, 41))
    kw = {}
    if kind == "mcdropout":
        kw = dict(K=4, seed=2, p_drop=0.2)
    if kind == "laplace":
        wsd, wsr = synthetic.laplace_weight_samples(t, seed=5, n_samples=30)
        kw = dict(ws_density=wsd.to(dev), ws_rgb=wsr.to(dev))
    sd = synthetic.scene_to_device(t, dev, **kw)
    assert sd.field.mfma16_blob is not None, "the weights themselves are inside the f16 range (5e4 < 6e4)"
    sd.chunk_rays = 256
    H, W = 32, 48
    cam = dict(fx=0.9 * W, fy=0.9 * W, cx=W / 2, cy=H / 2, H=H, W=W)
    c2w = synthetic.orbit_c2w(0.8)
    shade = dict(depth_seed=3) if kind == "laplace" else {}
    # the raw field kernel does overflow: NaN (split form) or inf / NaN (single-product form) samples
    o, d, _ = ops.generate_rays(c2w, cam["fx"], cam["fy"], cam["cx"], cam["cy"], H, W, dev)
    sb, _ = render.sample_rays(sd, o, d, None, want_prop_depth=False)
    sd.field.precision = precision
    dens, rgb, _, _ = ops.field_fwd(o, d, sb, sd.field, sd.near, sd.far)
    assert not (torch.isfinite(dens).all() and torch.isfinite(rgb).all()), "the stress scene must overflow the f16 operands"
    sd.field.precision = "fp32"
    want = render.render_camera(sd, c2w, rays_per_launch=512, **cam, **shade)
    assert all(torch.isfinite(v).all() for v in want.values())
    sd.field.precision = precision
    got = render.render_camera(sd, c2w, rays_per_launch=512, **cam, **shade)
    assert sd.overflow_rerenders >= 1 and sd.field.precision == precision
    if precision == "f16x2":    # every overflow of the split form is a NaN: every offending group is caught
        for k in want:
            assert torch.equal(got[k], want[k]), k
    assert all(torch.isfinite(v).all() for v in got.values())
    # without the guard the same frame is silently wrong
    sd.overflow_guard, n0 = False, sd.overflow_rerenders
    bad = render.render_camera(sd, c2w, rays_per_launch=512, **cam, **shade)
    assert sd.overflow_rerenders == n0
    assert (bad["rgb"] - want["rgb"]).abs().max() > 1e-2
    # ... and a scene that stays in range never pays for a re-render
    t2, _, _ = _scene(kind, dev)
    sd2 = synthetic.scene_to_device(t2, dev, **({} if kind != "laplace" else dict(ws_density=kw["ws_density"], ws_rgb=kw["ws_rgb"])),
                                    **({k: v for k, v in kw.items() if k in ("K", "seed", "p_drop")}))
    sd2.field.precision = precision
    render.render_camera(sd2, c2w, rays_per_launch=512, **cam, **shade)
    assert sd2.overflow_rerenders == 0


@pytest.mark.parametrize("precision", ["f16x2", "f16"])
@pytest.mark.parametrize("kind", ["active", "mcdropout", "laplace"])
def test_colour_head_overflow_is_flagged_too(dev, kind, precision):
    """An f16 operand overflow in the colour head's FIRST hidden layer: the ReLUs between the layers are
    integer maxima on the bit pattern and map a NaN or -inf whose sign bit is set to 0, so such an overflow could in
    principle be zeroed two layers before any output and leave a plausible finite colour behind.  It is not: the
    unit's f16 operand is +inf (it sits behind a ReLU), every unit of the next layer receives inf * w, the positive ones
    stay +inf through the ReLU and reach the 64 -> 3 layer, whose pre-activation is then inf or NaN -> flagged, and the
    launch group is rendered again on the exact-fp32 kernels."""
    from uncertainty_nerf_gs_amd import ops, render, synthetic
    t = synthetic.make_scene_tensors(seed=31, kind=kind, log2T=14, prop_log2T=12, sharp=True, head_overflow_units=(7, 50))
    kw = {}
    if kind == "mcdropout":
        kw = dict(K=4, seed=2, p_drop=0.2)
    if kind == "laplace":
        wsd, wsr = synthetic.laplace_weight_samples(t, seed=5, n_samples=30)
        kw = dict(ws_density=wsd.to(dev), ws_rgb=wsr.to(dev))
    sd = synthetic.scene_to_device(t, dev, **kw)
    assert sd.field.mfma16_blob is not None
    sd.chunk_rays = 256
    H, W = 32, 48
    cam = dict(fx=0.9 * W, fy=0.9 * W, cx=W / 2, cy=H / 2, H=H, W=W)
    c2w = synthetic.orbit_c2w(0.8)
    shade = dict(depth_seed=3) if kind == "laplace" else {}
    # the overflow is real: the oracle's first colour layer goes past 65504 on some samples
    sc = O.scene_from_tensors(t)
    o, d = _rays(H, W)
    x = O._color_inputs(d, 1, torch.zeros(d.shape[0], 1, 15), sc.field.appearance)
    over = torch.relu(torch.nn.functional.linear(x, sc.field.head_w[0], sc.field.head_b[0]))[:, [7, 50]] > 65504
    assert 0.0 < over.float().mean() < 0.98, "some samples overflow the f16 operand range in these units, not all"
    sd.field.precision = "fp32"
    want = render.render_camera(sd, c2w, rays_per_launch=512, **cam, **shade)
    assert all(torch.isfinite(v).all() for v in want.values())
    sd.field.precision = precision
    got = render.render_camera(sd, c2w, rays_per_launch=512, **cam, **shade)
    assert sd.overflow_rerenders >= 1
    for k in want:     # every launch group with an overflowing sample was caught: the frame IS the fp32-kernel frame
        assert torch.equal(got[k], want[k]), k


def test_composite_nan_flag(dev):
    """unerf_composite_*: nonfinite_flag is set by NaN densities / colours (not by inf, which the fp32 path produces
    too), through every composite entry point, and left alone otherwise"""
    from uncertainty_nerf_gs_amd import ops
    g = torch.Generator().manual_seed(4)
    B, R, S = 3, 70, 48
    dens = torch.exp(torch.randn(B, R, S, generator=g))
    rgb = torch.rand(B, R, S, 3, generator=g)
    sb = torch.sort(torch.rand(R, S + 1, generator=g), dim=-1).values.to(dev)

    def flags(dn, cl):
        out = []
        dp, cp = dn.permute(0, 2, 1).contiguous().to(dev), cl.permute(0, 2, 3, 1).contiguous().to(dev)
        for fn, a, b in ((ops.composite_var, dn.to(dev), cl.to(dev)), (ops.composite_moments, dn.to(dev), cl.to(dev)),
                         (ops.composite_var_planes, dp, cp), (ops.composite_moments_planes, dp, cp)):
            f = torch.zeros(1, dtype=torch.int32, device=dev)
            fn(a, b, sb, NEAR, FAR, nonfinite_flag=f)
            out.append(int(f.item()))
        return out

    assert flags(dens, rgb) == [0, 0, 0, 0]
    d2 = dens.clone()
    d2[1, 7, 3] = float("inf")
    assert flags(d2, rgb) == [0, 0, 0, 0]
    d2[2, 69, 47] = float("nan")
    assert flags(d2, rgb) == [1, 1, 1, 1]
    c2 = rgb.clone()
    c2[0, 0, 0, 2] = float("nan")
    assert flags(dens, c2) == [1, 1, 1, 1]
