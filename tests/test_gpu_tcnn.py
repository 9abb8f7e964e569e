"""tcnn-layout grids on the GPU (`implementation="tcnn"` checkpoints): the lookup against
its CPU restatement (row indices bit-exact, fp32 blend to an ulp-level tolerance), and the three NeRF methods
end to end with tcnn-layout tables in the main field and both proposal networks.
[UPSTREAM-RECALL tiny-cuda-nn, SURVEY.md A.6: parity unpinned.]"""
import math

import pytest
import torch

from oracle import nerf_oracle as O
from test_gpu_nerf_e2e import _cam, _gates, _img_close, _oracle_rays

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("L,base,max_res,log2T", [(16, 16, 2048, 14), (5, 16, 128, 12), (5, 16, 256, 17), (16, 16, 2048, 19)])
def test_tcnn_hashgrid_rows_bit_exact_and_features(dev, L, base, max_res, log2T):
    from uncertainty_nerf_gs_amd import ops
    growth = math.exp((math.log(max_res) - math.log(base)) / (L - 1))
    lv = ops.tcnn_grid_levels(L, base, growth, log2T)
    rows = lv[-1][2] + lv[-1][3]
    g = torch.Generator().manual_seed(L + log2T)
    table = (torch.rand(rows, 2, generator=g) * 2 - 1)
    x = torch.rand(3000, 3, generator=g)
    x[:40] = torch.tensor([0.0, 0.5, 1.0])[torch.randint(0, 3, (40, 3), generator=g)]   # box faces / corners
    x[40:80] = torch.round(x[40:80] * 15) / 15                                            # exact level-0 lattice points
    out, idx = ops.hashgrid_fwd_tcnn(x.to(dev), table.to(dev), lv, return_indices=True)
    ref_idx, _ = O.tcnn_hash_indices(x, lv)
    assert torch.equal(idx.cpu().long(), ref_idx), "row indices must match bit for bit"
    ref = O.tcnn_hash_encode(x, table, lv)
    torch.testing.assert_close(out.cpu(), ref, rtol=2e-6, atol=2e-7)
    assert ops.hashgrid_fwd_tcnn(x[:0].to(dev), table.to(dev), lv).shape == (0, 2 * L)


@pytest.mark.parametrize("kind", ["active", "mcdropout", "laplace"])
def test_tcnn_layout_camera_parity(dev, kind):
    diag = {}   # the oracle's median margins of the render the gates compare with
    from uncertainty_nerf_gs_amd import ops, render, synthetic
    t = synthetic.make_scene_tensors(seed=5, kind=kind, log2T=14, prop_log2T=12, grid="tcnn")
    sc = O.scene_from_tensors(t)
    assert sc.field.grid.tcnn_levels is not None and sc.field.sh_remap
    H, W = 28, 36
    cam, c2w = _cam(H, W), synthetic.orbit_c2w(1.3)
    o, d = _oracle_rays(c2w, cam)
    if kind == "active":
        sd = synthetic.scene_to_device(t, dev)
        sd.chunk_rays = 512
        out = render.render_camera(sd, c2w, rays_per_launch=1024, keep_density=True, **cam)
        ref = O.render_camera(lambda oo, dd, off: O.active_outputs(sc, oo, dd, diagnostics=diag), o, d, chunk=512)
    elif kind == "mcdropout":
        sd = synthetic.scene_to_device(t, dev, K=8, seed=9, p_drop=0.2)   # K = 8: the BASELINE config
        sd.chunk_rays = 512
        out = render.render_camera(sd, c2w, rays_per_launch=1024, **cam)
        ref = O.render_camera(lambda oo, dd, off: O.mcdropout_outputs(sc, oo, dd, 8, 9, 0.2, ray_offset=off, diagnostics=diag), o, d, chunk=512)
    else:
        wsd, wsr = synthetic.laplace_weight_samples(t, seed=42, n_samples=100)
        sd = synthetic.scene_to_device(t, dev, ws_density=wsd.to(dev), ws_rgb=wsr.to(dev))
        noise = torch.randn(100, H * W, 48, generator=torch.Generator().manual_seed(8))
        od, dd_, _ = ops.generate_rays(c2w, cam["fx"], cam["fy"], cam["cx"], cam["cy"], H, W, dev)
        out = {k: v.view(H, W, -1) for k, v in render.render_rays(sd, od, dd_, depth_noise=noise.to(dev)).items()}
        ref = {k: v.view(H, W, -1) for k, v in
               O.laplace_outputs(sc, o.reshape(-1, 3), d.reshape(-1, 3), wsd, wsr, noise, diagnostics=diag).items()}
    assert sd.field.tcnn_levels is not None and sd.props[0].tcnn_levels is not None and sd.field.sh_remap == 1
    _gates(f"tcnn_{kind}", out["rgb"].cpu(), out["rgb_std"].cpu(), ref["rgb"], ref["rgb_std"], out=out, ref=ref, diag=diag)
    _img_close(out["rgb"], ref["rgb"], 5e-5, 0, "rgb")
    _img_close(out["rgb_std"], ref["rgb_std"], 2e-5, 5e-3, "rgb_std", max_bad_frac=2e-3)
    _img_close(out["accumulation"], ref["accumulation"], 3e-4, 0, "accumulation")
    _img_close(out["expected_depth"], ref["expected_depth"], 0, 2e-3, "expected_depth", max_bad_frac=5e-3)


def test_valu_field_kernel_agrees_with_mfma_kernel_on_tcnn_grid(dev):
    from uncertainty_nerf_gs_amd import ops, render, synthetic
    t = synthetic.make_scene_tensors(seed=6, kind="active", log2T=14, prop_log2T=12, grid="tcnn")
    sd = synthetic.scene_to_device(t, dev)
    H, W = 16, 20
    o, d, _ = ops.generate_rays(synthetic.orbit_c2w(0.2), 18.0, 18.0, W / 2, H / 2, H, W, dev)
    sb, _ = render.sample_rays(sd, o, d, None, want_prop_depth=False)
    a = ops.field_fwd(o, d, sb, sd.field, sd.near, sd.far)
    sd.field.use_mfma = False
    b = ops.field_fwd(o, d, sb, sd.field, sd.near, sd.far)
    for x, y, name in zip(a[:3], b[:3], ("density", "rgb", "beta")):
        torch.testing.assert_close(x, y, rtol=2e-4, atol=2e-6, msg=name)


def test_model_from_tcnn_layout_checkpoint(dev):
    """a checkpoint with tinycudann parameter vectors loads by name and renders the same image as the direct
    pipeline built from the unpacked weights"""
    from uncertainty_nerf_gs_amd import fields as F
    from uncertainty_nerf_gs_amd import models, plugin, render, synthetic
    t = synthetic.make_scene_tensors(seed=7, kind="active", log2T=14, prop_log2T=12, grid="tcnn")
    for blk in [t["field"]] + t["props"]:
        for k in ("b0", "b1"):
            blk[k] = torch.zeros_like(blk[k])          # tcnn MLPs carry no biases
    t["field"]["head_b"] = [torch.zeros_like(b) for b in t["field"]["head_b"]]

    def pack(ws, in_dim, out_dim):   # inverse of unpack_tcnn_mlp
        pad = lambda n: -(-n // 16) * 16
        first = torch.zeros(ws[0].shape[0], pad(in_dim))
        first[:, :in_dim] = ws[0]
        last = torch.zeros(pad(out_dim), ws[-1].shape[1])
        last[:out_dim] = ws[-1]
        return torch.cat([first.reshape(-1)] + [w.reshape(-1) for w in ws[1:-1]] + [last.reshape(-1)])

    f = t["field"]
    sd_ckpt = {"field.mlp_base_grid.tcnn_encoding.params": f["table"].reshape(-1),
               "field.mlp_base_mlp.tcnn_encoding.params": pack([f["w0"], f["w1"]], 32, 17),
               "field.mlp_head.tcnn_encoding.params": pack(list(f["head_w"]), 63, 3),
               "field.embedding_appearance.embedding.weight": f["appearance"][None]}   # one image: its mean is exact
    for i, p in enumerate(t["props"]):
        sd_ckpt[f"proposal_networks.{i}.encoding.tcnn_encoding.params"] = p["table"].reshape(-1)
        sd_ckpt[f"proposal_networks.{i}.mlp_base.1.tcnn_encoding.params"] = pack([p["w0"], p["w1"]], 10, 1)
    cfg = plugin.MODEL_CONFIGS["active-nerfacto"]()
    cfg.implementation, cfg.log2_hashmap_size = "tcnn", 14
    cfg.proposal_net_args_list = [dict(a, log2_hashmap_size=12) for a in cfg.proposal_net_args_list]
    model = cfg._target(cfg, num_train_data=1)
    model.load_state_dict({"_model." + k: v for k, v in sd_ckpt.items()})
    H, W = 24, 32
    cam = models.Camera(camera_to_worlds=synthetic.orbit_c2w(0.9)[None], fx=torch.tensor([0.9 * W]),
                        fy=torch.tensor([0.9 * W]), cx=W / 2, cy=H / 2, height=H, width=W)
    with torch.cuda.device(dev):
        out = model.get_outputs_for_camera(cam)
    # implementation="tcnn": the reference's MLPs are fp16 FullyFusedMLPs, so the model renders with precision "f16" --
    # and its grids in tcnn's half arithmetic (half tables on the device, the fp32 master vector in the state dict)
    assert model.device_scene().field.precision == "f16"
    assert model.device_scene().field.grid_precision == "f16" and model.device_scene().props[0].grid_precision == "f16"
    assert model.device_scene().field.table.dtype == torch.float16
    t["grid_precision"] = "f16"
    sd = synthetic.scene_to_device(t, dev)
    sd.field.precision = "f16"
    direct = render.render_camera(sd, cam.camera_to_worlds[0], fx=0.9 * W, fy=0.9 * W,
                                  cx=W / 2, cy=H / 2, H=H, W=W, keep_density=True)
    for k in direct:
        assert torch.equal(out[k], direct[k]), k


# ---- round 5: tcnn's OWN arithmetic -- half tables, half interpolation (unerf_field_params.grid_half) -------------------

@pytest.mark.parametrize("L,base,max_res,log2T,scale", [(16, 16, 2048, 14, 1.0), (5, 16, 128, 12, 1.0), (5, 16, 256, 17, 1e-4),
                                                        (16, 16, 2048, 19, 30.0)])
def test_tcnn_half_grid_features_bit_exact(dev, L, base, max_res, log2T, scale):
    """unerf_hashgrid_fwd_tcnn_half against oracle.tcnn_hash_encode_half -- tcnn's kernel_grid with T = __half: half copy
    of the parameters, fp32 weight product rounded to half, half fused multiply-add per corner in corner order.  Every
    rounding is specified, so the features must be EQUAL (tables of order 1, of tcnn's init scale 1e-4 -- half denormals
    in the products -- and of order 30)."""
    from uncertainty_nerf_gs_amd import ops
    growth = math.exp((math.log(max_res) - math.log(base)) / (L - 1))
    lv = ops.tcnn_grid_levels(L, base, growth, log2T)
    rows = lv[-1][2] + lv[-1][3]
    g = torch.Generator().manual_seed(L + log2T)
    table = (torch.rand(rows, 2, generator=g) * 2 - 1) * scale
    x = torch.rand(3000, 3, generator=g)
    x[:40] = torch.tensor([0.0, 0.5, 1.0])[torch.randint(0, 3, (40, 3), generator=g)]
    x[40:80] = torch.round(x[40:80] * 15) / 15
    out = ops.hashgrid_fwd_tcnn_half(x.to(dev), table.to(dev), lv)
    ref = O.tcnn_hash_encode_half(x, table, lv)
    assert torch.equal(out.cpu(), ref), f"{(out.cpu() != ref).sum().item()} of {ref.numel()} half features differ"
    assert torch.equal(out.cpu(), out.cpu().half().float())          # they ARE half values
    # and they are what the fp32 blend gives, to half precision
    f32 = O.tcnn_hash_encode(x, table, lv)
    assert (out.cpu() - f32).abs().max() <= 4e-3 * scale
    assert ops.hashgrid_fwd_tcnn_half(x[:0].to(dev), table.to(dev), lv).shape == (0, 2 * L)


@pytest.mark.parametrize("use_mfma,precision", [(True, "f16"), (True, "f16x2"), (True, "fp32"), (False, "fp32")],
                         ids=["mfma-f16", "mfma-f16x2", "mfma-fp32", "valu"])
@pytest.mark.parametrize("kind", ["active", "mcdropout", "laplace"])
def test_field_kernels_on_the_half_grid(dev, kind, use_mfma, precision):
    """every field kernel reads the half grid (TCNN = 2 instances of the matrix kernels, the uniform branch of the VALU
    kernel): densities / colours against the oracle's field functions fed by tcnn_hash_encode_half.  The split and
    exact kernels then differ from the oracle by fp32 rounding only; "f16" by its operand rounding (the FEATURES carry
    none: they are half values on both sides)."""
    from uncertainty_nerf_gs_amd import ops, synthetic
    t = synthetic.make_scene_tensors(seed=8, kind=kind, log2T=14, prop_log2T=12, grid="tcnn")
    t["grid_precision"] = "f16"
    sc = O.scene_from_tensors(t)
    assert sc.field.grid.grid_half and sc.prop_nets[0].grid_half
    kw = {}
    if kind == "mcdropout":
        kw = dict(K=3, seed=9, p_drop=0.2)
    if kind == "laplace":
        wsd, wsr = synthetic.laplace_weight_samples(t, seed=5, n_samples=30)
        kw = dict(ws_density=wsd.to(dev), ws_rgb=wsr.to(dev))
    sd = synthetic.scene_to_device(t, dev, **kw)
    assert sd.field.table.dtype == torch.float16 and sd.props[0].table.dtype == torch.float16
    sd.field.use_mfma, sd.field.precision = use_mfma, precision
    H, W = 16, 24
    o, d, _ = O.generate_rays(synthetic.orbit_c2w(0.8), 30.0, 30.0, W / 2, H / 2, H, W)
    o, d = o.reshape(-1, 3).contiguous(), d.reshape(-1, 3).contiguous()
    sb, _, _ = O.proposal_sample(o, d, 0.05, 1000.0, sc.prop_nets, sc.num_prop, sc.num_nerf, 0.01)
    sb = sb.contiguous()
    eb = O.spacing_to_euclidean(sb, 0.05, 1000.0)
    dens, rgb, aux, aux2 = ops.field_fwd(o.to(dev), d.to(dev), sb.to(dev), sd.field, 0.05, 1000.0)
    f16 = precision == "f16"
    dtol, ctol = (4e-3, 2e-3) if f16 else (2e-5, 2e-6)
    close = lambda got, ref, rtol, atol, what: torch.testing.assert_close(got.cpu(), ref, rtol=rtol, atol=atol, msg=lambda m: f"{what}: {m}")
    if kind == "active":
        dr, cr, br = O.active_field(o, d, eb, sc.field)
        close(dens[0], dr, dtol, 1e-7, "density")
        close(rgb[0], cr, 0, ctol, "rgb")
        close(aux, br, dtol, 1e-6, "beta")
    elif kind == "mcdropout":
        import numpy as np
        R, S = sb.shape[0], 48
        sidx = (np.arange(R)[:, None] * S + np.arange(S)[None]).reshape(-1)
        for k in range(3):
            kt = torch.from_numpy(O.mc_keep_mask(9, k, sidx, 0, 64, 0.2))
            kh = torch.from_numpy(O.mc_keep_mask(9, k, sidx, 1, 64, 0.2))
            dr, cr = O.mcdropout_field(o, d, eb, sc.field, kt, kh, 0.2)
            close(dens[k], dr, dtol, 1e-7, f"density pass {k}")
            close(rgb[k], cr, 0, ctol, f"rgb pass {k}")
    else:
        mu_d, var_d, mu_rgb, var_rgb = O.laplace_field(o, d, eb, sc.field, wsd, wsr)
        close(dens[0], mu_d, dtol, 1e-7, "density mean")
        close(rgb[0], mu_rgb, 0, ctol, "rgb mean")


@pytest.mark.parametrize("kind", ["active", "mcdropout", "laplace"])
def test_half_grid_camera_parity(dev, kind):
    """the three methods end to end on tcnn-layout HALF grids (main field and both proposal networks) in the arithmetic the
    models pick for implementation="tcnn" -- "f16" dense layers for active / mc-dropout, fp32-equivalent for Laplace --
    against the oracle on tcnn_hash_encode_half: the north-star gates plus image tolerances."""
    diag = {}
    from uncertainty_nerf_gs_amd import render, synthetic
    t = synthetic.make_scene_tensors(seed=5, kind=kind, log2T=14, prop_log2T=12, grid="tcnn")
    t["grid_precision"] = "f16"
    sc = O.scene_from_tensors(t)
    H, W = 28, 36
    cam, c2w = _cam(H, W), synthetic.orbit_c2w(1.3)
    o, d = _oracle_rays(c2w, cam)
    o, d = o.reshape(-1, 3), d.reshape(-1, 3)
    kw, shade = {}, {}
    if kind == "active":
        ref = O.active_outputs(sc, o, d, diagnostics=diag)
    elif kind == "mcdropout":
        kw = dict(K=8, seed=9, p_drop=0.2)
        ref = O.mcdropout_outputs(sc, o, d, 8, 9, 0.2, diagnostics=diag)
    else:
        wsd, wsr = synthetic.laplace_weight_samples(t, seed=42, n_samples=100)
        kw = dict(ws_density=wsd.to(dev), ws_rgb=wsr.to(dev))
        noise = torch.randn(100, H * W, 48, generator=torch.Generator().manual_seed(8))
        shade = dict(depth_noise=noise.to(dev))
        ref = O.laplace_outputs(sc, o, d, wsd, wsr, noise, diagnostics=diag)
    sd = synthetic.scene_to_device(t, dev, **kw)
    prec = "f16x2" if kind == "laplace" else "f16"
    sd.field.precision = prec
    out = render.render_rays(sd, o.to(dev), d.to(dev), **shade)
    assert sd.overflow_rerenders == 0
    v = lambda x: x.cpu().view(H, W, -1)
    # (mc-dropout's depth is the mean of K = 8 medians: eight chances per pixel to sit on a CDF tie -- every differing
    # pixel is checked to BE one, TIE_MARGIN -- and "f16" moves a CDF by more than the split form: 2.9 % of 1,008 pixels)
    _gates(f"tcnn_half_{kind}", v(out["rgb"]), v(out["rgb_std"]), v(ref["rgb"]), v(ref["rgb_std"]), out=out, ref=ref, diag=diag,
           precision=prec, depth_off_max=5e-2 if kind == "mcdropout" else 1.6e-2)
    f16 = prec == "f16"
    _img_close(out["rgb"], ref["rgb"], 1e-4 if f16 else 5e-5, 0, "rgb")
    _img_close(out["rgb_std"], ref["rgb_std"], 3e-4 if f16 else 2e-5, 5e-3, "rgb_std", max_bad_frac=2e-3)
    _img_close(out["accumulation"], ref["accumulation"], 1e-3 if f16 else 3e-4, 0, "accumulation")
    # the half grid is a different function from the fp32 one -- by half rounding of 16 x 2 features, visible in the image
    t32 = dict(t)
    t32.pop("grid_precision")
    sd32 = synthetic.scene_to_device(t32, dev, **kw)
    sd32.field.precision = prec
    out32 = render.render_rays(sd32, o.to(dev), d.to(dev), **shade)
    assert not torch.equal(out32["rgb"], out["rgb"])
