"""Field widths other than nerfacto's, and use_linear proposal networks (VERDICT r4 "missing" #3).

The reference forwards hidden_dim, hidden_dim_color, features_per_level and appearance_embed_dim from the model config to
the field (activenerfacto_model.py:63-77, mcdropout_models.py:66-80, laplace_model.py:169-186), the field classes take
geo_feat_dim, and proposal_net_args_list carries use_linear.  The matrix kernels are built for 64 / 64 / 15 / 2; every other
combination runs the any-width kernel (field_kernel_generic: fp32, one lane per sample) -- one end-to-end parity case per
knob against the oracle (whose field functions take the shapes from the weights), same gates as every other case."""
import numpy as np
import pytest
import torch

from oracle import nerf_oracle as O
from test_gpu_nerf_e2e import _cam, _gates, _img_close, _oracle_rays

pytestmark = pytest.mark.gpu

CASES = [
    ("active", dict(hidden_dim=32)),
    ("active", dict(hidden_dim=128, hidden_dim_color=32)),
    ("active", dict(geo_feat_dim=7)),
    ("active", dict(appearance_dim=16)),               # default widths otherwise: the matrix kernels (bias fold only)
    ("mcdropout", dict(hidden_dim_color=128)),
    ("mcdropout", dict(features_per_level=4)),
    ("mcdropout", dict(hidden_dim=32, hidden_dim_color=32, geo_feat_dim=31, appearance_dim=8)),
    ("laplace", dict(hidden_dim=32, hidden_dim_color=128)),
    ("laplace", dict(features_per_level=4, geo_feat_dim=9)),
]


def _render(kind, t, sc, dev, H, W, c2w, cam, o, d, K=4):
    from uncertainty_nerf_gs_amd import render, synthetic
    diag, kw, shade = {}, {}, {}
    if kind == "active":
        ref = O.active_outputs(sc, o, d, diagnostics=diag)
    elif kind == "mcdropout":
        kw = dict(K=K, seed=9, p_drop=0.2)
        ref = O.mcdropout_outputs(sc, o, d, K, 9, 0.2, diagnostics=diag)
    else:
        wsd, wsr = synthetic.laplace_weight_samples(t, seed=5, n_samples=20)
        kw = dict(ws_density=wsd.to(dev), ws_rgb=wsr.to(dev))
        noise = torch.randn(10, H * W, 48, generator=torch.Generator().manual_seed(8))
        shade = dict(depth_noise=noise.to(dev), depth_draws=10)
        ref = O.laplace_outputs(sc, o, d, wsd, wsr, noise, diagnostics=diag)
    sd = synthetic.scene_to_device(t, dev, **kw)
    out = render.render_rays(sd, o.to(dev), d.to(dev), **shade)
    return sd, out, ref, diag


@pytest.mark.parametrize("kind,widths", CASES, ids=[f"{k}-" + "-".join(f"{a}{b}" for a, b in w.items()) for k, w in CASES])
def test_non_default_field_widths_end_to_end(dev, kind, widths):
    from uncertainty_nerf_gs_amd import synthetic
    with pytest.warns(UserWarning, match="any-width kernel") if set(widths) - {"appearance_dim"} else _nowarn():
        t = synthetic.make_scene_tensors(seed=21, kind=kind, log2T=13, prop_log2T=11, **widths)
        sc = O.scene_from_tensors(t)
        H, W = 24, 32
        cam, c2w = _cam(H, W), synthetic.orbit_c2w(1.1)
        o, d = _oracle_rays(c2w, cam)
        o, d = o.reshape(-1, 3), d.reshape(-1, 3)
        sd, out, ref, diag = _render(kind, t, sc, dev, H, W, c2w, cam, o, d)
    generic = bool(set(widths) - {"appearance_dim"})
    assert sd.field.any_width == generic and (sd.field.mfma16_blob is None) == generic
    v = lambda x: x.cpu().view(H, W, -1)
    _gates(f"widths-{kind}-{widths}", v(out["rgb"]), v(out["rgb_std"]), v(ref["rgb"]), v(ref["rgb_std"]), out=out, ref=ref, diag=diag)
    _img_close(out["rgb"], ref["rgb"], 1e-4, 0, "rgb")      # fp32 on both sides; sample-position amplification (worst 5.2e-5)
    _img_close(out["rgb_std"], ref["rgb_std"], 2e-5, 5e-3, "rgb_std", max_bad_frac=2e-3)
    _img_close(out["accumulation"], ref["accumulation"], 3e-4, 0, "accumulation")


class _nowarn:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def test_any_width_kernel_equals_the_matrix_kernels_at_the_default_widths(dev):
    """the any-width kernel forced onto a nerfacto-shaped field (num_levels = 16 kept, the widths passed explicitly with
    one of them off by the padding trick below is not possible -- so: compare the two on a field whose ONLY difference is
    a zero-padded hidden layer: 64 real units + 64 units with zero weights = hidden_dim 128) -- the same function"""
    from uncertainty_nerf_gs_amd import ops, render, synthetic
    t = synthetic.make_scene_tensors(seed=4, kind="active", log2T=13, prop_log2T=11)
    t2 = {**t, "field": dict(t["field"])}
    f = t2["field"]
    f["w0"] = torch.cat([f["w0"], torch.zeros(64, 32)])
    f["b0"] = torch.cat([f["b0"], torch.zeros(64)])
    f["w1"] = torch.cat([f["w1"], torch.zeros(17, 64)], dim=1)
    a = synthetic.scene_to_device(t, dev)
    a.field.precision = "fp32"
    with pytest.warns(UserWarning, match="any-width kernel"):
        b = synthetic.scene_to_device(t2, dev)
    assert b.field.any_width and b.field.hidden == 128
    H, W = 16, 24
    o, d, _ = ops.generate_rays(synthetic.orbit_c2w(0.4), 20.0, 20.0, W / 2, H / 2, H, W, dev)
    ra, rb = render.render_rays(a, o, d), render.render_rays(b, o, d)
    for k in ("rgb", "rgb_var", "accumulation", "expected_depth"):
        torch.testing.assert_close(rb[k], ra[k], rtol=2e-4, atol=2e-6, msg=k)


@pytest.mark.parametrize("grid", ["torch", "tcnn"])
def test_use_linear_proposal_networks(dev, grid):
    """proposal_net_args_list[i]["use_linear"] = True: HashMLPDensityField's single Linear on the grid features
    (unerf_density_net.hidden = 0) in both proposal passes, end to end"""
    from uncertainty_nerf_gs_amd import ops, synthetic
    t = synthetic.make_scene_tensors(seed=23, kind="active", log2T=13, prop_log2T=11, prop_linear=True, grid=grid)
    sc = O.scene_from_tensors(t)
    assert len(sc.prop_nets[0].weights) == 1
    H, W = 24, 32
    cam, c2w = _cam(H, W), synthetic.orbit_c2w(1.1)
    o, d = _oracle_rays(c2w, cam)
    o, d = o.reshape(-1, 3), d.reshape(-1, 3)
    sd, out, ref, diag = _render("active", t, sc, dev, H, W, c2w, cam, o, d)
    assert sd.props[0].b0.numel() == 0
    # the proposal densities themselves
    sb = torch.linspace(0, 1, 33)
    dens = ops.proposal_density(o.to(dev), d.to(dev), sb.to(dev), sd.props[0], 0.05, 1000.0, 0.01)
    eb = O.spacing_to_euclidean(sb[None].expand(o.shape[0], -1), 0.05, 1000.0)
    want = O.density_field(O.sample_positions(o, d, eb), sc.prop_nets[0], 0.01)
    torch.testing.assert_close(dens.cpu(), want, rtol=3e-5, atol=1e-9)
    v = lambda x: x.cpu().view(H, W, -1)
    _gates(f"use-linear-{grid}", v(out["rgb"]), v(out["rgb_std"]), v(ref["rgb"]), v(ref["rgb_std"]), out=out, ref=ref, diag=diag)
    # (linear proposal densities are steeper functions of position than the MLP's: sample-position amplification, see
    # test_active_nerfacto_camera_parity -- measured worst 6.5e-5 on one pixel with the tcnn layout)
    _img_close(out["rgb"], ref["rgb"], 1e-4, 0, "rgb")
    _img_close(out["accumulation"], ref["accumulation"], 3e-4, 0, "accumulation")


def test_models_forward_the_config_widths(dev):
    """config.hidden_dim / hidden_dim_color / features_per_level / appearance_embed_dim / use_linear reach the field and the
    proposal networks (activenerfacto_model.py:63-77), the checkpoint of such a model loads by name, and the frame equals
    the direct pipeline on the same weights"""
    from uncertainty_nerf_gs_amd import plugin, render, synthetic
    from test_gpu_models import _camera, _small_cfg, _state_dict_from_tensors
    widths = dict(hidden_dim=32, hidden_dim_color=128, features_per_level=4, appearance_dim=16)
    t = synthetic.make_scene_tensors(seed=3, kind="active", log2T=14, prop_log2T=12, prop_linear=True, **widths)
    cfg = _small_cfg(plugin.MODEL_CONFIGS["active-nerfacto"]())
    cfg.hidden_dim, cfg.hidden_dim_color, cfg.features_per_level, cfg.appearance_embed_dim = 32, 128, 4, 16
    cfg.proposal_net_args_list = [dict(a, use_linear=True) for a in cfg.proposal_net_args_list]
    with pytest.warns(UserWarning, match="any-width kernel"):
        model = cfg._target(cfg, num_train_data=4)
        sd_ck = _state_dict_from_tensors(t, "active")
        for i, p in enumerate(t["props"]):
            for k in [k for k in sd_ck if k.startswith(f"_model.proposal_networks.{i}.mlp_base.")]:
                del sd_ck[k]
            sd_ck[f"_model.proposal_networks.{i}.linear.weight"], sd_ck[f"_model.proposal_networks.{i}.linear.bias"] = p["w1"], p["b1"]
        model.load_state_dict(sd_ck)
        H, W = 24, 32
        cam = _camera(H, W)
        with torch.cuda.device(dev):
            out = model.get_outputs_for_camera(cam)
        scene = model.device_scene()
        assert scene.field.any_width and scene.field.hidden == 32 and scene.field.feat_per_level == 4 and scene.props[0].b0.numel() == 0
        direct = render.render_camera(synthetic.scene_to_device(t, dev), cam.camera_to_worlds[0], fx=0.9 * W, fy=0.9 * W,
                                      cx=W / 2, cy=H / 2, H=H, W=W, keep_density=True)
    for k in direct:
        assert torch.equal(out[k], direct[k]), k
