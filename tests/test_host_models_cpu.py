"""Host logic of the Field/Model mirrors: state-dict key names equal the reference's,
so reference checkpoints load; config defaults equal the reference's; nothing here touches a GPU."""
import pytest
import torch

from uncertainty_nerf_gs_amd import fields as F
from uncertainty_nerf_gs_amd import models as M
from uncertainty_nerf_gs_amd import plugin

SMALL = dict(log2_hashmap_size=6)


def test_active_field_keys():
    f = F.ActiveNerfactoField(num_images=3, **SMALL)
    keys = set(f.state_dict())
    for k in ("mlp_base_grid.hash_table", "mlp_base.0.hash_table", "mlp_base_mlp.layers.0.weight",
              "mlp_base.1.layers.1.bias", "mlp_head.layers.2.weight", "embedding_appearance.embedding.weight"):
        assert k in keys, k
    assert f.state_dict()["mlp_base_mlp.layers.1.weight"].shape == (17, 64)   # density + 15 geo + beta
    assert f.average_init_density == 1.0                                        # activenerfacto_field.py:159


def test_mcdropout_field_keys():
    f = F.NerfactoMCDropoutField(num_images=3, **SMALL)
    keys = set(f.state_dict())
    want = {"mlp_base_grid.hash_table", "mlp_base.0.weight", "mlp_base.0.bias", "mlp_base.3.weight", "mlp_base.3.bias",
            "mlp_head.0.weight", "mlp_head.2.weight", "mlp_head.5.weight", "mlp_head.5.bias"}
    assert want <= keys, want - keys
    assert f.state_dict()["mlp_base.3.weight"].shape == (16, 64)


def test_laplace_field_keys_and_quirks():
    f = F.NerfactoLaplaceField(num_images=3, **SMALL)
    sd = f.state_dict()
    for k in ("base_grid.hash_table", "base_mlp.0.weight", "mlp_density.weight", "mlp_hidden.weight",
              "mlp_head.0.weight", "mlp_head.2.weight", "mlp_rgb_ll.weight", "aabb", "max_res", "num_levels",
              "log2_hashmap_size"):
        assert k in sd, k
    assert len(f.base_mlp) == 1 and isinstance(f.base_mlp[0], torch.nn.Linear)   # bare Linear, no ReLU
    assert "mlp_density_ggn" not in sd and f.mlp_density_ggn.shape == (65,) and f.mlp_rgb_ggn.shape == (195,)
    g = torch.Generator().manual_seed(0)
    f.mlp_density_ggn = torch.rand(65) * 1e3
    ws_d, ws_r = f.sample_last_layers(n_samples=100, prior_prec=1.0, generator=g)
    assert ws_d.shape == (100, 65) and ws_r.shape == (100, 195)
    mu = torch.nn.utils.parameters_to_vector(f.mlp_density.parameters())
    assert (ws_d.mean(0) - mu).abs().max() < 0.5


def test_proposal_field_keys():
    """nerfstudio 1.1.0 HashMLPDensityField: mlp_base = MLPWithHashEncoding (encoder / mlp, mirrored under model.{0,1})
    + the four buffers every upstream field registers"""
    p = F.HashMLPDensityField(log2_hashmap_size=6)
    keys = set(p.state_dict())
    assert {"mlp_base.encoder.hash_table", "mlp_base.model.0.hash_table", "mlp_base.mlp.layers.0.weight",
            "mlp_base.model.1.layers.1.bias", "aabb", "max_res", "num_levels", "log2_hashmap_size"} <= keys
    assert p.encoding is p.mlp_base.encoder and len(list(p.parameters())) == 5
    pt = F.HashMLPDensityField(log2_hashmap_size=6, implementation="tcnn")
    assert {"mlp_base.encoder.tcnn_encoding.params", "mlp_base.mlp.tcnn_encoding.params"} <= set(pt.state_dict())
    n_mlp, n_grid = pt.mlp_base.fused_tcnn_sizes()
    assert n_mlp == 16 * 16 + 16 * 16 and n_grid == pt.encoding.table.numel()   # 10 -> 16 padded inputs, 1 -> 16 padded outputs


def test_plain_nerfacto_field_keys():
    f = F.NerfactoField(num_images=3, **SMALL)
    keys = set(f.state_dict())
    for k in ("mlp_base.encoder.hash_table", "mlp_base.mlp.layers.1.bias", "mlp_base.model.1.layers.0.weight",
              "mlp_head.layers.2.weight", "embedding_appearance.embedding.weight", "aabb", "max_res"):
        assert k in keys, k
    assert f.state_dict()["mlp_base.mlp.layers.1.weight"].shape == (16, 64)   # density + 15 geo


def test_model_configs_match_reference_defaults():
    assert M.NerfactoMCDropoutModelConfig().mc_samples == 10 and M.NerfactoMCDropoutModelConfig().dropout_rate == 0.2
    assert M.ActiveNerfactoModelConfig().beta_min == 0.01
    assert M.ActiveSplatfactoModelConfig().beta_min == 0.01
    assert plugin.METHOD_NAMES == ("nerfacto-mcdropout", "nerfacto-laplace", "active-nerfacto", "active-splatfacto")
    for name in plugin.METHOD_NAMES[:3]:
        cfg = plugin.MODEL_CONFIGS[name]()
        assert cfg.eval_num_rays_per_chunk == 1 << 15 and cfg.average_init_density == 0.01


def test_model_loads_nerfstudio_style_checkpoint_keys():
    cfg = M.ActiveNerfactoModelConfig(log2_hashmap_size=6)
    cfg.proposal_net_args_list = [dict(a, log2_hashmap_size=5) for a in cfg.proposal_net_args_list]
    m = M.ActiveNerfactoModel(cfg, num_train_data=4)
    src = M.ActiveNerfactoModel(cfg, num_train_data=4)
    ckpt = {"_model." + k: v + 1.0 for k, v in src.state_dict().items()}
    ckpt["_model.camera_optimizer.pose_adjustment"] = torch.zeros(4, 6)     # ignored extra key
    m.load_state_dict(ckpt)
    for k, v in src.state_dict().items():
        assert torch.equal(m.state_dict()[k], v + 1.0), k
    assert "proposal_networks.1.mlp_base.mlp.layers.0.weight" in m.state_dict()


def test_splat_model_resizes_on_load():
    m = M.ActiveSplatfactoModel(M.ActiveSplatfactoModelConfig(), num_points=10)
    ck = {f"gauss_params.{k}": torch.zeros((37,) + v.shape[1:]) for k, v in m.gauss_params.items()}
    m.load_state_dict(ck)
    assert m.gauss_params["means"].shape == (37, 3) and m.gauss_params["log_uncertainties"].shape == (37, 1)
    assert m.step == 30000
    legacy = {k: torch.ones((5,) + v.shape[1:]) for k, v in m.gauss_params.items()}   # un-prefixed names
    m.load_state_dict(legacy)
    assert m.gauss_params["quats"].shape == (5, 4) and float(m.gauss_params["quats"].sum()) == 20.0


def test_laplace_model_rejects_unbuilt_modes():
    cfg = M.NerfactoLaplaceModelConfig(log2_hashmap_size=6)
    cfg.proposal_net_args_list = [dict(a, log2_hashmap_size=5) for a in cfg.proposal_net_args_list]
    m = M.NerfactoLaplaceModel(cfg)
    with pytest.raises(NotImplementedError):
        m.get_outputs_for_camera_unc(None, is_inference=False)   # the training forward is not a render path


def test_laplace_deterministic_density_draws_only_the_colour_head():
    """use_deterministic_density=True (laplace_field.py:501-506): no density draw is made, so the generator
    reaches the colour head in its initial state; the density rows are copies of the mean parameters."""
    from torch.nn.utils import parameters_to_vector
    cfg = M.NerfactoLaplaceModelConfig(log2_hashmap_size=6)
    cfg.proposal_net_args_list = [dict(a, log2_hashmap_size=5) for a in cfg.proposal_net_args_list]
    f = M.NerfactoLaplaceModel(cfg).field
    f.mlp_rgb_ggn = torch.rand(195) * 10
    ws_d, ws_r = f.sample_last_layers(n_samples=100, generator=torch.Generator().manual_seed(3), deterministic_density=True)
    mu_d = parameters_to_vector(f.mlp_density.parameters()).detach()
    assert ws_d.shape == (100, 65) and torch.equal(ws_d, mu_d.view(1, -1).repeat(100, 1))
    mu_r = parameters_to_vector(f.mlp_rgb_ll.parameters()).detach()
    noise = torch.randn(100, 195, generator=torch.Generator().manual_seed(3))
    torch.testing.assert_close(ws_r, mu_r.view(1, -1) + noise / torch.sqrt(f.mlp_rgb_ggn + 1.0 + 1e-9))
    _, ws_r2 = f.sample_last_layers(n_samples=100, generator=torch.Generator().manual_seed(3))
    assert not torch.equal(ws_r, ws_r2)   # with density sampling the colour head sees a later generator state


def test_models_expose_the_eval_scripts_image_metrics():
    """scripts/eval_uncertainty.py:683-684 calls model.psnr / model.ssim on [1,3,H,W] tensors"""
    from uncertainty_nerf_gs_amd import metrics
    m = M.ActiveSplatfactoModel(M.ActiveSplatfactoModelConfig(), num_points=4)
    g = torch.Generator().manual_seed(0)
    img = torch.rand(1, 3, 24, 32, generator=g)
    rgb = torch.clamp(img + 0.05 * torch.randn(img.shape, generator=g), 0, 1)
    assert abs(float(m.psnr(img, rgb).item()) - metrics.psnr(rgb, img)) < 1e-12
    assert abs(float(m.ssim(img, rgb)) - metrics.ssim(rgb[0].permute(1, 2, 0), img[0].permute(1, 2, 0))) < 1e-6
    assert M.ActiveNerfactoModel.psnr is M.ActiveSplatfactoModel.psnr
    with pytest.raises(NotImplementedError):
        m.lpips(img, rgb)


def test_mcdropout_field_dropout_layer_options():
    """density_dropout_layers / rgb_dropout_layers (mcdropout_fields.py:80-81, 112-144): module layout follows
    create_mlp, the kernel's site bits follow the layout -- index 0 (dropout on the head's inputs) included"""
    from torch import nn
    from uncertainty_nerf_gs_amd import fields as F
    from uncertainty_nerf_gs_amd import lib as L
    kw = dict(num_images=2, log2_hashmap_size=6, max_res=64)
    f = F.NerfactoMCDropoutField(**kw)
    assert f.drop_sites == (L.DROP_TRUNK | L.DROP_HEAD1) and isinstance(f.mlp_base[2], nn.Dropout) and isinstance(f.mlp_head[4], nn.Dropout)
    f = F.NerfactoMCDropoutField(density_dropout_layers=False, rgb_dropout_layers=[1, -1], **kw)
    assert f.drop_sites == (L.DROP_HEAD0 | L.DROP_HEAD1)
    # the parent's trunk stays (mcdropout_fields.py:112): upstream NerfactoField.mlp_base = MLPWithHashEncoding
    assert isinstance(f.mlp_base, F.MLPWithHashEncoding) and not hasattr(f, "mlp_base_grid")
    assert "mlp_base.encoder.hash_table" in f.state_dict() and "mlp_base.mlp.layers.1.weight" in f.state_dict()
    assert [type(m).__name__ for m in f.mlp_head] == ["Linear", "ReLU", "Dropout", "Linear", "ReLU", "Dropout", "Linear", "Sigmoid"]
    f = F.NerfactoMCDropoutField(density_dropout_layers=False, rgb_dropout_layers=[], **kw)
    assert f.drop_sites == 0 and not any(isinstance(m, nn.Dropout) for m in f.modules())
    f = F.NerfactoMCDropoutField(rgb_dropout_layers=[0, -1], **kw)
    assert f.drop_sites == (L.DROP_TRUNK | L.DROP_HEADIN | L.DROP_HEAD1)
    assert [type(m).__name__ for m in f.mlp_head] == ["Dropout", "Linear", "ReLU", "Linear", "ReLU", "Dropout", "Linear", "Sigmoid"]
    import pytest
    with pytest.raises(ValueError, match="Linear layers 0, 1, 2"):
        F.NerfactoMCDropoutField(rgb_dropout_layers=[3], **kw)


def test_oracle_intersect_obb_known_answers_and_per_ray_planes():
    """intersect_obb is restated from nerfstudio.utils.math (not in this image: upstream recall) -- pinned
    here to hand-computed slab intersections; per-ray nears / fars through the sampler reduce to the scalar planes."""
    import math
    from oracle import nerf_oracle as O
    S = torch.tensor([2.0, 1.0, 4.0])
    o = torch.tensor([[-3.0, 0.0, 0.0], [-3.0, 0.0, 0.0], [0.0, 0.0, 0.0], [0.0, 0.25, 5.0], [-3.0, 2.0, 0.0]])
    d = torch.tensor([[1.0, 0.0, 0.0], [-1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, -1.0], [1.0, 0.0, 0.0]])
    n, f = O.intersect_obb(o, d, torch.eye(3), torch.zeros(3), S)
    #        enters x=-1 at t=2, leaves x=1 at t=4 | box behind the ray | starts inside: near clamps to 0 | along z | passes beside
    assert n.reshape(-1).tolist() == [2.0, 1e10, 0.0, 3.0, 1e10] and f.reshape(-1).tolist() == [4.0, 1e10, 0.5, 7.0, 1e10]
    # a rotated, shifted box is the same test in the box frame
    th = 0.6
    R = torch.tensor([[math.cos(th), -math.sin(th), 0.0], [math.sin(th), math.cos(th), 0.0], [0.0, 0.0, 1.0]])
    T = torch.tensor([0.3, -0.2, 0.1])
    n2, f2 = O.intersect_obb(o @ R.T + T, d @ R.T, R, T, S)
    torch.testing.assert_close(n2, n, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(f2, f, rtol=1e-5, atol=1e-5)
    # sampler with per-ray planes equal to the collider's constants == the scalar path
    from uncertainty_nerf_gs_amd import synthetic
    t = synthetic.make_scene_tensors(seed=3, kind="active", log2T=10, prop_log2T=9)
    sc = O.scene_from_tensors(t)
    oo = torch.tensor([[0.5, 0.1, 0.1], [0.4, -0.3, 0.2]])
    dd = torch.nn.functional.normalize(-oo + torch.tensor([[0.0, 0.05, 0.0], [0.02, 0.0, 0.0]]), dim=-1)
    a = O.active_outputs(sc, oo, dd)
    b = O.active_outputs(sc, oo, dd, torch.full((2, 1), sc.near), torch.full((2, 1), sc.far))
    for k in a:
        torch.testing.assert_close(a[k], b[k], rtol=0, atol=0)
    # ... and a narrower interval keeps every depth inside it
    c = O.active_outputs(sc, oo, dd, torch.tensor([[0.3], [0.25]]), torch.tensor([[0.8], [0.9]]))
    assert ((c["depth"] >= torch.tensor([[0.3], [0.25]])) & (c["depth"] <= torch.tensor([[0.8], [0.9]]))).all()


def test_mcdropout_model_draws_fresh_masks_for_every_render():
    """the mask seed was a class constant, so every camera of a dataset got the same mask per (pixel, sample);
    the reference's generator moves on between renders.  Frame 0 keeps the base seed (reproducible single renders)."""
    from types import SimpleNamespace
    assert M.frame_seed(77, 0) == 77
    seeds = [M.frame_seed(77, i) for i in range(200)]
    assert len(set(seeds)) == 200 and all(0 <= s < 2 ** 32 for s in seeds)
    assert M.frame_seed(78, 5) != M.frame_seed(77, 5)
    cfg = M.NerfactoMCDropoutModelConfig(log2_hashmap_size=6)
    cfg.proposal_net_args_list = [dict(a, log2_hashmap_size=5) for a in cfg.proposal_net_args_list]
    m = M.NerfactoMCDropoutModel(cfg)
    m.seed = 9
    scene = SimpleNamespace(field=SimpleNamespace(seed=None))
    got = []
    for _ in range(3):
        m._begin_render(scene)
        got.append(scene.field.seed)
    assert got == [M.frame_seed(9, 0), M.frame_seed(9, 1), M.frame_seed(9, 2)] and m.frame_counter == 3
    m.fresh_masks_per_render = False
    m._begin_render(scene)
    assert scene.field.seed == 9


def test_cameras_the_ray_kernel_does_not_model_are_refused():
    """unerf_generate_rays restates Cameras.generate_rays for one perspective / fisheye / equirectangular / orthophoto camera
    (with its OPENCV lens parameters); any other CameraType, a non-pinhole camera at a splat model, or a camera batch must
    not render silently wrong rays"""
    from types import SimpleNamespace
    base = dict(camera_to_worlds=torch.eye(4)[None, :3], fx=torch.tensor([[50.0]]), fy=torch.tensor([[50.0]]),
                cx=torch.tensor([[8.0]]), cy=torch.tensor([[6.0]]), height=torch.tensor([[12]]), width=torch.tensor([[16]]))
    c2w, cam = M._camera_args(SimpleNamespace(**base, distortion_params=torch.zeros(1, 6), camera_type=torch.tensor([[1]])))
    assert c2w.shape == (3, 4) and cam == dict(fx=50.0, fy=50.0, cx=8.0, cy=6.0, H=12, W=16)
    # lens parameters travel to the ray kernel (k1, k2, k3, k4, p1, p2); the splat models never read them
    _, cam = M._camera_args(SimpleNamespace(**base, distortion_params=torch.tensor([[-0.05, 0.02, 0, 0, 1e-3, -1e-3]])))
    assert cam["distortion"] == pytest.approx([-0.05, 0.02, 0.0, 0.0, 1e-3, -1e-3])
    _, cam = M._camera_args(SimpleNamespace(**base, distortion_params=torch.tensor([[-0.05, 0.02, 0, 0, 1e-3, -1e-3]])), lens=False)
    assert "distortion" not in cam
    with pytest.raises(ValueError, match="expected 6 values"):
        M._camera_args(SimpleNamespace(**base, distortion_params=torch.tensor([[0.1, 0, 0, 0]])))
    for ct in (2, 3, 8):     # FISHEYE, EQUIRECTANGULAR, ORTHOPHOTO travel to the ray kernel ...
        _, cam = M._camera_args(SimpleNamespace(**base, camera_type=torch.tensor([[ct]])))
        assert cam["camera_type"] == ct
        with pytest.raises(NotImplementedError, match="pinhole"):     # ... but not to gsplat's projection
            M._camera_args(SimpleNamespace(**base, camera_type=torch.tensor([[ct]])), lens=False)
    for ct in (4, 5, 6, 7, 9):
        with pytest.raises(NotImplementedError, match=f"camera_type {ct}"):
            M._camera_args(SimpleNamespace(**base, camera_type=torch.tensor([[ct]])))
    with pytest.raises(ValueError, match="one camera"):
        M._camera_args(SimpleNamespace(**dict(base, camera_to_worlds=torch.eye(4)[None, :3].repeat(3, 1, 1))))
