"""Drop-in surface on the GPU: the Model mirrors (reference class names / methods / output keys)
drive the HIP kernels; ensemble aggregation runs on the HIP moments kernel."""
import numpy as np
import pytest
import torch

from oracle import nerf_oracle as O

pytestmark = pytest.mark.gpu


def _small_cfg(cfg):
    cfg.log2_hashmap_size = 14
    cfg.proposal_net_args_list = [dict(a, log2_hashmap_size=12) for a in cfg.proposal_net_args_list]
    return cfg


def _state_dict_from_tensors(t, kind):
    """synthetic weight dict -> reference-named checkpoint (`_model.` prefixed like nerfstudio's)"""
    f = t["field"]
    sd = {}
    if kind == "active":
        sd["field.mlp_base_grid.hash_table"] = f["table"]
        sd["field.mlp_base_mlp.layers.0.weight"], sd["field.mlp_base_mlp.layers.0.bias"] = f["w0"], f["b0"]
        sd["field.mlp_base_mlp.layers.1.weight"], sd["field.mlp_base_mlp.layers.1.bias"] = f["w1"], f["b1"]
        for i in range(3):
            sd[f"field.mlp_head.layers.{i}.weight"], sd[f"field.mlp_head.layers.{i}.bias"] = f["head_w"][i], f["head_b"][i]
    elif kind == "mcdropout":
        sd["field.mlp_base_grid.hash_table"] = f["table"]
        sd["field.mlp_base.0.weight"], sd["field.mlp_base.0.bias"] = f["w0"], f["b0"]
        sd["field.mlp_base.3.weight"], sd["field.mlp_base.3.bias"] = f["w1"], f["b1"]
        for i, j in enumerate((0, 2, 5)):
            sd[f"field.mlp_head.{j}.weight"], sd[f"field.mlp_head.{j}.bias"] = f["head_w"][i], f["head_b"][i]
    else:
        sd["field.base_grid.hash_table"] = f["table"]
        sd["field.base_mlp.0.weight"], sd["field.base_mlp.0.bias"] = f["w0"], f["b0"]
        sd["field.mlp_hidden.weight"], sd["field.mlp_hidden.bias"] = f["w1"], f["b1"]
        sd["field.mlp_density.weight"], sd["field.mlp_density.bias"] = f["density_w"], f["density_b"]
        sd["field.mlp_head.0.weight"], sd["field.mlp_head.0.bias"] = f["head_w"][0], f["head_b"][0]
        sd["field.mlp_head.2.weight"], sd["field.mlp_head.2.bias"] = f["head_w"][1], f["head_b"][1]
        sd["field.mlp_rgb_ll.weight"], sd["field.mlp_rgb_ll.bias"] = f["head_w"][2], f["head_b"][2]
    sd["field.embedding_appearance.embedding.weight"] = f["appearance"][None].repeat(4, 1)  # mean == appearance
    for i, p in enumerate(t["props"]):
        sd[f"proposal_networks.{i}.encoding.hash_table"] = p["table"]
        sd[f"proposal_networks.{i}.mlp_base.1.layers.0.weight"] = p["w0"]
        sd[f"proposal_networks.{i}.mlp_base.1.layers.0.bias"] = p["b0"]
        sd[f"proposal_networks.{i}.mlp_base.1.layers.1.weight"] = p["w1"]
        sd[f"proposal_networks.{i}.mlp_base.1.layers.1.bias"] = p["b1"]
    return {"_model." + k: v for k, v in sd.items()}


def _camera(H, W, theta=0.7):
    from uncertainty_nerf_gs_amd import models, synthetic
    return models.Camera(camera_to_worlds=synthetic.orbit_c2w(theta)[None], fx=torch.tensor([0.9 * W]),
                         fy=torch.tensor([0.9 * W]), cx=W / 2, cy=H / 2, height=H, width=W)


@pytest.mark.parametrize("camera_type", [2, 3, 8])
def test_model_renders_the_other_camera_models_through_the_same_path(dev, camera_type):
    """A nerfstudio camera of type FISHEYE / EQUIRECTANGULAR / ORTHOPHOTO at Model.get_outputs_for_camera (what the reference's
    nerfstudio-format parsers hand over when transforms.json says so: sparse_nerfstudio_dataparser.py:277-279): the frame
    equals, bit for bit, the same model fed the bundle of ops.generate_rays(camera_type=...) -- whose rays the kernel test
    holds against the oracle's restatement of Cameras._generate_rays_from_coords -- and differs from the pinhole frame."""
    from types import SimpleNamespace
    from uncertainty_nerf_gs_amd import ops, plugin, synthetic
    t = synthetic.make_scene_tensors(seed=3, kind="active", log2T=14, prop_log2T=12)
    cfg = _small_cfg(plugin.MODEL_CONFIGS["active-nerfacto"]())
    model = cfg._target(cfg, num_train_data=4)
    model.load_state_dict(_state_dict_from_tensors(t, "active"))
    H, W = (32, 64) if camera_type == 3 else (40, 56)
    f = 32.0 if camera_type == 3 else 0.5 * W
    c2w = synthetic.orbit_c2w(0.7)
    cam = SimpleNamespace(camera_to_worlds=c2w[None, :3], fx=torch.tensor([[f]]), fy=torch.tensor([[f]]), cx=torch.tensor([[W / 2]]),
                          cy=torch.tensor([[H / 2]]), height=torch.tensor([[H]]), width=torch.tensor([[W]]),
                          camera_type=torch.tensor([[camera_type]]), distortion_params=None)
    with torch.cuda.device(dev):
        model.to(dev)
        out = model.get_outputs_for_camera(cam)
        o, d, _ = ops.generate_rays(c2w[:3], f, f, W / 2, H / 2, H, W, dev, camera_type=camera_type)
        want = model.get_outputs_for_camera_ray_bundle(SimpleNamespace(origins=o.view(H, W, 3), directions=d.view(H, W, 3)))
        pin = model.get_outputs_for_camera(SimpleNamespace(**{**cam.__dict__, "camera_type": torch.tensor([[1]])}))
    assert set(out) == set(want)
    for k in want:
        assert torch.equal(out[k], want[k]), k
    assert torch.isfinite(out["rgb"]).all() and not torch.equal(out["rgb"], pin["rgb"])


@pytest.mark.parametrize("kind,method", [("active", "active-nerfacto"), ("mcdropout", "nerfacto-mcdropout")])
def test_model_from_checkpoint_equals_direct_pipeline(dev, kind, method):
    from uncertainty_nerf_gs_amd import plugin, render, synthetic
    t = synthetic.make_scene_tensors(seed=3, kind=kind, log2T=14, prop_log2T=12)
    cfg = _small_cfg(plugin.MODEL_CONFIGS[method]())
    model = cfg._target(cfg, num_train_data=4)
    model.load_state_dict(_state_dict_from_tensors(t, kind))
    kw = {}
    if kind == "mcdropout":
        cfg.mc_samples = 4
        model.seed = 77
        model.invalidate()
        kw = dict(K=4, seed=77, p_drop=0.2)
    H, W = 40, 56
    cam = _camera(H, W)
    with torch.cuda.device(dev):
        out = model.get_outputs_for_camera(cam)
    sd = synthetic.scene_to_device(t, dev, **kw)
    # the model renders in the REFERENCE's arithmetic: mc-dropout under its forced fp16 autocast (mcdropout_models.py:86-92),
    # an implementation="torch" active-nerfacto in fp32 (the split-f16 kernels)
    assert model.device_scene().field.precision == sd.field.precision.replace("f16x2", "f16" if kind == "mcdropout" else "f16x2")
    sd.field.precision = model.device_scene().field.precision
    ref = render.render_camera(sd, cam.camera_to_worlds[0], fx=0.9 * W, fy=0.9 * W, cx=W / 2, cy=H / 2, H=H, W=W,
                               keep_density=(kind == "active"))
    assert set(out) == set(ref)
    for k in ref:
        assert torch.equal(out[k], ref[k]), k
    expect = {"active": {"rgb", "accumulation", "depth", "expected_depth", "density", "rgb_var", "rgb_std", "depth_var",
                         "depth_std", "prop_depth_0", "prop_depth_1"},
              "mcdropout": {"rgb", "accumulation", "depth", "expected_depth", "prop_depth_0", "prop_depth_1", "rgb_std",
                            "depth_std", "expected_depth_std"}}[kind]
    assert set(out) == expect
    assert out["rgb"].shape == (H, W, 3) and out["rgb_std"].shape == (H, W, 1)


def test_get_outputs_on_a_ray_bundle_equals_the_oracle_chunk(dev):
    """Model.get_outputs / forward on a flat bundle = one reference chunk (clip bounds of that bundle)."""
    from types import SimpleNamespace
    from uncertainty_nerf_gs_amd import plugin, synthetic
    t = synthetic.make_scene_tensors(seed=3, kind="active", log2T=14, prop_log2T=12)
    cfg = _small_cfg(plugin.MODEL_CONFIGS["active-nerfacto"]())
    model = cfg._target(cfg, num_train_data=4)
    model.load_state_dict(_state_dict_from_tensors(t, "active"))
    g = torch.Generator().manual_seed(1)
    o = torch.randn(700, 3, generator=g) * 0.3
    d = torch.nn.functional.normalize(torch.randn(700, 3, generator=g), dim=-1)
    with torch.cuda.device(dev):
        out = model(SimpleNamespace(origins=o.to(dev), directions=d.to(dev)))
        out2 = model.get_outputs((o.to(dev), d.to(dev)))
    ref = O.active_outputs(O.scene_from_tensors(t), o, d)
    assert set(ref) <= set(out)
    for k in out:
        assert torch.equal(out[k], out2[k])
    for k, atol, rtol in (("rgb", 5e-5, 0), ("accumulation", 3e-4, 0), ("expected_depth", 0, 2e-3), ("rgb_std", 1e-5, 2e-3)):
        got, want = out[k].cpu().double(), ref[k].double()
        bad = (got - want).abs() > atol + rtol * want.abs()
        assert bad.double().mean() <= 5e-3, (k, (got - want).abs().max().item())


def test_laplace_model_unc_render_matches_oracle(dev):
    from uncertainty_nerf_gs_amd import plugin, synthetic
    t = synthetic.make_scene_tensors(seed=4, kind="laplace", log2T=14, prop_log2T=12)
    cfg = _small_cfg(plugin.MODEL_CONFIGS["nerfacto-laplace"]())
    model = cfg._target(cfg, num_train_data=4)
    model.load_state_dict(_state_dict_from_tensors(t, "laplace"))
    g = torch.Generator().manual_seed(5)
    model.field.mlp_density_ggn = torch.rand(65, generator=g) * 1e3
    model.field.mlp_rgb_ggn = torch.rand(195, generator=g) * 1e3
    H, W = 16, 24
    cam = _camera(H, W, 2.0)
    with torch.cuda.device(dev):
        out = model.get_outputs_for_camera_unc(cam, is_inference=True, prior_prec=1.0, n_samples=100,
                                               generator=torch.Generator().manual_seed(9))
    assert set(out) == {"rgb", "rgb_std", "accumulation", "depth", "depth_std", "expected_depth", "prop_depth_0",
                        "prop_depth_1"}
    # oracle with the same last-layer draws (the generator is consumed in the same order: density, rgb)
    g2 = torch.Generator().manual_seed(9)
    f = t["field"]
    mu_d = torch.cat([f["density_w"].reshape(-1), f["density_b"].reshape(-1)])
    mu_r = torch.cat([f["head_w"][2].reshape(-1), f["head_b"][2].reshape(-1)])
    wsd = O.laplace_weight_samples(mu_d, model.field.mlp_density_ggn, 1.0, 1e-9, torch.randn(100, 65, generator=g2))
    wsr = O.laplace_weight_samples(mu_r, model.field.mlp_rgb_ggn, 1.0, 1e-9, torch.randn(100, 195, generator=g2))
    sc = O.scene_from_tensors(t)
    o, d, _ = O.generate_rays(cam.camera_to_worlds[0], 0.9 * W, 0.9 * W, W / 2, H / 2, H, W)
    sidx = (np.arange(H * W)[:, None] * 48 + np.arange(48)[None]).reshape(-1)
    noise = torch.from_numpy(np.stack([O.normal_noise(0, dd, sidx).reshape(H * W, 48) for dd in range(100)]))
    ref = O.laplace_outputs(sc, o.reshape(-1, 3), d.reshape(-1, 3), wsd, wsr, noise)
    for k, atol, rtol in (("rgb", 5e-5, 0), ("rgb_std", 2e-5, 5e-3), ("accumulation", 3e-4, 0), ("expected_depth", 0, 2e-3)):
        got, want = out[k].cpu().reshape(H * W, -1).double(), ref[k].double()
        bad = (got - want).abs() > atol + rtol * want.abs()
        assert bad.double().mean() <= 5e-3, (k, (got - want).abs().max().item())


def test_laplace_model_draws_fresh_last_layer_samples_in_every_eval_chunk(dev):
    """NerfactoLaplaceModel.resample = "chunk" (default): get_outputs_for_camera_unc renders every chunk of
    config.eval_num_rays_per_chunk rays with its own sample_laplace draw, consuming the generator as the reference's chunk
    loop does (laplace_model.py:432-443: chunk 0 density, chunk 0 colour, chunk 1 density, ...); "camera": one draw."""
    from oracle import sampled_frame as SF
    from uncertainty_nerf_gs_amd import plugin, synthetic
    t = synthetic.make_scene_tensors(seed=4, kind="laplace", log2T=14, prop_log2T=12)
    cfg = _small_cfg(plugin.MODEL_CONFIGS["nerfacto-laplace"]())
    cfg.eval_num_rays_per_chunk = 128
    model = cfg._target(cfg, num_train_data=4)
    model.load_state_dict(_state_dict_from_tensors(t, "laplace"))
    g = torch.Generator().manual_seed(5)
    model.field.mlp_density_ggn = torch.rand(65, generator=g) * 1e3
    model.field.mlp_rgb_ggn = torch.rand(195, generator=g) * 1e3
    assert model.resample == "chunk"
    H, W = 16, 24                                  # 384 rays = 3 chunks
    cam = _camera(H, W, 2.0)
    with torch.cuda.device(dev):
        out = model.get_outputs_for_camera_unc(cam, n_samples=50, generator=torch.Generator().manual_seed(9))
    g2 = torch.Generator().manual_seed(9)
    f = t["field"]
    mu_d = torch.cat([f["density_w"].reshape(-1), f["density_b"].reshape(-1)])
    mu_r = torch.cat([f["head_w"][2].reshape(-1), f["head_b"][2].reshape(-1)])
    sets = []
    for _ in range(3):     # quirk kept: the colour head always draws 100 samples (laplace_field.py:516-520)
        sets.append((O.laplace_weight_samples(mu_d, model.field.mlp_density_ggn, 1.0, 1e-9, torch.randn(50, 65, generator=g2)),
                     O.laplace_weight_samples(mu_r, model.field.mlp_rgb_ggn, 1.0, 1e-9, torch.randn(100, 195, generator=g2))))
    sc = O.scene_from_tensors(t)
    o, d, _ = O.generate_rays(cam.camera_to_worlds[0], 0.9 * W, 0.9 * W, W / 2, H / 2, H, W)

    def chunk_fn(oo, dd, off):
        ids = np.arange(off, off + oo.shape[0], dtype=np.int64)
        return O.laplace_outputs(sc, oo, dd, sets[off // 128][0], sets[off // 128][1], SF.depth_noise_for(ids, 48, 0, 100))

    ref = O.render_camera(chunk_fn, o, d, chunk=128)
    for k, atol, rtol in (("rgb", 5e-5, 0), ("rgb_std", 2e-5, 5e-3), ("accumulation", 3e-4, 0), ("expected_depth", 0, 2e-3)):
        got, want = out[k].cpu().reshape(H * W, -1).double(), ref[k].reshape(H * W, -1).double()
        bad = (got - want).abs() > atol + rtol * want.abs()
        assert bad.double().mean() <= 5e-3, (k, (got - want).abs().max().item())
    model.resample = "camera"
    with torch.cuda.device(dev):
        one = model.get_outputs_for_camera_unc(cam, n_samples=50, generator=torch.Generator().manual_seed(9))
    assert torch.allclose(one["rgb"].reshape(-1, 3)[:128], out["rgb"].reshape(-1, 3)[:128], atol=1e-6)   # chunk 0: the same draw
    assert float((one["rgb_std"] - out["rgb_std"]).abs().max()) > 1e-4
    # a chunk size that is not a multiple of 32 cannot carry per-chunk sets (a kernel tile is 32 rays): it
    # renders with one set for the frame -- what "camera" does -- and says so, instead of refusing the frame
    model.resample = "chunk"
    cfg.eval_num_rays_per_chunk = 100
    model.invalidate()
    with torch.cuda.device(dev), pytest.warns(UserWarning, match="not a multiple of 32"):
        odd = model.get_outputs_for_camera_unc(cam, n_samples=50, generator=torch.Generator().manual_seed(9))
    assert torch.allclose(odd["rgb_std"], one["rgb_std"], atol=2e-6)     # (expected_depth clips per chunk: not compared)
    # ... and the scratch arena of the frame path survives the two invalidate() calls of every *_unc frame
    ws = model.device_scene().workspace
    if ws is not None:
        held = ws.nbytes()
        with torch.cuda.device(dev):
            model.get_outputs_for_camera_unc(cam, n_samples=50, generator=torch.Generator().manual_seed(9))
        assert model.device_scene().workspace is ws and ws.nbytes() == held > 0
        model.release()
        assert ws.nbytes() == 0 and model._dev_scene is None


def test_splat_counts_in_flight_keep_their_own_values(dev):
    """ops.SplatCount: several counts started before any is awaited -- pipelined frames, ensemble members --
    each read their own number (a ring of pinned words, wait() caches its value), in any order and repeatedly"""
    from uncertainty_nerf_gs_amd import lib as L, ops
    hits = [torch.full((1000 + 37 * i,), i + 1, dtype=torch.int32, device=dev) for i in range(ops.SplatCount.RING)]
    counts = [ops.SplatCount(h) for h in hits]
    want = [int(h.sum()) for h in hits]
    assert [c.wait() for c in reversed(counts)] == list(reversed(want))
    assert [c.wait() for c in counts] == want                         # again: cached
    more = [ops.SplatCount(h) for h in hits]                          # the ring wraps: the awaited words are free
    with pytest.raises(L.UnerfError, match="without wait"):
        ops.SplatCount(hits[0])                                       # ... a ninth unawaited one is refused
    assert [c.wait() for c in more] == want and [c.wait() for c in counts] == want


def test_laplace_model_deterministic_density_matches_oracle(dev):
    """LaplaceConfig.use_deterministic_density=True (laplace_field.py:501-506): plain selector-masked density,
    sampled colour head, depth from the ordinary weights; the generator is consumed by the colour head only."""
    from uncertainty_nerf_gs_amd import plugin, synthetic
    t = synthetic.make_scene_tensors(seed=4, kind="laplace", log2T=14, prop_log2T=12)
    cfg = _small_cfg(plugin.MODEL_CONFIGS["nerfacto-laplace"]())
    model = cfg._target(cfg, num_train_data=4)
    model.load_state_dict(_state_dict_from_tensors(t, "laplace"))
    g = torch.Generator().manual_seed(5)
    model.field.mlp_density_ggn = torch.rand(65, generator=g) * 1e3
    model.field.mlp_rgb_ggn = torch.rand(195, generator=g) * 1e3
    H, W = 16, 24
    cam = _camera(H, W, 2.0)
    with torch.cuda.device(dev):
        out = model.get_outputs_for_camera_unc(cam, is_inference=True, use_deterministic_density=True, prior_prec=1.0,
                                               n_samples=100, generator=torch.Generator().manual_seed(9))
    f = t["field"]
    mu_r = torch.cat([f["head_w"][2].reshape(-1), f["head_b"][2].reshape(-1)])
    mu_d = torch.cat([f["density_w"].reshape(-1), f["density_b"].reshape(-1)])
    wsr = O.laplace_weight_samples(mu_r, model.field.mlp_rgb_ggn, 1.0, 1e-9,
                                   torch.randn(100, 195, generator=torch.Generator().manual_seed(9)))
    sc = O.scene_from_tensors(t)
    o, d, _ = O.generate_rays(cam.camera_to_worlds[0], 0.9 * W, 0.9 * W, W / 2, H / 2, H, W)
    ref = O.laplace_outputs(sc, o.reshape(-1, 3), d.reshape(-1, 3), mu_d.view(1, -1).repeat(100, 1), wsr, None,
                            use_deterministic_density=True)
    for k, atol, rtol in (("rgb", 5e-5, 0), ("rgb_std", 2e-5, 5e-3), ("accumulation", 3e-4, 0), ("expected_depth", 0, 2e-3),
                          ("depth_std", 0, 1e-2)):
        got, want = out[k].cpu().reshape(H * W, -1).double(), ref[k].double()
        bad = (got - want).abs() > atol + rtol * want.abs()
        assert bad.double().mean() <= 1e-2, (k, (got - want).abs().max().item())


@pytest.mark.parametrize("activation", ["trunc_exp", "softplus"])
def test_laplace_compute_hessian_naive_matches_autograd_oracle(dev, activation):
    """GGN fitting (laplace_model.py:343-400): closed-form Jacobian kernels vs one autograd backward per
    rendered value on the CPU oracle, through the Model method the eval script calls; both density activations of
    NerfactoLaplaceModelConfig (laplace_model.py:151)."""
    from uncertainty_nerf_gs_amd import plugin, synthetic
    t = synthetic.make_scene_tensors(seed=4, kind="laplace", log2T=14, prop_log2T=12)
    cfg = _small_cfg(plugin.MODEL_CONFIGS["nerfacto-laplace"]())
    cfg.density_activation = activation
    model = cfg._target(cfg, num_train_data=4)
    model.load_state_dict(_state_dict_from_tensors(t, "laplace"))
    sc = O.scene_from_tensors(t)
    sc.field.density_activation = "softplus" if activation == "softplus" else "exp"
    batches, want_d, want_r = [], torch.zeros(65), torch.zeros(195)
    for theta, (H, W) in ((2.0, (6, 8)), (0.4, (5, 9))):   # 48 + 45 rays: the second batch is not a multiple of 32
        cam = _camera(H, W, theta)
        o, d, _ = O.generate_rays(cam.camera_to_worlds[0], 0.9 * W, 0.9 * W, W / 2, H / 2, H, W)
        o, d = o.reshape(-1, 3), d.reshape(-1, 3)
        gd, gr = O.laplace_ggn_diag(sc, o, d)
        want_d += gd
        want_r += gr
        batches.append((o, d))
    with torch.cuda.device(dev):
        got_d, got_r = model.compute_hessian_naive(n_iters=5, ray_batches=batches)
    assert model.field.mlp_density_ggn is got_d and got_d.shape == (65,) and got_r.shape == (195,)
    # sums of squares of fp32 Jacobians; the sampler's parallel cumsum moves a few samples by an ulp
    torch.testing.assert_close(got_d.cpu(), want_d, rtol=2e-3, atol=1e-6 * want_d.max().item())
    torch.testing.assert_close(got_r.cpu(), want_r, rtol=2e-3, atol=1e-6 * want_r.max().item())
    # accumulates over calls: a second pass over the same batches doubles nothing in place (fresh buffers) ...
    with torch.cuda.device(dev):
        again_d, _ = model.compute_hessian_naive(n_iters=2, ray_batches=batches)
    assert torch.equal(again_d, got_d), "the reduction order is fixed: refitting is bit-reproducible"


def test_ensemble_aggregate_on_hip_moments(dev):
    from uncertainty_nerf_gs_amd import ensemble
    g = torch.Generator().manual_seed(12)
    members = []
    for _ in range(8):
        o = {"rgb": torch.rand(30, 41, 3, generator=g), "depth": torch.rand(30, 41, 1, generator=g) * 5,
             "expected_depth": torch.rand(30, 41, 1, generator=g) * 5, "accumulation": torch.rand(30, 41, 1, generator=g),
             "rgb_var": torch.rand(30, 41, 1, generator=g) * 0.1, "depth_var": torch.rand(30, 41, 1, generator=g)}
        o["rgb_std"], o["depth_std"] = o["rgb_var"].sqrt(), o["depth_var"].sqrt()
        members.append(o)
    ref = O.ensemble_aggregate(members)
    out = ensemble.aggregate([{k: v.to(dev) for k, v in m.items()} for m in members])
    assert set(out) == set(ref)
    for k, v in ref.items():
        torch.testing.assert_close(out[k].cpu(), v, rtol=2e-5, atol=1e-6, msg=k)
    plain = [{k: m[k] for k in ("rgb", "depth", "expected_depth", "accumulation")} for m in members]
    ref2 = O.ensemble_aggregate(plain)
    out2 = ensemble.aggregate([{k: v.to(dev) for k, v in m.items()} for m in plain])
    for k, v in ref2.items():
        torch.testing.assert_close(out2[k].cpu(), v, rtol=2e-5, atol=1e-6, msg=k)


def test_splat_model_get_outputs(dev):
    from uncertainty_nerf_gs_amd import models, splat, synthetic
    gp = synthetic.make_splat_tensors(7, 5000)
    gp["scales"] = gp["scales"] + 1.5
    m = models.ActiveSplatfactoModel(models.ActiveSplatfactoModelConfig(), num_points=10)
    m.load_state_dict({f"gauss_params.{k}": v for k, v in gp.items()})
    m.to(dev)
    H, W = 48, 64
    cam = models.Camera(synthetic.orbit_c2w(0.9, radius=2.5, height=0.5), 60.0, 60.0, W / 2, H / 2, H, W)
    out = m.get_outputs(cam)
    ref = splat.active_splatfacto_outputs({k: v.to(dev) for k, v in gp.items()}, cam.camera_to_worlds, 60.0, 60.0, W / 2,
                                          H / 2, H, W, splat.background_for("random").to(dev))   # config default
    assert torch.allclose(out["background"].cpu(), torch.tensor([0.1490, 0.1647, 0.2157]))
    assert set(out) == {"rgb", "depth", "accumulation", "background", "uncertainty", "rgb_var", "rgb_std", "depth_var",
                        "depth_std"}
    for k in ("rgb", "depth", "accumulation", "uncertainty", "depth_var"):
        assert torch.equal(out[k], ref[k]), k


def _nerfacto_upstream_state_dict(t, layout="encoder-mlp"):
    """synthetic weights (kind="mcdropout": 16-wide trunk) -> a plain `nerfacto` pipeline checkpoint in nerfstudio
    1.1.0's MLPWithHashEncoding layout (`model-sequential`: only the model.{0,1} alias names)"""
    f = t["field"]
    enc, mlp = ("model.0.", "model.1.") if layout == "model-sequential" else ("encoder.", "mlp.")
    sd = {f"field.mlp_base.{enc}hash_table": f["table"]}
    for i, (w, b) in enumerate(((f["w0"], f["b0"]), (f["w1"], f["b1"]))):
        sd[f"field.mlp_base.{mlp}layers.{i}.weight"], sd[f"field.mlp_base.{mlp}layers.{i}.bias"] = w, b
    for i in range(3):
        sd[f"field.mlp_head.layers.{i}.weight"], sd[f"field.mlp_head.layers.{i}.bias"] = f["head_w"][i], f["head_b"][i]
    sd["field.embedding_appearance.embedding.weight"] = f["appearance"][None].repeat(4, 1)
    for i, p in enumerate(t["props"]):
        sd[f"proposal_networks.{i}.mlp_base.{enc}hash_table"] = p["table"]
        for j, (w, b) in enumerate(((p["w0"], p["b0"]), (p["w1"], p["b1"]))):
            sd[f"proposal_networks.{i}.mlp_base.{mlp}layers.{j}.weight"] = w
            sd[f"proposal_networks.{i}.mlp_base.{mlp}layers.{j}.bias"] = b
    return {"_model." + k: v for k, v in sd.items()}


@pytest.mark.parametrize("layout", ["encoder-mlp", "model-sequential"])
def test_plain_nerfacto_member_from_an_upstream_layout_checkpoint(dev, layout):
    """ensemble_utils.py:149-150: ensembles are built from plain `nerfacto` runs.  NerfactoModel loaded from upstream's
    MLPWithHashEncoding key layout renders like the oracle's plain-nerfacto restatement."""
    from uncertainty_nerf_gs_amd import plugin, synthetic
    t = synthetic.make_scene_tensors(seed=12, kind="mcdropout", log2T=14, prop_log2T=12)
    # upstream's nerfacto hands config.average_init_density (0.01 in its method config) to the field AND the proposal nets
    t["field"]["average_init_density"] = 0.01
    t["field"]["b1"][0] += 4.6     # ... so shift the density logit to keep the synthetic scene as opaque as before
    cfg = _small_cfg(plugin.MODEL_CONFIGS["nerfacto"]())
    model = cfg._target(cfg, num_train_data=4)
    rep = model.load_state_dict(_nerfacto_upstream_state_dict(t, layout), strict=True)
    assert rep.unexpected_keys == [] and rep.loaded == rep.expected
    H, W = 36, 48
    cam = _camera(H, W, theta=1.4)
    with torch.cuda.device(dev):
        out = model.get_outputs_for_camera(cam)
    assert set(out) == {"rgb", "accumulation", "depth", "expected_depth", "prop_depth_0", "prop_depth_1"}
    o, d, _ = O.generate_rays(cam.camera_to_worlds[0], 0.9 * W, 0.9 * W, W / 2, H / 2, H, W)
    ref = O.nerfacto_outputs(O.scene_from_tensors(t), o.reshape(-1, 3), d.reshape(-1, 3))
    for k, atol, rtol, frac in (("rgb", 5e-5, 0, 0.0), ("accumulation", 2e-4, 0, 0.0), ("expected_depth", 0, 1e-3, 5e-3),
                                ("depth", 0, 1e-3, 2e-2)):
        got, want = out[k].reshape(H * W, -1).cpu().double(), ref[k].double()
        bad = (got - want).abs() > atol + rtol * want.abs()
        assert bad.double().mean() <= frac, (k, (got - want).abs().max().item())


def test_plain_splatfacto_member(dev):
    """ensemble_utils.py:153-154: splat ensembles are built from plain `splatfacto` runs (no log_uncertainties)"""
    from oracle import splat_oracle as SO
    from uncertainty_nerf_gs_amd import ensemble, plugin, synthetic
    gp = synthetic.make_splat_tensors(7, 3000)
    gp["scales"] = gp["scales"] + 1.5
    gp.pop("log_uncertainties")
    m = plugin.build_model("splatfacto", num_points=5)
    m.load_state_dict({f"_model.gauss_params.{k}": v for k, v in gp.items()}, strict=True)
    m.to(dev)
    H, W = 40, 56
    from uncertainty_nerf_gs_amd import models
    cam = models.Camera(synthetic.orbit_c2w(0.9, radius=2.5, height=0.5), 60.0, 60.0, W / 2, H / 2, H, W)
    out = m.get_outputs(cam)
    assert set(out) == {"rgb", "depth", "accumulation", "background"}
    bg = np.array([0.1490, 0.1647, 0.2157], np.float32)
    ref = SO.splatfacto_outputs({k: v.numpy() for k, v in gp.items()}, cam.camera_to_worlds.numpy(), 60.0, 60.0, W / 2, H / 2,
                                H, W, bg)
    for k, tol in (("rgb", 2e-5), ("depth", 2e-4), ("accumulation", 2e-5)):
        np.testing.assert_allclose(out[k].cpu().numpy(), ref[k], atol=tol, rtol=tol, err_msg=k)
    # two such members aggregate through the ensemble's no-std branch (rgb / depth std over the members)
    m2 = plugin.build_model("splatfacto", num_points=5)
    gp2 = {k: v + 0.01 * torch.randn(v.shape, generator=torch.Generator().manual_seed(1)) for k, v in gp.items()}
    m2.load_state_dict({f"gauss_params.{k}": v for k, v in gp2.items()})
    m2.to(dev)
    agg = ensemble.aggregate([out, m2.get_outputs(cam)])
    assert {"rgb_std", "depth_std"} <= set(agg) and agg["rgb_std"].shape == (H, W, 1)


def test_nerfstudio_plugin_model_renders_like_the_mirror(dev):
    """The nerfstudio-facing Model subclass of plugin.py (real nerfstudio when installed, else tests/stubs): built from
    the registered MethodSpecification, loaded from a reference-style pipeline checkpoint, called with a RayBundle --
    get_outputs_for_camera_ray_bundle(bundle) equals get_outputs_for_camera(camera) of the mirror bit for bit."""
    import os
    import sys
    from conftest import ROOT
    try:
        import nerfstudio  # noqa: F401
    except ImportError:
        sys.path.insert(0, os.path.join(ROOT, "tests", "stubs"))
    from nerfstudio.cameras.rays import RayBundle
    from uncertainty_nerf_gs_amd import models, ops, plugin, synthetic
    cfg = plugin.method_specifications()["active-nerfacto"].config.pipeline.model
    cfg.log2_hashmap_size = 12
    cfg.proposal_net_args_list = [dict(a, log2_hashmap_size=10) for a in cfg.proposal_net_args_list]
    torch.manual_seed(3)
    model = cfg.setup(scene_box=None, num_train_data=2)
    ref_model = plugin.build_model("active-nerfacto")        # the plain mirror with the same config values
    ref_model.config.log2_hashmap_size, ref_model.config.proposal_net_args_list = 12, cfg.proposal_net_args_list
    ref_model = type(ref_model)(ref_model.config, num_train_data=2)
    # a pipeline checkpoint as the reference writes it: `_model.` prefix (ensemble_pipeline.py:77-91)
    sd = {"_model." + k: (torch.randn_like(v) * (0.05 if "hash_table" not in k else 0.3) if v.is_floating_point() and "aabb" not in k else v)
          for k, v in ref_model.state_dict().items()}
    model.load_state_dict(sd)
    ref_model.load_state_dict(sd)
    H, W = 24, 40
    c2w = synthetic.orbit_c2w(0.4)
    cam = models.Camera(c2w, 40.0, 40.0, W / 2, H / 2, H, W)
    want = ref_model.to(dev).get_outputs_for_camera(cam)
    o, d, _ = ops.generate_rays(c2w, 40.0, 40.0, W / 2, H / 2, H, W, dev)
    got = model.to(dev).get_outputs_for_camera_ray_bundle(RayBundle(origins=o.view(H, W, 3), directions=d.view(H, W, 3)))
    assert set(got) == set(want) and "density" in got and got["density"].shape == (H, W, 48)
    for k in want:
        assert torch.equal(got[k], want[k]), k
    one = model.get_outputs(RayBundle(origins=o[:100], directions=d[:100]))
    assert one["rgb"].shape == (100, 3) and torch.isfinite(one["rgb_var"]).all()


def test_model_obb_box_equals_a_bundle_that_carries_the_box_planes(dev):
    """get_outputs_for_camera(camera, obb_box=box) [UPSTREAM Model.get_outputs_for_camera: generate_rays(obb_box=...)]
    == get_outputs_for_camera_ray_bundle(bundle with nears / fars from the same box): the two routes into
    render.crop_bins (unerf_ray_box_bins / unerf_ray_planes_bins) agree, and the ensemble pipeline hands the box on."""
    from types import SimpleNamespace
    from uncertainty_nerf_gs_amd import ensemble, ops, plugin, synthetic
    t = synthetic.make_scene_tensors(seed=3, kind="active", log2T=14, prop_log2T=12)
    cfg = _small_cfg(plugin.MODEL_CONFIGS["active-nerfacto"]())
    model = cfg._target(cfg, num_train_data=4)
    model.load_state_dict(_state_dict_from_tensors(t, "active"))
    H, W = 30, 44
    cam = _camera(H, W)
    th = 0.5
    box = SimpleNamespace(R=torch.tensor([[np.cos(th), -np.sin(th), 0.0], [np.sin(th), np.cos(th), 0.0], [0.0, 0.0, 1.0]],
                                         dtype=torch.float32),
                          T=torch.tensor([0.03, 0.02, -0.01]), S=torch.tensor([0.4, 0.5, 0.3]))
    with torch.cuda.device(dev):
        out = model.get_outputs_for_camera(cam, obb_box=box)
        plain = model.get_outputs_for_camera(cam)
        o, d, _ = ops.generate_rays(cam.camera_to_worlds[0], 0.9 * W, 0.9 * W, W / 2, H / 2, H, W, dev)
        nears, fars = O.intersect_obb(o.cpu(), d.cpu(), box.R, box.T, box.S)
        bundle = SimpleNamespace(origins=o.view(H, W, 3), directions=d.view(H, W, 3), nears=nears.view(H, W, 1).to(dev),
                                 fars=fars.view(H, W, 1).to(dev))
        via = model.get_outputs_for_camera_ray_bundle(bundle)
        ens = ensemble.EnsemblePipeline([model, model]).get_ensemble_outputs_for_camera_ray_bundle(cam, obb_box=box)
    hit = (fars > nears).view(H, W)
    assert 0.1 < hit.float().mean() < 0.9
    inner = hit.to(dev)
    assert (out["rgb"] - plain["rgb"])[inner].abs().max() > 1e-3           # the crop changes the picture
    assert out["accumulation"][~inner].abs().max() < 1e-6                   # outside the box: empty
    # planes computed on the device vs by the oracle differ by an ulp -> compare with a float tolerance
    for k, tol in (("rgb", 2e-5), ("accumulation", 1e-4)):
        assert (out[k] - via[k])[inner].abs().max() < tol, k
    assert torch.equal(ens["rgb"], out["rgb"]) and ens["rgb_var_epi"].abs().max() == 0     # two copies of one member


def test_model_loaded_from_a_run_directory_renders_like_its_source(dev, tmp_path):
    """checkpoints.load_model: the latest `step-*.ckpt` of a `nerfstudio_models` directory (pipeline state dict with
    the `_model.` prefix, next to optimizer state) into a fresh Model mirror -> the same image as the model it was
    saved from"""
    from uncertainty_nerf_gs_amd import checkpoints, plugin, synthetic
    t = synthetic.make_scene_tensors(seed=3, kind="active", log2T=14, prop_log2T=12)
    cfg = _small_cfg(plugin.MODEL_CONFIGS["active-nerfacto"]())
    src = cfg._target(cfg, num_train_data=4)
    src.load_state_dict(_state_dict_from_tensors(t, "active"))
    d = tmp_path / "run" / "nerfstudio_models"
    d.mkdir(parents=True)
    for step, scale in ((100, 0.5), (2000, 1.0)):      # the older checkpoint holds different weights
        sd = {"_model." + k: v * scale for k, v in src.state_dict().items()}
        torch.save({"step": step, "pipeline": sd, "optimizers": {}}, d / f"step-{step:09d}.ckpt")
    dst = cfg._target(cfg, num_train_data=4)
    path, step = checkpoints.load_model(dst, d)
    assert step == 2000 and path.name == "step-000002000.ckpt"
    cam = _camera(24, 32)
    with torch.cuda.device(dev):
        a, b = src.get_outputs_for_camera(cam), dst.get_outputs_for_camera(cam)
    for k in a:
        assert torch.equal(a[k], b[k]), k


def test_mcdropout_model_with_dropout_on_the_colour_heads_inputs(dev):
    """`rgb_dropout_layers=[0, -1]` (mcdropout_models.py:35, create_mlp utils.py:24-25): a Dropout in front of the colour
    head's first Linear.  The checkpoint's Sequential indices shift by one (Dropout, Linear, ReLU, Linear, ReLU, Dropout,
    Linear), the render goes through UNERF_DROP_HEADIN, and the frame matches the oracle with the same mask streams."""
    from uncertainty_nerf_gs_amd import plugin, synthetic
    t = synthetic.make_scene_tensors(seed=3, kind="mcdropout", log2T=14, prop_log2T=12)
    t["field"]["appearance"] = torch.linspace(-0.5, 0.7, 32)
    cfg = _small_cfg(plugin.MODEL_CONFIGS["nerfacto-mcdropout"]())
    cfg.rgb_dropout_layers = [0, -1]
    cfg.mc_samples = 4
    cfg.use_average_appearance_embedding = True
    model = cfg._target(cfg, num_train_data=4)
    sd = _state_dict_from_tensors(t, "mcdropout")
    for i, (old, new) in enumerate(((0, 1), (2, 3), (5, 6))):          # Linear modules of the shifted Sequential
        for leaf in ("weight", "bias"):
            sd[f"_model.field.mlp_head.{new}.{leaf}"] = {"weight": t["field"]["head_w"], "bias": t["field"]["head_b"]}[leaf][i]
    for k in [k for k in sd if k.startswith("_model.field.mlp_head.") and int(k.split(".")[3]) in (0, 2, 5)]:
        del sd[k]
    model.load_state_dict(sd, strict=True)
    model.seed = 77
    model.invalidate()
    H, W = 24, 32
    cam = _camera(H, W)
    with torch.cuda.device(dev):
        out = model.get_outputs_for_camera(cam)
    sc = O.scene_from_tensors(t)
    o, d, _ = O.generate_rays(cam.camera_to_worlds[0], 0.9 * W, 0.9 * W, W / 2, H / 2, H, W)
    ref = O.mcdropout_outputs(sc, o.reshape(-1, 3), d.reshape(-1, 3), 4, 77, 0.2, drop_sites=1 | 8 | 4)
    for k, atol in (("rgb", 5e-5), ("rgb_std", 5e-5), ("accumulation", 3e-4)):
        got, want = out[k].cpu().double().reshape(-1), ref[k].double().reshape(-1)
        assert float((got - want).abs().max()) <= atol, (k, float((got - want).abs().max()))
    plain = O.mcdropout_outputs(sc, o.reshape(-1, 3), d.reshape(-1, 3), 4, 77, 0.2, drop_sites=1 | 4)
    assert float((plain["rgb"] - ref["rgb"]).abs().max()) > 1e-3      # the input masks do change the picture


@pytest.mark.parametrize("method", ["nerfacto-mcdropout", "active-splatfacto"])
def test_repeated_renders_do_not_grow_device_memory(dev, method):
    """a render leaves nothing behind on the device: after a warm-up the allocator's live bytes are the same after 1 and
    after 25 more frames (side streams, pinned count buffers, overflow flags, per-frame scratch all return)"""
    from uncertainty_nerf_gs_amd import plugin, synthetic
    from uncertainty_nerf_gs_amd import models
    cfg = plugin.MODEL_CONFIGS[method]()
    if method == "active-splatfacto":
        gp = synthetic.make_splat_tensors(3, 3000)
        gp["scales"] = gp["scales"] + 1.5
        model = models.ActiveSplatfactoModel(models.ActiveSplatfactoModelConfig(), num_points=10)
        model.load_state_dict({f"gauss_params.{k}": v for k, v in gp.items()})
        model.to(dev)
    else:
        cfg = _small_cfg(cfg)
        cfg.mc_samples = 3
        model = cfg._target(cfg, num_train_data=4)
        t = synthetic.make_scene_tensors(seed=3, kind="mcdropout", log2T=14, prop_log2T=12)
        model.load_state_dict(_state_dict_from_tensors(t, "mcdropout"))
    cam = _camera(48, 64)
    with torch.cuda.device(dev):
        for _ in range(3):
            out = model.get_outputs_for_camera(cam)
        del out
        torch.cuda.synchronize()
        base = torch.cuda.memory_allocated()
        for i in range(25):
            out = model.get_outputs_for_camera(_camera(48, 64, theta=0.1 * i))
            assert torch.isfinite(out["rgb"]).all()
            del out
        torch.cuda.synchronize()
        assert torch.cuda.memory_allocated() <= base + (1 << 16), (base, torch.cuda.memory_allocated())
