"""The nerfstudio plugin boundary (uncertainty-nerf-gs_amd/plugin.py + pyproject.toml) without a GPU: the four entry
points of the reference (/root/reference/pyproject.toml:18-22) resolve to full MethodSpecifications whose model
configs instantiate nerfstudio `Model` subclasses; state-dict names follow the reference; the rendering methods take
nerfstudio's argument types (RayBundle, camera, obb_box) and forward to the HIP mirrors.  Uses the real nerfstudio when
it is importable, else the stand-in of tests/stubs/ (README there)."""
import importlib
import math
import os
import sys

import pytest
import torch

from conftest import ROOT

try:
    import nerfstudio  # noqa: F401
except ImportError:
    sys.path.insert(0, os.path.join(ROOT, "tests", "stubs"))

try:
    import tomllib as _toml
except ImportError:
    import tomli as _toml


def _entry_points():
    with open(os.path.join(ROOT, "pyproject.toml"), "rb") as f:
        return _toml.load(f)["project"]["entry-points"]["nerfstudio.method_configs"]


def test_pyproject_registers_the_references_four_entry_points():
    eps = _entry_points()
    assert set(eps) == {"dropout", "laplace_d", "activenerfacto", "activesplatfacto"}     # reference pyproject.toml:19-22
    from uncertainty_nerf_gs_amd import plugin
    for name, target in eps.items():
        mod, attr = target.split(":")
        assert mod == "uncertainty_nerf_gs_amd.plugin" and plugin.ENTRY_POINTS[name][0] == attr
        spec = getattr(importlib.import_module(mod), attr)                              # what nerfstudio's registry does
        assert spec.config.method_name == plugin.ENTRY_POINTS[name][1]
        assert spec.description == plugin.DESCRIPTIONS[spec.config.method_name]


def test_method_specifications_carry_the_references_trainer_and_model_config():
    from uncertainty_nerf_gs_amd import plugin
    specs = plugin.method_specifications()
    assert set(specs) == set(plugin.METHOD_NAMES)
    for name in ("nerfacto-mcdropout", "nerfacto-laplace", "active-nerfacto"):
        c = specs[name].config
        # mcdropout_configs.py:19-52 / laplace_config.py:23-56 / activenerfacto_config.py:26-59
        assert (c.steps_per_eval_batch, c.steps_per_save, c.max_num_iterations, c.mixed_precision, c.vis) == (500, 2000, 30000, True, "viewer")
        assert c.pipeline.datamanager.train_num_rays_per_batch == 4096 and c.pipeline.datamanager.eval_num_rays_per_batch == 4096
        m = c.pipeline.model
        assert m.eval_num_rays_per_chunk == 1 << 15 and m.average_init_density == 0.01 and m.camera_optimizer.mode == "SO3xR3"
        assert set(c.optimizers) == {"proposal_networks", "fields", "camera_opt"}
        assert c.optimizers["fields"]["optimizer"].lr == 1e-2 and c.optimizers["fields"]["scheduler"].max_steps == 200000
        assert c.viewer.num_rays_per_chunk == 1 << 15
    s = specs["active-splatfacto"].config                                               # activesplatfacto_config.py:33-90
    assert (s.steps_per_eval_image, s.steps_per_eval_batch, s.steps_per_eval_all_images, s.mixed_precision) == (100, 0, 1000, False)
    assert s.pipeline.datamanager.cache_images_type == "uint8" and s.pipeline.datamanager.dataparser.load_3D_points is True
    assert set(s.optimizers) == {"means", "features_dc", "features_rest", "opacities", "scales", "quats", "log_uncertainties", "camera_opt"}
    assert s.optimizers["log_uncertainties"]["optimizer"].lr == 0.0025 and s.optimizers["means"]["scheduler"].lr_final == 1.6e-6
    # config fields of the reference's model configs that shape eval rendering
    mc = specs["nerfacto-mcdropout"].config.pipeline.model
    assert (mc.mc_samples, mc.dropout_rate, mc.rgb_dropout_layers, mc.density_dropout_layers) == (10, 0.2, [-1], True)  # mcdropout_models.py:37-48
    assert specs["active-nerfacto"].config.pipeline.model.beta_min == 0.01
    sp = s.pipeline.model
    assert (sp.sh_degree, sp.background_color, sp.beta_min, sp.rasterize_mode) == (3, "random", 0.01, "classic")


def _small(cfg):
    cfg.log2_hashmap_size, cfg.num_levels, cfg.max_res = 8, 16, 64
    cfg.proposal_net_args_list = [dict(a, log2_hashmap_size=6) for a in cfg.proposal_net_args_list]
    return cfg


@pytest.mark.parametrize("name", ["nerfacto-mcdropout", "nerfacto-laplace", "active-nerfacto"])
def test_nerf_model_is_a_nerfstudio_model_with_the_references_state_dict_names(name):
    from nerfstudio.models.base_model import Model
    from uncertainty_nerf_gs_amd import plugin
    cfg = _small(plugin.method_specifications()[name].config.pipeline.model)
    model = cfg.setup(scene_box=None, num_train_data=3)          # what VanillaPipeline does with config.model
    assert isinstance(model, Model) and type(model).__name__ == {"nerfacto-mcdropout": "NerfactoMCDropoutModel",
                                                               "nerfacto-laplace": "NerfactoLaplaceModel",
                                                               "active-nerfacto": "ActiveNerfactoModel"}[name]
    keys = set(model.state_dict())
    assert any(k.startswith("proposal_networks.0.") for k in keys) and any(k.startswith("proposal_networks.1.") for k in keys)
    want = {"nerfacto-mcdropout": ["field.mlp_base_grid.hash_table", "field.mlp_base.0.weight", "field.mlp_base.3.weight",
                                   "field.mlp_head.0.weight", "field.mlp_head.5.bias"],
            "nerfacto-laplace": ["field.base_grid.hash_table", "field.base_mlp.0.weight", "field.mlp_density.weight",
                                 "field.mlp_hidden.weight", "field.mlp_rgb_ll.weight"],
            "active-nerfacto": ["field.mlp_base_grid.hash_table", "field.mlp_base_mlp.layers.0.weight", "field.mlp_head.layers.2.bias",
                                "field.embedding_appearance.embedding.weight"]}[name]
    for k in want:
        assert k in keys, (k, sorted(keys)[:40])
    groups = model.get_param_groups()
    assert set(groups) == {"proposal_networks", "fields"} and len(groups["fields"]) > 0
    assert model.get_training_callbacks(None) == []
    # loading a nerfstudio pipeline checkpoint (keys carry the `_model.` prefix): the mirror's tensors are the model's
    sd = {"_model." + k: torch.full_like(v, 0.25) for k, v in model.state_dict().items()
          if k.startswith(("field.", "proposal_networks.")) and v.is_floating_point()}
    res = model.load_state_dict(sd, strict=True)      # nerfstudio's pipeline passes strict=True first
    assert res.missing_keys == [] and res.unexpected_keys == [] and res.loaded == res.expected > 0
    assert float(model.field.state_dict()[want[0][len("field."):]].flatten()[0]) == 0.25
    # a checkpoint that does not cover the model (here: no proposal networks) must not load "successfully"
    with pytest.raises(RuntimeError, match="not in the checkpoint"):
        model.load_state_dict({k: v for k, v in sd.items() if k.startswith("_model.field.")})
    assert model._mirror.field is model.field
    with pytest.raises(NotImplementedError):
        model.get_loss_dict({}, {})


def test_rendering_methods_take_nerfstudio_types_and_forward_to_the_mirror(monkeypatch):
    """RayBundle / camera / obb_box arrive at the mirror's rendering entry points unchanged (no GPU here: the
    mirror's render functions are replaced by recorders)."""
    from nerfstudio.cameras.rays import RayBundle
    from uncertainty_nerf_gs_amd import models, plugin, render
    cfg = _small(plugin.method_specifications()["nerfacto-mcdropout"].config.pipeline.model)
    cfg.mc_samples = 8
    model = cfg.setup(scene_box=None, num_train_data=1)
    seen = {}

    def fake_render_rays(scene, o, d, **kw):
        seen["rays"] = (o.shape, d.shape, kw.get("image_width"), kw.get("total_rays"))
        return {"rgb": torch.zeros(o.shape[0], 3)}

    monkeypatch.setattr(render, "render_rays", fake_render_rays)
    monkeypatch.setattr(models._NerfactoBase, "device_scene",
                        lambda self, device=None: type("S", (), {"device": torch.device("cpu"), "chunk_rays": 1 << 15,
                                                                 "field": type("F", (), {"seed": 0, "use_mfma": False})()})())
    H, W = 6, 9
    bundle = RayBundle(origins=torch.zeros(H, W, 3), directions=torch.ones(H, W, 3))
    out = model.get_outputs_for_camera_ray_bundle(bundle)
    assert out["rgb"].shape == (H, W, 3) and seen["rays"] == ((H * W, 3), (H * W, 3), W, H * W)
    flat = RayBundle(origins=torch.zeros(11, 3), directions=torch.ones(11, 3))
    assert model.get_outputs(flat)["rgb"].shape == (11, 3) and model(flat)["rgb"].shape == (11, 3)

    def fake_render_camera(scene, c2w, **kw):
        seen["camera"] = (tuple(c2w.shape), kw["H"], kw["W"])
        seen["obb"] = kw.get("obb")
        return {"rgb": torch.zeros(kw["H"], kw["W"], 3)}

    monkeypatch.setattr(render, "render_camera", fake_render_camera)
    cam = models.Camera(torch.eye(4)[None, :3], torch.tensor([[50.0]]), torch.tensor([[50.0]]), torch.tensor([[4.5]]),
                        torch.tensor([[3.0]]), torch.tensor([[H]]), torch.tensor([[W]]))
    assert model.get_outputs_for_camera(cam, obb_box=None)["rgb"].shape == (H, W, 3) and seen["camera"] == ((3, 4), H, W)
    assert seen["obb"] is None
    # an OrientedBox (R, T, S) becomes the inverse rigid transform + the edge lengths render.crop_bins takes
    th = 0.3
    Rm = torch.tensor([[math.cos(th), -math.sin(th), 0.0], [math.sin(th), math.cos(th), 0.0], [0.0, 0.0, 1.0]])
    box = type("Box", (), {"R": Rm, "T": torch.tensor([0.5, -0.25, 1.0]), "S": torch.tensor([1.0, 2.0, 3.0])})()
    model.get_outputs_for_camera(cam, obb_box=box)
    w2b, S = seen["obb"]
    assert torch.allclose(w2b[:, :3], Rm.T, atol=1e-6) and torch.allclose(w2b[:, 3], -(Rm.T @ box.T), atol=1e-6)
    assert torch.equal(S, box.S)
    # a bundle that already carries planes keeps them (SceneCollider.forward): they become first-level bins
    monkeypatch.setattr(render, "crop_bins", lambda scene, o, d, obb=None, nears=None, fars=None:
                        seen.__setitem__("planes", (nears.shape, fars.shape)) or torch.zeros(o.shape[0], 257))
    monkeypatch.setattr(render, "render_rays", lambda scene, o, d, **kw:
                        seen.__setitem__("init", kw["init_bins"].shape) or {"rgb": torch.zeros(o.shape[0], 3)})
    b2 = RayBundle(origins=torch.zeros(H, W, 3), directions=torch.ones(H, W, 3), nears=torch.zeros(H, W, 1),
                   fars=torch.ones(H, W, 1))
    model.get_outputs_for_camera_ray_bundle(b2)
    assert seen["planes"] == ((H * W,), (H * W,)) and seen["init"] == (H * W, 257)


def test_splat_model_plugin_surface():
    from nerfstudio.models.base_model import Model
    from uncertainty_nerf_gs_amd import plugin
    cfg = plugin.method_specifications()["active-splatfacto"].config.pipeline.model
    pts = torch.rand(37, 3)
    model = cfg.setup(scene_box=None, num_train_data=2, seed_points=(pts, torch.zeros(37, 3)))
    assert isinstance(model, Model)
    names = ["means", "scales", "quats", "features_dc", "features_rest", "opacities", "log_uncertainties"]
    assert set(model.state_dict()) >= {f"gauss_params.{n}" for n in names}           # activesplatfacto_model.py:72, 93
    assert torch.equal(model.gauss_params["means"].detach(), pts) and model.gauss_params["log_uncertainties"].shape == (37, 1)
    assert set(model.get_param_groups()) == set(names)
    # load_state_dict resizes every parameter to the checkpoint's point count and pins step = 30000 (:87-100)
    ck = {f"_model.gauss_params.{n}": torch.zeros((5,) + tuple(model.gauss_params[n].shape[1:])) for n in names}
    model.load_state_dict(ck)
    assert model.gauss_params["means"].shape == (5, 3) and model._mirror.step == 30000
    assert torch.allclose(model._mirror.background_color, torch.tensor([0.1490, 0.1647, 0.2157]))


def test_plugin_needs_nerfstudio_only_for_the_registry_objects():
    from uncertainty_nerf_gs_amd import plugin
    assert plugin.build_model("active-nerfacto").config.eval_num_rays_per_chunk == 1 << 15
    with pytest.raises(AttributeError):
        plugin.NoSuchThing


def test_ns_config_annotations_are_resolved_types_not_strings():
    """models.py uses postponed annotations; the dynamic nerfstudio ModelConfig classes must carry evaluated types (tyro /
    typing.get_type_hints resolve a dynamic class's strings in plugin.py's namespace, where they resolved only by luck)"""
    import dataclasses
    import typing
    from uncertainty_nerf_gs_amd import plugin
    for name in ("NerfactoMCDropoutModelConfig", "ActiveNerfactoModelConfig", "NerfactoLaplaceModelConfig", "ActiveSplatfactoModelConfig"):
        cls = getattr(plugin, name)
        ann = {}
        for k in reversed(cls.__mro__):
            ann.update(getattr(k, "__annotations__", {}))
        own = {f.name for f in dataclasses.fields(cls)}
        for fname in ("background_color", "rasterize_mode", "proposal_net_args_list", "num_proposal_samples_per_ray", "mc_samples"):
            if fname in own:
                assert not isinstance(ann[fname], str), (name, fname, ann[fname])
        typing.get_type_hints(cls)      # and the whole class resolves without this module's globals
    nerf = plugin.NerfactoMCDropoutModelConfig
    assert {"proposal_initial_sampler", "background_color"} <= {f.name for f in dataclasses.fields(nerf)}
