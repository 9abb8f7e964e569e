"""Checkpoint key layouts at the drop-in boundary: every layout the reference's stack
writes loads with every model parameter covered, and anything else fails loudly instead of rendering from random weights.

Layouts [UPSTREAM nerfstudio 1.1.0, the version the reference pins (README.md:23); the reference's own source shows
the fused module at activenerfacto_field.py:124-137]:
  MLPWithHashEncoding, torch:  `mlp_base.encoder.hash_table`, `mlp_base.mlp.layers.{i}.*` (+ `mlp_base.model.{0,1}.*`)
  MLPWithHashEncoding, tcnn:   `mlp_base.model.params` = ONE NetworkWithInputEncoding vector (MLP weights, then grid)
  HashEncoding + Sequential:   `encoding.*` / `mlp_base.{0,1}.*` (older upstream; the reference's own split fields)
"""
import pickle
import warnings

import pytest
import torch

from uncertainty_nerf_gs_amd import checkpoints as C
from uncertainty_nerf_gs_amd import fields as F
from uncertainty_nerf_gs_amd import models as M
from uncertainty_nerf_gs_amd import plugin


def _small(cfg, implementation="torch"):
    cfg.log2_hashmap_size = 6
    cfg.implementation = implementation
    cfg.proposal_net_args_list = [dict(a, log2_hashmap_size=5) for a in cfg.proposal_net_args_list]
    return cfg


def _model(method, implementation="torch", **cfg_kw):
    cfg = _small(plugin.MODEL_CONFIGS[method](), implementation)
    for k, v in cfg_kw.items():
        setattr(cfg, k, v)
    return cfg._target(cfg, num_train_data=3)


def _params(model):
    """one (canonical name, tensor) per distinct parameter"""
    return {n: p.detach().clone() for n, p in model.named_parameters()}


def _fill(model):
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for p in model.parameters():
            p.copy_(torch.randn(p.shape, generator=g))
    return _params(model)


def _assert_loaded(dst, want):
    got = _params(dst)
    assert set(got) == set(want)
    for k in want:
        assert torch.equal(got[k], want[k]), k


def _upstream_extras(sd):
    """what a real nerfstudio checkpoint carries besides the model's tensors"""
    sd = dict(sd)
    sd["_model.camera_optimizer.pose_adjustment"] = torch.zeros(3, 6)
    sd["_model.lpips.net.lin0.model.1.weight"] = torch.zeros(1, 64, 1, 1)
    sd["_model.device_indicator_param"] = torch.empty(0)
    sd["datamanager.train_camera_optimizer.pose_adjustment"] = torch.zeros(3, 6)
    sd["_model.field.direction_encoding.tcnn_encoding.params"] = torch.empty(0)     # parameterless tcnn modules
    sd["_model.field.position_encoding.tcnn_encoding.params"] = torch.empty(0)
    return sd


@pytest.mark.parametrize("method", ["nerfacto", "active-nerfacto", "nerfacto-mcdropout", "nerfacto-laplace"])
def test_own_state_dict_round_trips_with_nothing_missing_or_unexpected(method):
    src, dst = _model(method), _model(method)
    want = _fill(src)
    sd = _upstream_extras({"_model." + k: v for k, v in src.state_dict().items()})
    with warnings.catch_warnings():
        warnings.simplefilter("error")                      # no "ignored keys" warning either
        rep = dst.load_state_dict(sd, strict=True)
    assert rep.missing_keys == [] and rep.unexpected_keys == [] and rep.loaded == rep.expected == len(want)
    _assert_loaded(dst, want)


@pytest.mark.parametrize("layout", ["encoder-mlp", "model-sequential", "encoding-sequential"])
def test_proposal_networks_and_plain_nerfacto_load_from_every_torch_layout(layout):
    """upstream's MLPWithHashEncoding names, its `model.{0,1}` aliases alone, and the older HashEncoding + Sequential
    names -- for the proposal networks of every method and for the plain nerfacto field"""
    src, dst = _model("nerfacto"), _model("nerfacto")
    want = _fill(src)
    sd = {}
    for k, v in src.state_dict().items():
        if ".mlp_base.model." in k:
            continue                                        # aliases: re-created below per layout
        for holder in ("field", "proposal_networks.0", "proposal_networks.1"):
            pre = holder + ".mlp_base."
            if not k.startswith(pre):
                continue
            rest = k[len(pre):]
            if layout == "model-sequential":
                k = pre + ("model.0." + rest[len("encoder."):] if rest.startswith("encoder.") else "model.1." + rest[len("mlp."):])
            elif layout == "encoding-sequential":
                k = (holder + ".encoding." + rest[len("encoder."):]) if rest.startswith("encoder.") else pre + "1." + rest[len("mlp."):]
        sd["_model." + k] = v
    assert any(("model.0" if layout == "model-sequential" else "encoding." if layout == "encoding-sequential" else "encoder.") in k
               for k in sd)
    rep = dst.load_state_dict(sd, strict=True)
    assert rep.unexpected_keys == [] and rep.loaded == len(want)
    _assert_loaded(dst, want)


@pytest.mark.parametrize("fused_key", ["model.params", "tcnn_encoding.params"])
def test_fused_tcnn_vector_is_split_into_mlp_weights_then_grid(fused_key):
    """tcnn.NetworkWithInputEncoding keeps ONE parameter vector: FullyFusedMLP weights first, HashGrid parameters after
    [UPSTREAM-RECALL tiny-cuda-nn]; upstream's attribute is `model` (-> `mlp_base.model.params`)"""
    src, dst = _model("nerfacto", "tcnn"), _model("nerfacto", "tcnn")
    want = _fill(src)
    sd = {}
    for k, v in src.state_dict().items():
        if ".mlp_base.encoder." in k or ".mlp_base.mlp." in k:
            continue
        sd["_model." + k] = v
    for holder in ("field", "proposal_networks.0", "proposal_networks.1"):
        m = dict(src.named_modules())[holder + ".mlp_base"]
        sd[f"_model.{holder}.mlp_base.{fused_key}"] = torch.cat([m.mlp.tcnn_encoding.params.detach(), m.encoder.tcnn_encoding.params.detach()])
    rep = dst.load_state_dict(sd, strict=True)
    assert rep.unexpected_keys == [] and rep.loaded == len(want)
    _assert_loaded(dst, want)
    # the unpacked proposal MLP is what the kernels get: [16, pad16(10)] then [pad16(1), 16], padding dropped
    (w0, _), (w1, _) = dst.proposal_networks[0].mlp_base.mlp.linear_layers()
    vec = want["proposal_networks.0.mlp_base.mlp.tcnn_encoding.params"]
    assert torch.equal(w0, vec[:256].reshape(16, 16)[:, :10]) and torch.equal(w1, vec[256:].reshape(16, 16)[:1])
    # a vector of another configuration is refused with the sizes spelled out
    bad = dict(sd)
    bad[f"_model.field.mlp_base.{fused_key}"] = torch.zeros(123)
    with pytest.raises(RuntimeError, match=r"123 values, expected \d+ \(FullyFusedMLP\) \+ \d+ \(HashGrid\)"):
        dst.load_state_dict(bad)


def test_checkpoint_of_the_other_implementation_is_refused():
    """a tcnn-implementation checkpoint into a torch-implementation model (and the reverse) used to load
    'successfully' with every tensor left at its random initial value"""
    for a, b in (("tcnn", "torch"), ("torch", "tcnn")):
        src, dst = _model("active-nerfacto", a), _model("active-nerfacto", b)
        before = _params(dst)
        with pytest.raises(RuntimeError, match="parameters are not in the checkpoint"):
            dst.load_state_dict({"_model." + k: v for k, v in src.state_dict().items()})
        _assert_loaded(dst, before)                          # and nothing was half-loaded


def test_unknown_layout_and_other_methods_are_refused_and_extras_are_reported():
    src, dst = _model("nerfacto-mcdropout"), _model("active-nerfacto")
    with pytest.raises(RuntimeError, match="not in the checkpoint"):       # another method's field
        dst.load_state_dict({"_model." + k: v for k, v in src.state_dict().items()})
    with pytest.raises(RuntimeError, match="not in the checkpoint"):       # the probe of VERDICT r2
        dst.load_state_dict({"_model.proposal_networks.0.mlp_base.encoder.hash_table": torch.zeros(5 << 5, 2)})
    with pytest.raises(RuntimeError, match="not in the checkpoint"):
        dst.load_state_dict({})
    # a config mismatch names the key and both shapes
    big = _model("active-nerfacto", log2_hashmap_size=7)
    with pytest.raises(RuntimeError, match=r"field\.mlp_base_grid\.hash_table has shape \(2048, 2\) in the checkpoint, \(1024, 2\)"):
        dst.load_state_dict({"_model." + k: v for k, v in big.state_dict().items()})
    # model keys this build has no use for: reported, a RuntimeError under strict=True (nerfstudio then retries non-strict)
    good = {"_model." + k: v for k, v in _model("active-nerfacto").state_dict().items()}
    good["_model.field.mlp_pred_normals.layers.0.weight"] = torch.zeros(64, 31)
    with pytest.raises(RuntimeError, match="unexpected keys"):
        dst.load_state_dict(good, strict=True)
    with pytest.warns(UserWarning, match="ignored checkpoint keys"):
        rep = dst.load_state_dict(good)
    assert rep.unexpected_keys == ["field.mlp_pred_normals.layers.0.weight"]


def test_mcdropout_without_density_dropout_keeps_the_parents_fused_trunk():
    """density_dropout_layers=False (mcdropout_fields.py:112, :162-166): the checkpoint has upstream's
    field.mlp_base.encoder / .mlp names and no field.mlp_base_grid"""
    src = _model("nerfacto-mcdropout", density_dropout_layers=False)
    dst = _model("nerfacto-mcdropout", density_dropout_layers=False)
    want = _fill(src)
    assert "field.mlp_base.encoder.hash_table" in want and "field.mlp_base.mlp.layers.1.weight" in want
    assert not any("mlp_base_grid" in k for k in want)
    dst.load_state_dict({"_model." + k: v for k, v in src.state_dict().items()}, strict=True)
    _assert_loaded(dst, want)
    with pytest.raises(RuntimeError, match="not in the checkpoint"):     # ... and is not the default model's layout
        _model("nerfacto-mcdropout").load_state_dict({"_model." + k: v for k, v in src.state_dict().items()})


def test_splat_members_plain_and_active():
    plain = plugin.build_model("splatfacto", num_points=7)
    active = plugin.build_model("active-splatfacto", num_points=4)
    assert set(plain.gauss_params) == {"means", "scales", "quats", "features_dc", "features_rest", "opacities"}
    assert set(active.gauss_params) == set(plain.gauss_params) | {"log_uncertainties"}
    ck = {"_model.gauss_params." + k: torch.randn((11,) + tuple(v.shape[1:])) for k, v in plain.gauss_params.items()}
    rep = plain.load_state_dict(ck, strict=True)
    assert rep.loaded == 6 and plain.gauss_params["means"].shape == (11, 3) and plain.step == 30000
    assert torch.equal(plain.gauss_params["opacities"].detach(), ck["_model.gauss_params.opacities"])
    # a plain splatfacto run is not an active-splatfacto checkpoint: log_uncertainties would stay random
    with pytest.raises(RuntimeError, match="gauss_params.log_uncertainties"):
        active.load_state_dict(ck)
    # the reverse direction reports the extra tensor
    ck["_model.gauss_params.log_uncertainties"] = torch.rand(11, 1)
    active.load_state_dict(ck, strict=True)
    with pytest.warns(UserWarning, match="log_uncertainties"):
        rep = plain.load_state_dict(ck)
    assert rep.unexpected_keys == ["gauss_params.log_uncertainties"]
    with pytest.raises(RuntimeError, match="no gauss_params.means"):
        plain.load_state_dict({"_model.field.x": torch.zeros(1)})


def test_load_model_propagates_and_reads_pickles_safely(tmp_path, monkeypatch):
    src, dst = _model("active-nerfacto"), _model("active-nerfacto")
    want = _fill(src)
    d = tmp_path / "run" / "nerfstudio_models"
    d.mkdir(parents=True)
    torch.save({"step": 30, "pipeline": {"_model." + k: v for k, v in src.state_dict().items()}}, d / "step-000000030.ckpt")
    path, step = C.load_model(dst, d)
    assert step == 30 and dst.last_load_report.loaded == len(want)
    _assert_loaded(dst, want)
    # a run of another method in the directory: the error names the file
    other = _model("nerfacto-laplace")
    torch.save({"step": 40, "pipeline": {"_model." + k: v for k, v in other.state_dict().items()}}, d / "step-000000040.ckpt")
    with pytest.raises(RuntimeError, match=r"step-000000040\.ckpt.*not in the checkpoint"):
        C.load_model(dst, d)

    class Odd:                                     # an object the safe unpickler refuses
        pass

    import __main__
    monkeypatch.setattr(__main__, "Odd", Odd, raising=False)
    Odd.__module__, Odd.__qualname__ = "__main__", "Odd"
    torch.save({"step": 50, "pipeline": {}, "config": Odd()}, d / "step-000000050.ckpt")
    monkeypatch.delenv("UNERF_TRUST_CHECKPOINT_PICKLE", raising=False)
    with pytest.raises(pickle.UnpicklingError, match="trust_pickle=True"):
        C.read_pipeline_state(d / "step-000000050.ckpt")
    sd, step = C.read_pipeline_state(d / "step-000000050.ckpt", trust_pickle=True)
    assert step == 50 and sd == {}
    (d / "step-000000060.ckpt").write_bytes(b"not a checkpoint")
    with pytest.raises(Exception) as ei:           # corruption is reported as such, not retried unsafely
        C.read_pipeline_state(d / "step-000000060.ckpt", trust_pickle=True)
    assert not isinstance(ei.value, KeyError)


def test_ensemble_member_configs_exist_for_every_method_the_reference_ensembles():
    """ensemble_utils.py:149-156: nerfacto, active-nerfacto, splatfacto, active-splatfacto"""
    for name, cls in (("nerfacto", M.NerfactoModel), ("active-nerfacto", M.ActiveNerfactoModel),
                      ("splatfacto", M.SplatfactoModel), ("active-splatfacto", M.ActiveSplatfactoModel)):
        cfg = plugin.MODEL_CONFIGS[name]()
        assert cfg._target is cls
    assert plugin.MODEL_CONFIGS["nerfacto"]().average_init_density == 0.01
    m = _model("nerfacto")
    assert isinstance(m.field, F.NerfactoField) and m.field.average_init_density == 0.01
    assert M.NerfactoModelConfig().proposal_initial_sampler == "piecewise" and M.NerfactoModelConfig().background_color == "last_sample"


def test_an_alias_next_to_its_canonical_key_must_agree():
    """a checkpoint that carries BOTH `n.0.hash_table` (older Sequential layout) and `n.encoder.hash_table` with
    different tensors used to keep the canonical one and drop the alias silently; equal tensors are fine, different ones
    raise, and keys under the alias prefixes that are not one of the modules' leaves are left alone (unexpected keys)."""
    m = _model("nerfacto")
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    canon = "proposal_networks.0.mlp_base.encoder.hash_table"
    assert canon in sd
    both = dict(sd)
    both["proposal_networks.0.mlp_base.0.hash_table"] = sd[canon].clone()
    M.load_checked(m, both, strict=True)                                       # the same tensor twice: loads
    both["proposal_networks.0.mlp_base.0.hash_table"] = sd[canon] + 1.0
    with pytest.raises(RuntimeError, match="different contents"):
        M.load_checked(m, both)
    odd = dict(sd)
    odd["proposal_networks.0.mlp_base.0.running_mean"] = torch.zeros(3)        # not a leaf these modules own
    rep = M.load_checked(m, odd)
    assert "proposal_networks.0.mlp_base.0.running_mean" in rep.unexpected_keys
