"""Generate the reference-pinned golden vectors under tests/golden/.

Runs ONLY in the build container, where /root/reference is mounted.  It imports the
reference's own Python modules and records inputs + outputs as small data fixtures; nothing of
the reference's source text is copied.  The GPU box never runs this (no /root/reference there);
it only reads the .npz/.json files.

Importable as-is from the reference: utils.create_mlp, metrics.ause, metrics.auce.
Importable behind a stub shim for the absent third-party packages (nerfstudio, gsplat, backpack,
icecream ...): NerfactoLaplaceField.sample_laplace, laplace_model.ComputeWeightsModule,
EnsemblePipeline.get_ensemble_outputs_for_camera_ray_bundle,
NerfactoMCDropoutModel.get_outputs_for_camera_ray_bundle (aggregation part).

    python tests/golden/make_golden.py
"""
import importlib.abc
import importlib.machinery
import json
import os
import sys
import types
import warnings
from types import SimpleNamespace

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
STUB_ROOTS = ("nerfstudio", "gsplat", "backpack", "icecream", "tinycudann", "mediapy", "tyro", "cv2",
              "torchmetrics", "torchvision", "torchtyping", "imageio", "nerfacc")


class _StubModule(types.ModuleType):
    """Module whose every attribute is an empty, subclassable class (cached per name)."""
    __path__ = []

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        cls = type(name, (), {"__module__": self.__name__})
        setattr(self, name, cls)
        return cls


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        if fullname.split(".")[0] in STUB_ROOTS:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        return _StubModule(spec.name)

    def exec_module(self, module):
        pass


def install_stubs():
    sys.meta_path.insert(0, _StubFinder())
    sys.path.insert(0, REF)


def golden_create_mlp():
    from nerfuncertainty.utils import create_mlp
    from torch import nn
    cases = {
        "mcdropout_trunk": dict(in_dim=32, num_layers=2, layer_width=64, out_dim=16, activation=nn.ReLU,
                                out_activation=None, dropout_layers=[-1], dropout_rate=0.2),
        "mcdropout_head": dict(in_dim=63, num_layers=3, layer_width=64, out_dim=3, activation=nn.ReLU,
                               out_activation=nn.Sigmoid, dropout_layers=[-1], dropout_rate=0.2),
        "laplace_base": dict(in_dim=32, num_layers=1, layer_width=64, out_dim=64, activation=nn.ReLU,
                             out_activation=nn.ReLU),
        "laplace_head": dict(in_dim=63, num_layers=2, layer_width=64, out_dim=64, activation=nn.ReLU,
                             out_activation=nn.ReLU),
        "skip_and_mid_dropout": dict(in_dim=8, num_layers=4, layer_width=16, out_dim=2, skip_connections=(2,),
                                     activation=nn.ReLU, out_activation=None, dropout_layers=[1, 3],
                                     dropout_rate=0.5),
    }
    out = {}
    for name, kw in cases.items():
        m = create_mlp(**kw)
        mods = []
        for layer in (m if isinstance(m, nn.Sequential) else [m]):
            d = {"type": type(layer).__name__}
            if isinstance(layer, nn.Linear):
                d.update(in_features=layer.in_features, out_features=layer.out_features)
            if isinstance(layer, nn.Dropout):
                d.update(p=layer.p)
            mods.append(d)
        out[name] = {"kwargs": {k: (v.__name__ if isinstance(v, type) else v) for k, v in kw.items()}, "modules": mods}
    json.dump(out, open(os.path.join(OUT, "create_mlp.json"), "w"), indent=1)


def golden_metrics():
    from nerfuncertainty.metrics.ause import ause
    from nerfuncertainty.metrics.auce import auce
    g = torch.Generator().manual_seed(11)
    n = 4096
    err = torch.rand(n, generator=g) ** 2
    unc_good = err * (0.5 + torch.rand(n, generator=g))
    unc_bad = torch.rand(n, generator=g)
    res = {"err": err.numpy(), "unc_good": unc_good.numpy(), "unc_bad": unc_bad.numpy()}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for tag, unc in (("good", unc_good), ("bad", unc_bad)):
            for et in ("rmse", "mae", "mse"):
                ratio, e, ev, a = ause(unc, err, err_type=et)
                res[f"ause_{tag}_{et}"] = np.float64(a)
                res[f"ause_{tag}_{et}_curve"] = np.asarray(e, dtype=np.float64)
                res[f"ause_{tag}_{et}_curve_by_var"] = np.asarray(ev, dtype=np.float64)
        mean = torch.rand(n, generator=g).numpy()
        sigma = (0.05 + 0.2 * torch.rand(n, generator=g)).numpy()
        target = mean + sigma * torch.randn(n, generator=g).numpy() * 1.3
        d = auce(mean, sigma, target)
    res.update(auce_mean=mean, auce_sigma=sigma, auce_target=target)
    for k, v in d.items():
        res["auce_" + k] = np.asarray(v, dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "metrics.npz"), **res)


def golden_sample_laplace():
    from nerfuncertainty.models.laplace.laplace_field import NerfactoLaplaceField
    g = torch.Generator().manual_seed(5)
    res = {}
    for tag, out_dim, act in (("density", 1, torch.exp), ("rgb", 3, torch.sigmoid)):
        lin = torch.nn.Linear(64, out_dim)
        with torch.no_grad():
            lin.weight.copy_(torch.randn(out_dim, 64, generator=g) * 0.2)
            lin.bias.copy_(torch.randn(out_dim, generator=g) * 0.2)
        x = torch.randn(257, 64, generator=g)
        ggn = torch.rand(64 * out_dim + out_dim, generator=g) * 1e3
        mu_q = torch.nn.utils.parameters_to_vector(lin.parameters()).detach().clone()
        seed = 1234 + out_dim
        torch.manual_seed(seed)
        mu, var = NerfactoLaplaceField.sample_laplace(None, module=lin, activation=act, diag_ggn=ggn, input=x,
                                                      n_samples=100, prior_prec=1.0, eps=1e-9)
        res.update({f"{tag}_mu_q": mu_q.numpy(), f"{tag}_ggn": ggn.numpy(), f"{tag}_x": x.numpy(),
                    f"{tag}_seed": np.int64(seed), f"{tag}_mean": mu.detach().numpy(),
                    f"{tag}_var": var.detach().numpy()})
    np.savez_compressed(os.path.join(OUT, "sample_laplace.npz"), **res)


def golden_get_weights():
    from nerfuncertainty.models.laplace.laplace_model import ComputeWeightsModule
    g = torch.Generator().manual_seed(3)
    dens = torch.exp(torch.randn(64, 48, 1, generator=g) * 2.0)
    dens[5] = 0.0
    dens[6, 10:] = 1e6
    deltas = torch.rand(64, 48, 1, generator=g) * 0.2
    w = ComputeWeightsModule()(dens, deltas)
    np.savez_compressed(os.path.join(OUT, "get_weights.npz"), density=dens.numpy(), deltas=deltas.numpy(),
                        weights=w.numpy())


def _fake_member_outputs(g, with_std, M=5, H=6, W=7):
    outs = []
    for _ in range(M):
        o = {"rgb": torch.rand(H, W, 3, generator=g), "depth": torch.rand(H, W, 1, generator=g) * 5,
             "expected_depth": torch.rand(H, W, 1, generator=g) * 5, "accumulation": torch.rand(H, W, 1, generator=g)}
        if with_std:
            o["rgb_var"] = torch.rand(H, W, 1, generator=g) * 0.1
            o["rgb_std"] = o["rgb_var"].sqrt()
            o["depth_var"] = torch.rand(H, W, 1, generator=g)
            o["depth_std"] = o["depth_var"].sqrt()
        outs.append(o)
    return outs


def golden_ensemble():
    from nerfuncertainty.models.ensemble.ensemble_pipeline import EnsemblePipeline
    g = torch.Generator().manual_seed(9)
    res = {}
    for tag, with_std in (("plain", False), ("alea", True)):
        members = _fake_member_outputs(g, with_std)
        models = [SimpleNamespace(get_outputs_for_camera=(lambda cam, obb_box=None, o=o: o)) for o in members]
        fake = SimpleNamespace(models=models)
        out = EnsemblePipeline.get_ensemble_outputs_for_camera_ray_bundle(fake, None)
        for i, o in enumerate(members):
            for k, v in o.items():
                res[f"{tag}_in{i}_{k}"] = v.numpy()
        for k, v in out.items():
            res[f"{tag}_out_{k}"] = v.numpy()
    np.savez_compressed(os.path.join(OUT, "ensemble.npz"), **res)


def golden_mc_aggregate():
    import nerfstudio.models.nerfacto as stub_nerfacto
    from nerfuncertainty.models.mcdropout.mcdropout_models import NerfactoMCDropoutModel
    g = torch.Generator().manual_seed(21)
    K = 8
    passes = _fake_member_outputs(g, False, M=K)
    it = iter(passes)
    # the reference calls super().get_outputs_for_camera_ray_bundle K times; hand it the passes
    stub_nerfacto.NerfactoModel.get_outputs_for_camera_ray_bundle = lambda self, b: next(it)
    obj = object.__new__(NerfactoMCDropoutModel)
    obj.training = False
    obj.apply = lambda fn: None
    obj.eval = lambda: None
    obj.config = SimpleNamespace(mc_samples=K)
    out = obj.get_outputs_for_camera_ray_bundle(None)
    res = {}
    for i, o in enumerate(passes):
        for k, v in o.items():
            res[f"in{i}_{k}"] = v.numpy()
    for k, v in out.items():
        res[f"out_{k}"] = v.numpy()
    np.savez_compressed(os.path.join(OUT, "mc_aggregate.npz"), **res)


def golden_eval_configs():
    """field names and defaults of the eval script's configuration dataclasses (scripts/eval_configs.py)"""
    import dataclasses
    from nerfuncertainty.scripts import eval_configs as ec
    out = {}
    for name in ("EvalUncertainty", "LaplaceConfig", "EnsembleConfig", "MCDropoutConfig", "ActiveNerfactoConfig",
                 "ActiveSplatfactoConfig"):
        fields = {}
        for f in dataclasses.fields(getattr(ec, name)):
            d = None if f.default is dataclasses.MISSING else f.default
            fields[f.name] = {"required": f.default is dataclasses.MISSING, "default": str(d) if d is not None and not isinstance(d, (bool, int, float)) else d}
        out[name] = fields
    json.dump(out, open(os.path.join(OUT, "eval_configs.json"), "w"), indent=1)


if __name__ == "__main__":
    if not os.path.isdir(REF):
        raise SystemExit("/root/reference is not mounted: golden vectors can only be regenerated in the build container")
    install_stubs()
    for fn in (golden_create_mlp, golden_metrics, golden_sample_laplace, golden_get_weights, golden_ensemble,
               golden_mc_aggregate, golden_eval_configs):
        fn()
        print("wrote", fn.__name__)
