"""Seeded synthetic workloads: no dataset, no checkpoint, no network.

Everything here is plain torch-CPU tensor generation; the result is a dict of tensors in the
torch nn.Linear layout ([out,in]) that both the device path (`scene_to_device`) and the CPU
oracle (`oracle.nerf_oracle.scene_from_tensors`) are built from, so they see identical weights.
"""
from __future__ import annotations

import math
from typing import Dict, List, Tuple

import numpy as np
import torch

from . import lib as _l
from . import ops
from .render import NerfSceneDev


def hash_scalings(num_levels: int, min_res: int, max_res: int) -> torch.Tensor:
    """nerfstudio HashEncoding.__init__: floor(min_res * growth**levels) with the growth factor a
    numpy float64 raised to an int64 torch tensor (-> float32 pow; a 16..2048 grid ends at 2047)."""
    levels = torch.arange(num_levels)
    growth = np.exp((np.log(max_res) - np.log(min_res)) / (num_levels - 1)) if num_levels > 1 else 1
    return torch.floor(min_res * growth ** levels).to(torch.float32)


def _linear(gen, out_dim, in_dim):
    bound = 1.0 / math.sqrt(in_dim)
    w = (torch.rand(out_dim, in_dim, generator=gen) * 2 - 1) * bound
    b = (torch.rand(out_dim, generator=gen) * 2 - 1) * bound
    return w, b


def _grid(gen, num_levels, min_res, max_res, log2T, table_scale, features=2):
    T = 1 << log2T
    table = (torch.rand(num_levels * T, features, generator=gen) * 2 - 1) * table_scale
    return {"table": table, "scalings": hash_scalings(num_levels, min_res, max_res), "log2T": log2T}


def _grid_tcnn(gen, num_levels, min_res, max_res, log2T, table_scale):
    """tcnn-layout grid as nerfstudio's HashEncoding(implementation="tcnn") sets it up: per_level_scale =
    exp((ln max_res - ln min_res) / (L - 1)); `table` is the flat parameter vector viewed as [rows, 2]."""
    growth = math.exp((math.log(max_res) - math.log(min_res)) / (num_levels - 1))
    levels = ops.tcnn_grid_levels(num_levels, min_res, growth, log2T)
    rows = levels[-1][2] + levels[-1][3]
    table = (torch.rand(rows, 2, generator=gen) * 2 - 1) * table_scale
    return {"table": table, "scalings": torch.zeros(num_levels), "log2T": log2T, "tcnn_levels": levels}


def make_scene_tensors(seed: int = 0, kind: str = "active", log2T: int = 19, prop_log2T: int = 17,
                       max_res: int = 2048, table_scale: float = 0.5, density_gain: float = 16.0,
                       density_bias: float = -2.0, color_gain: float = 4.0, beta_gain: float = 12.0,
                       grid: str = "torch", sharp: bool = False, overflow_units: Tuple[int, ...] = (),
                       head_overflow_units: Tuple[int, ...] = (), color_contrast: float = 1.0, hidden_dim: int = 64,
                       hidden_dim_color: int = 64, geo_feat_dim: int = 15, features_per_level: int = 2,
                       appearance_dim: int = 32, prop_linear: bool = False) -> Dict:
    """Random-init nerfacto-shaped scene.  Tables U(-1,1)*table_scale; Linear layers
    Kaiming-uniform like nn.Linear; the density row is gained up so accumulation, depth and the
    variances vary over the image instead of saturating.

    sharp: magnitudes of a TRAINED field instead of a fresh one -- density logits spanning about +-12 (opaque surfaces
    next to empty space: densities from e^-12 to e^12), colour-head activations of the order 1e3 (first hidden layer
    scaled up, the next layer's weights scaled down to match), proposal logits likewise.
    color_contrast: the last colour layer scaled by this factor.  At 1 the colour logits stay within about +-0.3 -- a
    grey image (rgb 0.47 .. 0.59) whose MC-dropout / Laplace rgb_std all lies within a factor of four (0.007 .. 0.03 at
    K = 8); at 10 the image spans 0.02 .. 1.0 with saturated and unsaturated regions and rgb_std 0.04 .. 0.34, the dynamic
    range of a trained model's output (and a ranking by variance that a 1e-5 perturbation does not reshuffle:
    tests/tools/ause_conditioning.py).
    overflow_units: trunk hidden units whose pre-activations reach past 65504 -- beyond the f16 operand range of the
    f16 matrix kernels -- on ~9 % of the samples (first-layer rows of +-5e4, inside the weight limit the packer checks);
    their outgoing weights are 1e-6-small, so in fp32 arithmetic they move the outputs by less than 1.
    head_overflow_units: the same for units of the COLOUR head's first hidden layer (rows of +-5e4 around a bias of 6e4
    -- the bias is not an f16 operand --, i.e. pre-activations of 6e4 +- 2e4: past 65504 on about a third of the samples;
    outgoing weights of 1e-6 into the second hidden layer): the overflow then happens behind a ReLU and two layers
    away from any output."""
    assert kind in ("active", "mcdropout", "laplace") and grid in ("torch", "tcnn")
    gen = torch.Generator().manual_seed(seed)
    make_grid = _grid_tcnn if grid == "tcnn" else _grid
    # hidden_dim / hidden_dim_color / geo_feat_dim / features_per_level / appearance_dim: the widths the reference's model
    # configs forward to the field; anything but 64 / 64 / 15 / 2 runs the any-width kernel.  prop_linear: use_linear=True
    # proposal networks (one Linear on the grid features)
    f = make_grid(gen, 16, 16, max_res, log2T, table_scale) if features_per_level == 2 else \
        _grid(gen, 16, 16, max_res, log2T, table_scale, features_per_level)
    assert features_per_level == 2 or grid == "torch"
    f["sh_remap"] = grid == "tcnn"   # tcnn's SphericalHarmonics encoding maps (d+1)/2 back to [-1,1]
    f["w0"], f["b0"] = _linear(gen, hidden_dim, 16 * features_per_level)
    out1 = geo_feat_dim + {"active": 2, "mcdropout": 1, "laplace": 0}[kind]
    f["w1"], f["b1"] = _linear(gen, out1, hidden_dim)
    head_w, head_b = [], []
    for i, o in ((16 + geo_feat_dim + appearance_dim, hidden_dim_color), (hidden_dim_color, hidden_dim_color), (hidden_dim_color, 3)):
        w, b = _linear(gen, o, i)
        head_w.append(w * (color_gain if o == 3 else 1.0))
        head_b.append(b)
    f["head_w"], f["head_b"] = head_w, head_b
    f["appearance"] = torch.randn(appearance_dim, generator=gen) * 0.1
    f["average_init_density"] = 1.0
    f["beta_min"] = 0.01
    if kind == "laplace":
        f["w1"] = f["w1"] * color_gain
        dw, db = _linear(gen, 1, hidden_dim)
        f["density_w"] = dw * (density_gain / 4)
        f["density_b"] = db * 0 + density_bias
    else:
        f["w1"][0] *= density_gain
        f["w1"][1:1 + geo_feat_dim] *= color_gain
        f["b1"][0] = density_bias
        if kind == "active":
            f["w1"][1 + geo_feat_dim] *= beta_gain  # learned-variance logit: give the variance image dynamic range
            f["b1"][1 + geo_feat_dim] = -2.0
    props = []
    for mr in (128, 256):
        p = make_grid(gen, 5, 16, mr, prop_log2T, table_scale)
        if prop_linear:
            p["w0"] = p["b0"] = None
            p["w1"], p["b1"] = _linear(gen, 1, 10)
        else:
            p["w0"], p["b0"] = _linear(gen, 16, 10)
            p["w1"], p["b1"] = _linear(gen, 1, 16)
        p["w1"][0] *= density_gain
        p["b1"][0] = density_bias + math.log(100.0)  # proposal nets carry average_init_density = 0.01
        props.append(p)
    if sharp:
        if kind == "laplace":
            f["density_w"] = f["density_w"] * 8.0
            f["density_b"] = f["density_b"] * 0.0
        else:
            f["w1"][0] *= 8.0
            f["b1"][0] = 0.0
        f["head_w"][0] = f["head_w"][0] * 4e3
        f["head_b"][0] = f["head_b"][0] * 4e3
        f["head_w"][1] = f["head_w"][1] * 2.5e-4
        for p in props:
            p["w1"][0] *= 3.0
    if color_contrast != 1.0:
        f["head_w"][2] = f["head_w"][2] * color_contrast
    for u in overflow_units:
        f["w0"][u] = torch.sign(f["w0"][u]) * 5e4
        f["b0"][u] = 0.0
        f["w1"][:, u] = torch.sign(f["w1"][:, u]) * 1e-6
        if kind == "laplace":
            f["density_w"][:, u] = torch.sign(f["density_w"][:, u]) * 1e-6
    for u in head_overflow_units:
        f["head_w"][0][u] = torch.sign(f["head_w"][0][u]) * 5e4
        f["head_b"][0][u] = 6e4
        f["head_w"][1][:, u] = torch.sign(f["head_w"][1][:, u]) * 1e-6
    return {"kind": kind, "field": f, "props": props, "near": 0.05, "far": 1000.0, "num_prop": (256, 96),
            "num_nerf": 48, "prop_average_init_density": 0.01}


_MODE = {"active": _l.FIELD_ACTIVE, "mcdropout": _l.FIELD_MCDROPOUT, "laplace": _l.FIELD_LAPLACE}


def scene_to_device(t: Dict, device, **field_kw) -> NerfSceneDev:
    """t["grid_precision"] = "f16" (tcnn-layout scenes only): half tables + tcnn's half interpolation in the main field and
    the proposal networks -- the oracle reads the same key (scene_from_tensors)"""
    f = t["field"]
    gp = t.get("grid_precision", "f32")
    fd = ops.FieldDev.from_torch(_MODE[t["kind"]], f["table"], f["scalings"], f["log2T"], f["w0"], f["b0"], f["w1"],
                                 f["b1"], f["head_w"], f["head_b"], f["appearance"], device,
                                 average_init_density=float(f["average_init_density"]),
                                 beta_min=float(f["beta_min"]), tcnn_levels=f.get("tcnn_levels"),
                                 sh_remap=int(bool(f.get("sh_remap", False))),
                                 **({"grid_precision": gp} if gp != "f32" else {}), **field_kw)
    props = [ops.DensityNetDev.from_torch(p["table"], p["scalings"], p["log2T"], p["w0"], p["b0"], p["w1"], p["b1"],
                                          device, tcnn_levels=p.get("tcnn_levels"), grid_precision=gp) for p in t["props"]]
    if t.get("aabb") is not None:   # disable_scene_contraction: scene-box normalisation in every network
        box = tuple(float(v) for v in t["aabb"].reshape(-1))
        fd.aabb = box
        for p in props:
            p.aabb = box
    return NerfSceneDev(field=fd, props=props, near=float(t["near"]), far=float(t["far"]),
                        num_prop=tuple(t["num_prop"]), num_nerf=int(t["num_nerf"]),
                        prop_average_init_density=float(t["prop_average_init_density"]),
                        spacing=(_l.SPACING_UNIFORM if t.get("proposal_initial_sampler", "piecewise") == "uniform"
                                 else _l.SPACING_PIECEWISE),
                        background=ops.background_of(t.get("background_color", "last_sample")))


def laplace_weight_samples(t: Dict, seed: int = 42, n_samples: int = 100, prior_prec: float = 1.0,
                           eps: float = 1e-9, ggn_scale: float = 1e3):
    """Synthetic diagonal GGN ~ U(0, ggn_scale) and the n_samples last-layer parameter draws
    mu + randn * 1/sqrt(ggn + prior + eps) for both heads (laplace_field.py:538-547).
    -> (ws_density [n,65], ws_rgb [n,195])  (CPU tensors)"""
    g = torch.Generator().manual_seed(seed)
    f = t["field"]
    out = []
    for w, b in ((f["density_w"], f["density_b"]), (f["head_w"][2], f["head_b"][2])):
        mu = torch.cat([w.reshape(-1), b.reshape(-1)])
        ggn = torch.rand(mu.numel(), generator=g) * ggn_scale
        std = 1 / torch.sqrt(ggn + prior_prec + eps)
        out.append(mu.view(1, -1) + torch.randn(n_samples, mu.numel(), generator=g) * std.view(1, -1))
    return out[0], out[1]


def orbit_c2w(theta: float, radius: float = 0.6, height: float = 0.15) -> torch.Tensor:
    """3x4 camera-to-world on a circle, looking at the origin, up = +z (nerfstudio: camera looks down -z)."""
    eye = np.array([radius * math.cos(theta), radius * math.sin(theta), height], dtype=np.float64)
    fwd = -eye / np.linalg.norm(eye)
    up = np.array([0.0, 0.0, 1.0])
    right = np.cross(fwd, up)
    right /= np.linalg.norm(right)
    up2 = np.cross(right, fwd)
    return torch.from_numpy(np.stack([right, up2, -fwd, eye], axis=1).astype(np.float32))


# cameras of SURVEY.md 8(d)
CAMERA_1080P = dict(fx=1111.0, fy=1111.0, cx=960.0, cy=540.0, H=1080, W=1920)
CAMERA_LEGO200 = dict(fx=277.78, fy=277.78, cx=100.0, cy=100.0, H=200, W=200)


def make_splat_tensors(seed: int = 7, N: int = 1_000_000) -> Dict[str, torch.Tensor]:
    """Synthetic splat set of SURVEY.md 8(d) (gauss_params names of activesplatfacto_model.py:72)."""
    g = torch.Generator().manual_seed(seed)
    q = torch.randn(N, 4, generator=g)
    return {
        "means": torch.rand(N, 3, generator=g) * 2 - 1,
        "scales": torch.randn(N, 3, generator=g) * 0.5 - 4.0,
        "quats": q / q.norm(dim=-1, keepdim=True),
        "opacities": torch.randn(N, 1, generator=g) * 2.0,
        "features_dc": torch.randn(N, 3, generator=g) * 0.5,
        "features_rest": torch.randn(N, 15, 3, generator=g) * 0.05,
        "log_uncertainties": torch.rand(N, 1, generator=g),
    }
