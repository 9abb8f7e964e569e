"""Host-side Field mirrors: same class names, constructor arguments and state-dict key names as the
reference's fields, holding the weights as ordinary torch parameters (so reference checkpoints
load with `load_state_dict`) and lowering themselves to the device parameter packs the HIP
kernels consume (`to_device`).

These classes do not compute anything in Python: `get_density` / `get_outputs` of the reference
are replaced by the fused `unerf_field_fwd` kernel, reached through `models.py` / `render.py`
(whole-frame path) or through the Field-level calls nerfstudio makes on a RaySamples --
`field.forward(ray_samples)` / `field(ray_samples)` and `density_field.density_fn(positions)` /
`get_density(ray_samples)` -- which run the same kernels on the caller's own samples.
nerfstudio is not required; when it is installed, `plugin.py` wraps the Model mirrors of `models.py` (which own
these fields under the reference's attribute names) in `nerfstudio.models.base_model.Model` subclasses and registers
them through the `nerfstudio.method_configs` entry points of `pyproject.toml`.

Reference:
  ActiveNerfactoField      models/activenerfacto/activenerfacto_field.py:33-215
  NerfactoMCDropoutField   models/mcdropout/mcdropout_fields.py:22-174
  NerfactoLaplaceField     models/laplace/laplace_field.py:36-608
  HashMLPDensityField      nerfstudio 1.1.0 fields/density_fields.py (proposal networks)
"""
from __future__ import annotations

import enum
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch
from torch import nn

from . import lib as _l
from . import ops
from .synthetic import hash_scalings
from .utils import create_mlp


class FieldHeadNames(enum.Enum):
    """[UPSTREAM nerfstudio.field_components.field_heads.FieldHeadNames] the members this path emits"""
    RGB = "rgb"
    DENSITY = "density"
    UNCERTAINTY = "uncertainty"


@dataclass
class Frustums:
    """[UPSTREAM nerfstudio.cameras.rays.Frustums] the four tensors the field kernels read"""
    origins: torch.Tensor      # [R,S,3]
    directions: torch.Tensor   # [R,S,3]
    starts: torch.Tensor       # [R,S,1] Euclidean
    ends: torch.Tensor         # [R,S,1]


@dataclass
class RaySamples:
    """[UPSTREAM nerfstudio.cameras.rays.RaySamples] stand-in for use without nerfstudio; nerfstudio's own object is
    accepted wherever this one is (only `.frustums.{origins,directions,starts,ends}` is read)."""
    frustums: Frustums
    camera_indices: Optional[torch.Tensor] = None

    @staticmethod
    def from_bins(origins: torch.Tensor, directions: torch.Tensor, euclid_bins: torch.Tensor) -> "RaySamples":
        """rays [R,3] + bin edges [R,S+1] -> the per-sample layout nerfstudio's samplers produce"""
        S = euclid_bins.shape[-1] - 1
        return RaySamples(Frustums(origins[:, None, :].expand(-1, S, -1), directions[:, None, :].expand(-1, S, -1),
                                   euclid_bins[:, :-1, None], euclid_bins[:, 1:, None]))


def ray_samples_to_bins(ray_samples) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """-> origins [R,3], directions [R,3], Euclidean bin edges [R,S+1].  The kernels take contiguous bins (sample i
    ends where sample i+1 starts), which is what every nerfstudio sampler emits."""
    fr = ray_samples.frustums
    starts, ends = fr.starts[..., 0], fr.ends[..., 0]
    if starts.dim() != 2:
        raise _l.UnerfError("ray samples must be [num_rays, num_samples, ...]")
    if starts.shape[1] > 1 and not torch.equal(starts[:, 1:], ends[:, :-1]):
        raise _l.UnerfError("ray samples are not contiguous bins (ends[i] != starts[i+1])")
    f = lambda t: t.to(torch.float32).contiguous()
    return f(fr.origins[:, 0]), f(fr.directions[:, 0]), f(torch.cat([starts, ends[:, -1:]], dim=-1))


class _TcnnParams(nn.Module):
    """what a tinycudann module exposes to the state dict: one flat fp32 `params` vector"""

    def __init__(self, n: int, init_scale: float):
        super().__init__()
        self.params = nn.Parameter((torch.rand(n) * 2 - 1) * init_scale)


def _pad16(n: int) -> int:
    return -(-n // 16) * 16


def unpack_tcnn_mlp(params: torch.Tensor, in_dim: int, width: int, num_layers: int, out_dim: int):
    """tiny-cuda-nn FullyFusedMLP parameter vector -> torch-layout [out,in] weights (tcnn MLPs have no biases).
    Layout [UPSTREAM-RECALL tiny-cuda-nn, SURVEY.md A.6]: first layer [width, pad16(in_dim)], then
    (num_layers - 2) x [width, width], last [pad16(out_dim), width]; row-major, concatenated."""
    shapes = [(width, _pad16(in_dim))] + [(width, width)] * (num_layers - 2) + [(_pad16(out_dim), width)]
    need = sum(r * c for r, c in shapes)
    if params.numel() != need:
        raise ValueError(f"tcnn MLP params: {params.numel()} values, expected {need} for {in_dim}->{width}x"
                         f"{num_layers - 1}->{out_dim}")
    ws, o = [], 0
    for r, c in shapes:
        ws.append(params[o:o + r * c].reshape(r, c))
        o += r * c
    ws[0] = ws[0][:, :in_dim]
    ws[-1] = ws[-1][:out_dim]
    return ws


class HashEncoding(nn.Module):
    """nerfstudio HashEncoding.  implementation="torch": parameter `hash_table` [L*T, F] (every level hashed).
    implementation="tcnn": parameter `tcnn_encoding.params`, the flat fp32 vector of a tiny-cuda-nn HashGrid
    (dense coarse levels, +0.5 cell shift; include/unerf.h: unerf_tcnn_level) -- the layout the reference's
    default configuration trains with (activenerfacto_field.py:89).
    grid_precision (tcnn only): "f16" (default) = the device copy of the parameters is half and the lookup runs in
    tiny-cuda-nn's own half arithmetic (unerf_field_params.grid_half) -- what tcnn computes on every GPU the reference
    targets; "f32" = fp32 rows and blend (a tcnn built without TCNN_HALF_PRECISION).  The parameter itself stays the
    fp32 master vector either way (checkpoints load unchanged)."""

    def __init__(self, num_levels=16, min_res=16, max_res=1024, log2_hashmap_size=19, features_per_level=2,
                 hash_init_scale=0.001, implementation="torch"):
        super().__init__()
        assert implementation in ("torch", "tcnn")
        if features_per_level not in (2, 4) or (features_per_level == 4 and implementation == "tcnn"):
            raise ValueError(f"features_per_level={features_per_level} ({implementation}): 2, or 4 on the torch-layout grid "
                             "(any-width kernel, include/unerf.h)")
        self.features_per_level = features_per_level
        self.num_levels, self.log2_hashmap_size, self.implementation = num_levels, log2_hashmap_size, implementation
        self.register_buffer("scalings", hash_scalings(num_levels, min_res, max_res), persistent=False)
        self.tcnn_levels = None
        self.grid_precision = "f16" if implementation == "tcnn" else "f32"
        if implementation == "tcnn":
            import math
            growth = math.exp((math.log(max_res) - math.log(min_res)) / (num_levels - 1)) if num_levels > 1 else 1.0
            self.tcnn_levels = ops.tcnn_grid_levels(num_levels, min_res, growth, log2_hashmap_size)
            rows = self.tcnn_levels[-1][2] + self.tcnn_levels[-1][3]
            self.tcnn_encoding = _TcnnParams(rows * 2, 1e-4)
        else:
            table = (torch.rand((1 << log2_hashmap_size) * num_levels, features_per_level) * 2 - 1) * hash_init_scale
            self.hash_table = nn.Parameter(table)

    @property
    def table(self) -> torch.Tensor:
        return self.tcnn_encoding.params if self.implementation == "tcnn" else self.hash_table

    def get_out_dim(self) -> int:
        return self.num_levels * self.features_per_level


class MLP(nn.Module):
    """nerfstudio MLP.  implementation="torch": `layers` ModuleList of Linear, ReLU between.
    implementation="tcnn": `tcnn_encoding.params` of a FullyFusedMLP (no biases, padded to multiples of 16)."""

    def __init__(self, in_dim, num_layers, layer_width, out_dim, implementation="torch"):
        super().__init__()
        assert implementation in ("torch", "tcnn")
        self.in_dim, self.num_layers, self.layer_width, self.out_dim = in_dim, num_layers, layer_width, out_dim
        self.implementation = implementation
        dims = [in_dim] + [layer_width] * (num_layers - 1) + [out_dim]
        if implementation == "tcnn":
            n = layer_width * _pad16(in_dim) + (num_layers - 2) * layer_width ** 2 + _pad16(out_dim) * layer_width
            self.tcnn_encoding = _TcnnParams(n, (6.0 / (2 * layer_width)) ** 0.5)
        else:
            self.layers = nn.ModuleList([nn.Linear(dims[i], dims[i + 1]) for i in range(num_layers)])

    def linear_layers(self):
        """-> [(weight [out,in], bias [out])] of the Linear layers, whichever way they are stored"""
        if self.implementation == "tcnn":
            ws = unpack_tcnn_mlp(self.tcnn_encoding.params, self.in_dim, self.layer_width, self.num_layers, self.out_dim)
            return [(w, torch.zeros(w.shape[0], dtype=w.dtype, device=w.device)) for w in ws]
        return [(l.weight, l.bias) for l in self.layers]


class Embedding(nn.Module):
    """nerfstudio Embedding: `embedding` nn.Embedding; mean(dim) = mean of the weight."""

    def __init__(self, in_dim, out_dim):
        super().__init__()
        self.embedding = nn.Embedding(in_dim, out_dim)

    def mean(self, dim=0):
        return self.embedding.weight.mean(dim)


class MLPWithHashEncoding(nn.Module):
    """[UPSTREAM nerfstudio 1.1.0 field_components.mlp.MLPWithHashEncoding -- the block the reference's authors split
    apart at activenerfacto_field.py:124-157; upstream's NerfactoField.mlp_base and HashMLPDensityField.mlp_base are
    this module].  State-dict keys:
      implementation="torch": `encoder.hash_table`, `mlp.layers.{i}.{weight,bias}`, and the same tensors again under
                              `model.0.*` / `model.1.*` (upstream keeps `model = Sequential(encoder, mlp)`);
      implementation="tcnn":  upstream holds ONE tcnn.NetworkWithInputEncoding whose `params` vector is the
                              FullyFusedMLP weights followed by the HashGrid parameters (`model.params`).  This mirror
                              keeps the two parts as `mlp.tcnn_encoding.params` / `encoder.tcnn_encoding.params`;
                              models.remap_checkpoint_keys splits a fused vector by size when a checkpoint has one."""

    def __init__(self, num_levels=16, min_res=16, max_res=1024, log2_hashmap_size=19, features_per_level=2,
                 num_layers=2, layer_width=64, out_dim=1, implementation="torch"):
        super().__init__()
        self.implementation = implementation
        self.encoder = HashEncoding(num_levels, min_res, max_res, log2_hashmap_size, features_per_level,
                                    implementation=implementation)
        self.mlp = MLP(self.encoder.get_out_dim(), num_layers, layer_width, out_dim, implementation=implementation)
        if implementation == "torch":
            self.model = nn.Sequential(self.encoder, self.mlp)

    def fused_tcnn_sizes(self) -> Tuple[int, int]:
        """(FullyFusedMLP values, HashGrid values) of a fused NetworkWithInputEncoding vector"""
        assert self.implementation == "tcnn"
        return self.mlp.tcnn_encoding.params.numel(), self.encoder.tcnn_encoding.params.numel()


class _FieldBuffers:
    """[UPSTREAM nerfstudio 1.1.0 NerfactoField / HashMLPDensityField __init__] the four buffers every field registers
    (and every checkpoint therefore carries): aabb, max_res, num_levels, log2_hashmap_size"""

    def _register_field_buffers(self, aabb, max_res, num_levels, log2_hashmap_size):
        self.register_buffer("aabb", torch.zeros(2, 3) if aabb is None else torch.as_tensor(aabb, dtype=torch.float32))
        self.register_buffer("max_res", torch.tensor(max_res))
        self.register_buffer("num_levels", torch.tensor(num_levels))
        self.register_buffer("log2_hashmap_size", torch.tensor(log2_hashmap_size))


class HashMLPDensityField(nn.Module, _FieldBuffers):
    """Proposal network [UPSTREAM nerfstudio 1.1.0 fields/density_fields.py].  Own keys: `mlp_base.encoder.*`,
    `mlp_base.mlp.layers.{0,1}.*` (+ `mlp_base.model.{0,1}.*`), the MLPWithHashEncoding layout; the older
    `encoding.*` / `mlp_base.{0,1}.*` names (HashEncoding + Sequential) are accepted as aliases on load
    (models.remap_checkpoint_keys)."""

    def __init__(self, aabb=None, num_layers=2, hidden_dim=16, num_levels=5, max_res=128, base_res=16, log2_hashmap_size=17,
                 average_init_density=1.0, implementation="torch", use_linear=False, **_unused):
        super().__init__()
        assert num_layers == 2, "proposal kernels are built for Linear-ReLU-Linear"
        self.use_linear = bool(use_linear)
        self.average_init_density = average_init_density
        self._register_field_buffers(aabb, max_res, num_levels, log2_hashmap_size)
        if use_linear:
            # [UPSTREAM density_fields.py] use_linear=True: `self.encoding = HashEncoding(...)` and
            # `self.linear = nn.Linear(encoding.get_out_dim(), 1)` straight on the grid features, no hidden layer
            # (keys encoding.*, linear.{weight,bias}); unerf_density_net.hidden = 0
            self.encoding = HashEncoding(num_levels, base_res, max_res, log2_hashmap_size, 2, implementation=implementation)
            self.linear = nn.Linear(self.encoding.get_out_dim(), 1)
        else:
            self.mlp_base = MLPWithHashEncoding(num_levels, base_res, max_res, log2_hashmap_size, 2, num_layers, hidden_dim, 1,
                                                implementation=implementation)
            # `encoding`: the grid, under the same attribute name in both forms -- here a plain reference (NOT a second
            # registration: the state dict keeps the one set of mlp_base.* keys)
            object.__setattr__(self, "encoding", self.mlp_base.encoder)

    def to_device(self, device) -> ops.DensityNetDev:
        if self.use_linear:
            w0 = b0 = None
            w1, b1 = self.linear.weight, self.linear.bias
        else:
            (w0, b0), (w1, b1) = self.mlp_base.mlp.linear_layers()
        return ops.DensityNetDev.from_torch(self.encoding.table, self.encoding.scalings, self.encoding.log2_hashmap_size,
                                            w0, b0, w1, b1, device, tcnn_levels=self.encoding.tcnn_levels,
                                            grid_precision=self.encoding.grid_precision)

    # -- Field-level calls (nerfstudio HashMLPDensityField.get_density / Field.density_fn) on the proposal kernel --
    _dev: Optional[ops.DensityNetDev] = None

    def invalidate(self):
        """drop the device copy of the parameters (call after changing them)"""
        self._dev = None

    def _net(self, device) -> ops.DensityNetDev:
        if self._dev is None or self._dev.table.device != torch.device(device):
            self._dev = self.to_device(device)
        return self._dev

    @torch.no_grad()
    def density_fn(self, positions: torch.Tensor, times=None) -> torch.Tensor:
        """[UPSTREAM Field.density_fn] world positions [..., 3] -> density [..., 1].  The kernel evaluates
        origin + direction * t: a zero direction makes every position its own ray origin."""
        _l.require_gpu()
        shp = positions.shape[:-1]
        o = positions.reshape(-1, 3).to(torch.float32).contiguous()
        sb = torch.tensor([0.25, 0.75], device=o.device)   # any two spacing bins: the direction is 0
        d = ops.proposal_density(o, torch.zeros_like(o), sb, self._net(o.device), 0.05, 1000.0, self.average_init_density)
        return d.view(*shp, 1)

    @torch.no_grad()
    def get_density(self, ray_samples) -> Tuple[torch.Tensor, None]:
        """[UPSTREAM HashMLPDensityField.get_density] -> (density [R,S,1], None)"""
        fr = ray_samples.frustums
        pos = fr.origins + fr.directions * (fr.starts + fr.ends) / 2
        return self.density_fn(pos), None

    def forward(self, ray_samples, compute_normals: bool = False):
        density, _ = self.get_density(ray_samples)
        return {FieldHeadNames.DENSITY: density}


class _NerfactoFieldBase(nn.Module, _FieldBuffers):
    """Pieces shared with nerfstudio NerfactoField: colour head input = SH16 + geo15 + appearance32."""

    def __init__(self, num_images, geo_feat_dim=15, appearance_embedding_dim=32,
                 use_average_appearance_embedding=False, implementation="torch"):
        super().__init__()
        assert implementation in ("torch", "tcnn")
        self.implementation = implementation
        self.geo_feat_dim = geo_feat_dim
        self.appearance_embedding_dim = appearance_embedding_dim
        self.use_average_appearance_embedding = use_average_appearance_embedding
        self.embedding_appearance = Embedding(num_images, appearance_embedding_dim)
        self.average_init_density = 1.0

    def _grid_kw(self, grid: HashEncoding):
        """tcnn layout of the grid + tcnn's SphericalHarmonics convention (it maps the (d+1)/2 input back to [-1,1])"""
        return {"tcnn_levels": grid.tcnn_levels, "sh_remap": 1 if self.implementation == "tcnn" else 0,
                "grid_precision": grid.grid_precision}

    def eval_appearance(self) -> torch.Tensor:
        """constant eval embedding: mean of the table or zeros (laplace_field.py:386-398)"""
        if self.use_average_appearance_embedding:
            return self.embedding_appearance.mean(dim=0).detach()
        return torch.zeros(self.appearance_embedding_dim)

    # -- Field-level call on a RaySamples (eval): one fused unerf_field_fwd launch on the caller's samples --------
    _dev: Optional[ops.FieldDev] = None
    _dev_kw: Optional[dict] = None

    def invalidate(self):
        """drop the device copy of the parameters (call after changing them)"""
        self._dev = None

    def _field_dev(self, device, **kw) -> ops.FieldDev:
        if self._dev is None or self._dev.table.device != torch.device(device) or self._dev_kw != kw:
            self._dev, self._dev_kw = self.to_device(device, **kw), dict(kw)
        return self._dev

    def _run(self, ray_samples, ray_offset: int = 0, **dev_kw):
        _l.require_gpu()
        o, d, eb = ray_samples_to_bins(ray_samples)
        f = self._field_dev(o.device, **dev_kw)
        return f, ops.field_fwd(o, d, eb, f, 0.0, 0.0, ray_offset, euclidean_bins=True)

    @torch.no_grad()
    def forward(self, ray_samples, compute_normals: bool = False) -> Dict:
        """Field.forward at eval (deterministic pass): {DENSITY [R,S,1], RGB [R,S,3]} + the method's extra keys."""
        assert not compute_normals, "normals need autograd; not on the render path"
        f, (density, rgb, aux, aux2) = self._run(ray_samples, **self._forward_kw())
        out = {FieldHeadNames.DENSITY: density[0].unsqueeze(-1), FieldHeadNames.RGB: rgb[0]}
        self._extra_outputs(out, aux, aux2)
        return out

    def _forward_kw(self) -> dict:
        return {}

    def _extra_outputs(self, out: Dict, aux, aux2) -> None:
        pass


class ActiveNerfactoField(_NerfactoFieldBase):
    """17-wide trunk output: density, 15 geo features, learned variance logit (beta)."""

    def __init__(self, aabb=None, num_images=1, num_layers=2, hidden_dim=64, geo_feat_dim=15, num_levels=16,
                 base_res=16, max_res=2048, log2_hashmap_size=19, num_layers_color=3, features_per_level=2,
                 hidden_dim_color=64, appearance_embedding_dim=32, use_average_appearance_embedding=False,
                 spatial_distortion=None, implementation="torch", beta_min=0.01, **_unused):
        super().__init__(num_images, geo_feat_dim, appearance_embedding_dim, use_average_appearance_embedding,
                         implementation)
        assert (num_layers, num_layers_color) == (2, 3), "depths other than nerfacto's (2-layer trunk, 3-layer head) are not built"
        self._register_field_buffers(aabb, max_res, num_levels, log2_hashmap_size)
        self.beta_min = beta_min
        self.mlp_base_grid = HashEncoding(num_levels, base_res, max_res, log2_hashmap_size, features_per_level,
                                          implementation=implementation)
        self.mlp_base_mlp = MLP(self.mlp_base_grid.get_out_dim(), num_layers, hidden_dim, 1 + geo_feat_dim + 1,
                                implementation=implementation)
        self.mlp_base = nn.Sequential(self.mlp_base_grid, self.mlp_base_mlp)  # alias keys mlp_base.{0,1}.*
        self.mlp_head = MLP(16 + geo_feat_dim + appearance_embedding_dim, num_layers_color, hidden_dim_color, 3,
                            implementation=implementation)
        self.average_init_density = 1.0  # activenerfacto_field.py:159

    def to_device(self, device, **kw) -> ops.FieldDev:
        (w0, b0), (w1, b1) = self.mlp_base_mlp.linear_layers()
        h = self.mlp_head.linear_layers()
        g = self.mlp_base_grid
        return ops.FieldDev.from_torch(
            _l.FIELD_ACTIVE, g.table, g.scalings, g.log2_hashmap_size, w0, b0, w1, b1,
            [w for w, _ in h], [b for _, b in h], self.eval_appearance(), device,
            average_init_density=self.average_init_density, beta_min=self.beta_min, **self._grid_kw(g), **kw)

    def _extra_outputs(self, out, aux, aux2):
        out["rgb_var"] = aux.unsqueeze(-1)   # beta under the key "rgb_var" (activenerfacto_field.py:209)


class NerfactoMCDropoutField(_NerfactoFieldBase):
    """create_mlp trunk (Linear,ReLU,Dropout,Linear -> keys 0,3) and head (keys 0,2,5)."""

    def __init__(self, aabb=None, num_images=1, num_layers=2, hidden_dim=64, geo_feat_dim=15, num_levels=16,
                 base_res=16, max_res=2048, log2_hashmap_size=19, num_layers_color=3, features_per_level=2,
                 hidden_dim_color=64, appearance_embedding_dim=32, use_average_appearance_embedding=False,
                 spatial_distortion=None, implementation="torch", dropout_rate=0.2,
                 rgb_dropout_layers: Optional[List[int]] = None, density_dropout_layers=True, **_unused):
        super().__init__(num_images, geo_feat_dim, appearance_embedding_dim, use_average_appearance_embedding,
                         implementation)
        assert (num_layers, num_layers_color) == (2, 3), "depths other than nerfacto's (2-layer trunk, 3-layer head) are not built"
        rgb_dropout_layers = [-1] if rgb_dropout_layers is None else list(rgb_dropout_layers)
        # create_mlp (utils.py:6-43) puts a Dropout in front of Linear i for every i in dropout_layers; -1 and
        # num_layers - 1 both mean "in front of the last Linear".  The trunk's hidden layer (density_dropout_layers) and
        # the colour head's two hidden layers run in the matrix kernels; index 0 drops the head's INPUTS (direction
        # encoding, geo features, appearance embedding) and is served by the VALU kernel (DROP_HEADIN: correct, slow).
        bad = [i for i in rgb_dropout_layers if i not in (-1, 0, 1, 2)]
        if bad:
            raise ValueError(f"rgb_dropout_layers={rgb_dropout_layers}: the colour head has Linear layers 0, 1, 2 (-1 = 2)")
        self.density_dropout_layers = bool(density_dropout_layers)
        self.rgb_dropout_layers = rgb_dropout_layers
        self.drop_sites = ((_l.DROP_TRUNK if density_dropout_layers else 0) | (_l.DROP_HEAD0 if 1 in rgb_dropout_layers else 0)
                           | (_l.DROP_HEAD1 if (-1 in rgb_dropout_layers or 2 in rgb_dropout_layers) else 0)
                           | (_l.DROP_HEADIN if 0 in rgb_dropout_layers else 0))
        self.dropout_rate = dropout_rate
        self._register_field_buffers(aabb, max_res, num_levels, log2_hashmap_size)
        if density_dropout_layers:
            self.mlp_base_grid = HashEncoding(num_levels, base_res, max_res, log2_hashmap_size, features_per_level,
                                              implementation=implementation)
            self.mlp_base = create_mlp(self.mlp_base_grid.get_out_dim(), num_layers, hidden_dim, 1 + geo_feat_dim,
                                       activation=nn.ReLU, dropout_layers=[-1], dropout_rate=dropout_rate)
        else:
            # density_dropout_layers=False leaves the PARENT's trunk in place (mcdropout_fields.py:112, :162-166):
            # upstream NerfactoField.mlp_base = MLPWithHashEncoding, keys mlp_base.encoder.* / mlp_base.mlp.layers.*
            self.mlp_base = MLPWithHashEncoding(num_levels, base_res, max_res, log2_hashmap_size, features_per_level,
                                                num_layers, hidden_dim, 1 + geo_feat_dim, implementation=implementation)
        self.mlp_head = create_mlp(16 + geo_feat_dim + appearance_embedding_dim, num_layers_color, hidden_dim_color, 3,
                                   activation=nn.ReLU, out_activation=nn.Sigmoid, dropout_layers=rgb_dropout_layers,
                                   dropout_rate=dropout_rate)

    def to_device(self, device, mc_samples=10, seed=0, **kw) -> ops.FieldDev:
        if self.density_dropout_layers:
            grid = self.mlp_base_grid
            (w0, b0), (w1, b1) = [(m.weight, m.bias) for m in self.mlp_base if isinstance(m, nn.Linear)]
        else:
            grid = self.mlp_base.encoder
            (w0, b0), (w1, b1) = self.mlp_base.mlp.linear_layers()
        hl = [m for m in self.mlp_head if isinstance(m, nn.Linear)]
        # no Dropout module anywhere: the K passes are identical -- p = 0 switches the mask generation off
        p_drop = self.dropout_rate if self.drop_sites else 0.0
        return ops.FieldDev.from_torch(
            _l.FIELD_MCDROPOUT, grid.table, grid.scalings, grid.log2_hashmap_size, w0, b0, w1, b1,
            [m.weight for m in hl], [m.bias for m in hl], self.eval_appearance(), device,
            average_init_density=self.average_init_density, K=mc_samples, seed=seed, p_drop=p_drop,
            drop_sites=self.drop_sites, **self._grid_kw(grid), **kw)

    def _forward_kw(self):
        return {"mc_samples": 0}   # eval-mode Dropout is the identity; the K stochastic passes are the Model's job

    @torch.no_grad()
    def forward_passes(self, ray_samples, mc_samples: int, seed: int = 0, ray_offset: int = 0) -> Dict:
        """The K dropout passes of mcdropout_models.py:116-119 on one RaySamples, fused (grid lookup and the first
        layer shared): {DENSITY [K,R,S,1], RGB [K,R,S,3]}; masks keyed by (seed, pass, ray_offset*S + sample)."""
        f, (density, rgb, _, _) = self._run(ray_samples, ray_offset, mc_samples=mc_samples, seed=seed)
        return {FieldHeadNames.DENSITY: density.unsqueeze(-1), FieldHeadNames.RGB: rgb}


class NerfactoField(_NerfactoFieldBase):
    """[UPSTREAM nerfstudio 1.1.0 fields/nerfacto_field.py] the plain nerfacto field -- what an ensemble member of
    `nerfacto` runs (README.md:106-108, ensemble_utils.py:149-156) and the parent of the three fields above.
    Keys: mlp_base.encoder.hash_table, mlp_base.mlp.layers.{0,1}.* (MLPWithHashEncoding; tcnn: one fused
    mlp_base.model.params vector), mlp_head.layers.{0,1,2}.* (tcnn: mlp_head.tcnn_encoding.params),
    embedding_appearance.embedding.weight; buffers aabb, max_res, num_levels, log2_hashmap_size.
    Rendered by the MCDROPOUT kernel mode with the mask generation off (one deterministic pass)."""

    def __init__(self, aabb=None, num_images=1, num_layers=2, hidden_dim=64, geo_feat_dim=15, num_levels=16,
                 base_res=16, max_res=2048, log2_hashmap_size=19, num_layers_color=3, features_per_level=2,
                 hidden_dim_color=64, appearance_embedding_dim=32, use_average_appearance_embedding=False,
                 spatial_distortion=None, implementation="torch", average_init_density=1.0, **_unused):
        super().__init__(num_images, geo_feat_dim, appearance_embedding_dim, use_average_appearance_embedding,
                         implementation)
        assert (num_layers, num_layers_color) == (2, 3), "depths other than nerfacto's (2-layer trunk, 3-layer head) are not built"
        self._register_field_buffers(aabb, max_res, num_levels, log2_hashmap_size)
        self.average_init_density = average_init_density
        self.mlp_base = MLPWithHashEncoding(num_levels, base_res, max_res, log2_hashmap_size, features_per_level,
                                            num_layers, hidden_dim, 1 + geo_feat_dim, implementation=implementation)
        self.mlp_head = MLP(16 + geo_feat_dim + appearance_embedding_dim, num_layers_color, hidden_dim_color, 3,
                            implementation=implementation)

    def to_device(self, device, **kw) -> ops.FieldDev:
        g = self.mlp_base.encoder
        (w0, b0), (w1, b1) = self.mlp_base.mlp.linear_layers()
        h = self.mlp_head.linear_layers()
        kw.pop("mc_samples", None)
        return ops.FieldDev.from_torch(
            _l.FIELD_MCDROPOUT, g.table, g.scalings, g.log2_hashmap_size, w0, b0, w1, b1,
            [w for w, _ in h], [b for _, b in h], self.eval_appearance(), device,
            average_init_density=self.average_init_density, K=0, p_drop=0.0, **self._grid_kw(g), **kw)


class NerfactoLaplaceField(_NerfactoFieldBase):
    """Explicit last layers for the last-layer Laplace approximation.  Keys: base_grid.hash_table,
    base_mlp.0.*, mlp_density.*, mlp_hidden.*, mlp_head.{0,2}.*, mlp_rgb_ll.*; buffers aabb, max_res,
    num_levels, log2_hashmap_size; plain attributes mlp_density_ggn / mlp_rgb_ggn (laplace_field.py:231-238)."""

    def __init__(self, aabb=None, num_images=1, num_layers=2, hidden_dim=64, geo_feat_dim=15, num_levels=16,
                 base_res=16, max_res=2048, log2_hashmap_size=19, num_layers_color=3, features_per_level=2,
                 hidden_dim_color=64, appearance_embedding_dim=32, use_average_appearance_embedding=False,
                 spatial_distortion=None, implementation="torch", density_activation="trunc_exp", **_unused):
        super().__init__(num_images, geo_feat_dim, appearance_embedding_dim, use_average_appearance_embedding,
                         implementation)
        assert (num_layers, num_layers_color) == (2, 3), "depths other than nerfacto's (2-layer trunk, 3-layer head) are not built"
        if density_activation not in ("trunc_exp", "softplus"):        # laplace_model.py:151
            raise ValueError(f"density_activation={density_activation!r}: expected 'trunc_exp' or 'softplus'")
        self.density_activation = density_activation
        self.register_buffer("aabb", torch.zeros(2, 3) if aabb is None else aabb)
        self.register_buffer("max_res", torch.tensor(max_res))
        self.register_buffer("num_levels", torch.tensor(num_levels))
        self.register_buffer("log2_hashmap_size", torch.tensor(log2_hashmap_size))
        self.base_grid = HashEncoding(num_levels, base_res, max_res, log2_hashmap_size, features_per_level,
                                      implementation=implementation)
        # num_layers-1 == 1 -> a bare Linear: activation AND out_activation are dropped (utils.py:22-23)
        self.base_mlp = create_mlp(self.base_grid.get_out_dim(), num_layers - 1, hidden_dim, hidden_dim,
                                   activation=nn.ReLU, out_activation=nn.ReLU)
        self.mlp_density = nn.Linear(hidden_dim, 1)
        self.mlp_hidden = nn.Linear(hidden_dim, geo_feat_dim)
        self.mlp_head = create_mlp(16 + geo_feat_dim + appearance_embedding_dim, num_layers_color - 1, hidden_dim_color,
                                   hidden_dim_color, activation=nn.ReLU, out_activation=nn.ReLU)
        self.mlp_rgb_ll = nn.Linear(hidden_dim_color, 3)
        self.mlp_density_ggn = torch.zeros(hidden_dim + 1)
        self.mlp_rgb_ggn = torch.zeros(hidden_dim_color * 3 + 3)

    def sample_last_layers(self, n_samples=100, prior_prec=1.0, eps=1e-9, generator=None, rgb_prior_prec=1.0,
                           rgb_n_samples=100, rgb_eps=1e-9, deterministic_density: bool = False, n_sets: Optional[int] = None):
        """The draw of `sample_laplace` (laplace_field.py:538-547) for both heads: mu + randn * 1/sqrt(ggn+prior+eps).
        Quirk kept: forward_unc does not forward prior_prec / n_samples / eps to the colour head
        (laplace_field.py:516-520), which therefore always runs with the defaults 1.0 / 100 / 1e-9.
        deterministic_density (use_deterministic_density=True, laplace_field.py:501-506): no density draw is made
        (the generator is consumed by the colour head only, as in the reference); the density rows are copies of
        the mean.
        n_sets: None -> one set, (ws_density [n,65], ws_rgb [n,195]).  An int -> that many INDEPENDENT sets
        ([n_sets,n,65], [n_sets,n,195]), drawn in the order the reference consumes its generator when it renders a frame
        chunk by chunk (laplace_model.py:432-443): chunk 0 density, chunk 0 colour, chunk 1 density, ..."""
        from torch.nn.utils import parameters_to_vector
        heads = []
        for mod, ggn, pp, n, e in ((self.mlp_density, self.mlp_density_ggn, prior_prec, n_samples, eps),
                                   (self.mlp_rgb_ll, self.mlp_rgb_ggn, rgb_prior_prec, rgb_n_samples, rgb_eps)):
            mu = parameters_to_vector(mod.parameters()).detach()
            heads.append((mu, 1 / torch.sqrt(ggn.to(mu) + pp + e), n))
        sets = []
        for _ in range(1 if n_sets is None else int(n_sets)):
            out = []
            for i, (mu, std, n) in enumerate(heads):
                if i == 0 and deterministic_density:
                    out.append(mu.view(1, -1).repeat(rgb_n_samples, 1))
                    continue
                noise = torch.randn(n, mu.numel(), generator=generator, device=mu.device)
                out.append(mu.view(1, -1) + noise * std.view(1, -1))
            sets.append(out)
        if n_sets is None:
            return sets[0][0], sets[0][1]
        return torch.stack([s_[0] for s_ in sets]), torch.stack([s_[1] for s_ in sets])

    @torch.no_grad()
    def forward_unc(self, ray_samples, compute_normals: bool = False, is_inference: bool = False,
                    use_deterministic_density: bool = False, prior_prec: float = 1.0, n_samples: int = 100,
                    eps: float = 1e-9, generator=None) -> Dict:
        """laplace_field.py:487-525 with is_inference=True on one RaySamples: {DENSITY mu_d [R,S,1], "density_var"
        [R,S,1] | None, RGB mu_rgb [R,S,3], "rgb_var" [R,S,1]}; the last-layer samples are drawn as sample_laplace
        does (sample_last_layers), the 2 x n_samples head evaluations run inside the kernel."""
        if not is_inference:
            raise NotImplementedError("is_inference=False is the training forward")
        assert not compute_normals
        ws_d, ws_r = self.sample_last_layers(n_samples=n_samples, prior_prec=prior_prec, eps=eps, generator=generator,
                                             deterministic_density=use_deterministic_density)
        self.invalidate()   # fresh weight samples every call, like the reference
        _l.require_gpu()
        o, d, eb = ray_samples_to_bins(ray_samples)
        f = self.to_device(o.device, ws_density=ws_d, ws_rgb=ws_r, lap_mask_density=int(use_deterministic_density))
        density, rgb, dvar, rvar = ops.field_fwd(o, d, eb, f, 0.0, 0.0, 0, euclidean_bins=True)
        return {FieldHeadNames.DENSITY: density[0].unsqueeze(-1),
                "density_var": None if use_deterministic_density else dvar.unsqueeze(-1),
                FieldHeadNames.RGB: rgb[0], "rgb_var": rvar.unsqueeze(-1)}

    def forward(self, ray_samples, compute_normals: bool = False) -> Dict:
        """the deterministic field (is_inference=False branch at eval): mean heads, selector-masked density"""
        mu_d = torch.nn.utils.parameters_to_vector(self.mlp_density.parameters()).detach().view(1, -1)
        mu_r = torch.nn.utils.parameters_to_vector(self.mlp_rgb_ll.parameters()).detach().view(1, -1)
        _l.require_gpu()
        o, d, eb = ray_samples_to_bins(ray_samples)
        f = self.to_device(o.device, ws_density=mu_d, ws_rgb=mu_r, lap_mask_density=1)
        density, rgb, _, _ = ops.field_fwd(o, d, eb, f, 0.0, 0.0, 0, euclidean_bins=True)
        return {FieldHeadNames.DENSITY: density[0].unsqueeze(-1), FieldHeadNames.RGB: rgb[0]}

    def to_device(self, device, ws_density=None, ws_rgb=None, **kw) -> ops.FieldDev:
        h = self.mlp_head
        f = lambda t: None if t is None else t.detach().to(device=device, dtype=torch.float32).contiguous()
        return ops.FieldDev.from_torch(
            _l.FIELD_LAPLACE, self.base_grid.table, self.base_grid.scalings, self.base_grid.log2_hashmap_size,
            self.base_mlp[0].weight, self.base_mlp[0].bias, self.mlp_hidden.weight, self.mlp_hidden.bias,
            [h[0].weight, h[2].weight, self.mlp_rgb_ll.weight], [h[0].bias, h[2].bias, self.mlp_rgb_ll.bias],
            self.eval_appearance(), device, ws_density=f(ws_density), ws_rgb=f(ws_rgb),
            lap_softplus=int(self.density_activation == "softplus"), **self._grid_kw(self.base_grid), **kw)
