"""CPU (torch fp32) restatement of the NeRF half of the hot path.

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.

Two kinds of functions live here, tagged in their docstrings:

  [REF file:line]         restates code that is under /root/reference
  [UPSTREAM nerfstudio]   restates nerfstudio==1.1.0 (README.md:23 of the
                          reference pins it); not vendored, not installed ->
                          "parity unpinned" for these.

Everything is written as plain functions over tensors so that each HIP kernel
has a same-shaped counterpart.  fp32 by default; `autocast=torch.float16 / bfloat16`
on the Linear layers (`_linear`) emulates the autocast the reference forces at
eval for MC-dropout (mcdropout_models.py:86-92) -- the arithmetic the kernels'
`precision="f16"` form follows ("Precision").
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------
# L0.1  HashEncoding (torch path)                         [UPSTREAM nerfstudio]
# --------------------------------------------------------------------------

HASH_PRIMES = (1, 2654435761, 805459861)


def hash_scalings(num_levels: int, min_res: int, max_res: int) -> torch.Tensor:
    """scalings[l] = floor(min_res * growth**l), evaluated the way upstream does:
    numpy float64 growth factor raised to an int64 *torch* tensor -> float32 pow.
    (This is why the last level of a 16..2048 grid comes out as 2047.)"""
    levels = torch.arange(num_levels)
    growth = np.exp((np.log(max_res) - np.log(min_res)) / (num_levels - 1)) if num_levels > 1 else 1
    return torch.floor(min_res * growth ** levels).to(torch.float32)


def hash_fn(coords: torch.Tensor, log2_T: int, num_levels: int) -> torch.Tensor:
    """coords [..., L, 3] int32 -> table row index [..., L] (int64), level offset included."""
    c = coords.to(torch.int64) * torch.tensor(HASH_PRIMES, dtype=torch.int64)
    x = torch.bitwise_xor(c[..., 0], c[..., 1])
    x = torch.bitwise_xor(x, c[..., 2])
    x = x % (1 << log2_T)
    x = x + torch.arange(num_levels, dtype=torch.int64) * (1 << log2_T)
    return x


# corner order of upstream pytorch_fwd: (x,y,z) pick ceil 'c' or floor 'f'
_CORNERS = ("ccc", "cfc", "ffc", "fcc", "ccf", "cff", "fff", "fcf")


def hash_indices(x: torch.Tensor, scalings: torch.Tensor, log2_T: int):
    """x [N,3] in [0,1] -> (idx [N,L,8] int64, offset [N,L,3] fp32)."""
    L = scalings.numel()
    scaled = x[..., None, :] * scalings.view(-1, 1)
    sc = torch.ceil(scaled).to(torch.int32)
    sf = torch.floor(scaled).to(torch.int32)
    offset = scaled - sf
    idx = []
    for pat in _CORNERS:
        comps = [(sc if pat[a] == "c" else sf)[..., a:a + 1] for a in range(3)]
        idx.append(hash_fn(torch.cat(comps, dim=-1), log2_T, L))
    return torch.stack(idx, dim=-1), offset


def hash_encode(x: torch.Tensor, table: torch.Tensor, scalings: torch.Tensor, log2_T: int) -> torch.Tensor:
    """x [N,3] -> [N, L*F]; blend order f03,f12,f56,f47 -> y -> z as upstream."""
    idx, o = hash_indices(x, scalings, log2_T)
    f = [table[idx[..., k]] for k in range(8)]  # each [N,L,F]
    ox, oy, oz = o[..., 0:1], o[..., 1:2], o[..., 2:3]
    f03 = f[0] * ox + f[3] * (1 - ox)
    f12 = f[1] * ox + f[2] * (1 - ox)
    f56 = f[5] * ox + f[6] * (1 - ox)
    f47 = f[4] * ox + f[7] * (1 - ox)
    f0312 = f03 * oy + f12 * (1 - oy)
    f4756 = f47 * oy + f56 * (1 - oy)
    enc = f0312 * oz + f4756 * (1 - oz)
    return torch.flatten(enc, start_dim=-2)


# --------------------------------------------------------------------------
# L0.1b tiny-cuda-nn HashGrid, as HashEncoding(implementation="tcnn") configures it
#                                         [UPSTREAM-RECALL tiny-cuda-nn, SURVEY.md A.6]
# --------------------------------------------------------------------------

def tcnn_grid_levels(num_levels: int, base_res: int, per_level_scale: float, log2_hashmap_size: int):
    """scale_l = exp2f(l * log2f(per_level_scale)) * base_res - 1 (float32); res_l = ceil(scale_l) + 1;
    rows_l = min(next_multiple(res_l^3, 8), 2^log2_hashmap_size); levels concatenated.
    -> [(scale, res, offset, size, dense)] with offset / size in rows of 2 features."""
    log2_pls = np.float32(np.log2(np.float32(per_level_scale)))
    out, offset = [], 0
    for l in range(num_levels):
        scale = np.float32(np.float32(np.exp2(np.float32(np.float32(l) * log2_pls))) * np.float32(base_res) - np.float32(1))
        res = int(np.ceil(scale)) + 1
        size = min(-(-res ** 3 // 8) * 8, 1 << log2_hashmap_size)
        out.append((float(scale), res, offset, size, int(res ** 3 <= size)))
        offset += size
    return out


def tcnn_hash_indices(x: torch.Tensor, levels):
    """x [N,3] in [0,1] -> (rows [N,L,8] int64 absolute row per corner, w [N,L,3] interpolation weights).
    pos = fma(scale, x, 0.5); cell = floor(pos); corner k steps +1 along dim d where bit d of k is set;
    dense levels: (x + y res + z res^2) mod size; hashed: (x ^ y*2654435761 ^ z*805459861) mod size (uint32)."""
    rows, ws = [], []
    x64 = x.double()
    for scale, res, offset, size, dense in levels:
        pos = (x64 * float(np.float32(scale)) + 0.5).float()   # fp32 product is exact in fp64: one rounding = fmaf
        cell = torch.floor(pos)
        ws.append(pos - cell)
        c = cell.long()
        per = []
        for k in range(8):
            cx, cy, cz = c[:, 0] + (k & 1), c[:, 1] + ((k >> 1) & 1), c[:, 2] + ((k >> 2) & 1)
            if dense:
                idx = (cx + cy * res + cz * res * res) % size
            else:
                M = 0xFFFFFFFF
                idx = ((cx & M) ^ ((cy * 2654435761) & M) ^ ((cz * 805459861) & M)) % size
            per.append(idx + offset)
        rows.append(torch.stack(per, dim=-1))
    return torch.stack(rows, dim=1), torch.stack(ws, dim=1)


def tcnn_hash_encode(x: torch.Tensor, params: torch.Tensor, levels) -> torch.Tensor:
    """-> [N, 2L], level-major.  Corner weight = prod_d (bit ? w_d : 1 - w_d); the 8 corners are accumulated in
    corner order with a fused multiply-add (float64 product + sum rounded once to float32).  fp32 values; tcnn
    itself interpolates fp16 copies of these fp32 master parameters (documented divergence)."""
    rows, w = tcnn_hash_indices(x, levels)
    tab = params.reshape(-1, 2)
    N, L = x.shape[0], len(levels)
    out = torch.zeros(N, L, 2, dtype=torch.float32)
    for k in range(8):
        wk = torch.ones(N, L, dtype=torch.float32)
        for d in range(3):
            wd = w[..., d]
            wk = wk * (wd if (k >> d) & 1 else (1.0 - wd))
        val = tab[rows[..., k]]
        out = (wk[..., None].double() * val.double() + out.double()).float()
    return out.reshape(N, 2 * L)


def _round_f16_of_sum(p: np.ndarray, c: np.ndarray) -> np.ndarray:
    """round_to_nearest_even_f16(p + c) for float64 arrays whose EXACT sum may need more than 53 bits (p: a 22-bit product of
    two halves, c: a half).  TwoSum gives s = fl64(p + c) and the exact error e; rounding s to f16 directly is wrong only
    when s sits exactly on the midpoint of two neighbouring halves and e != 0, where the exact sum is off the midpoint on
    e's side -- there the neighbour of s on that side is rounded instead."""
    s = p + c
    bb = s - p
    e = (p - (s - bb)) + (c - bb)
    with np.errstate(over="ignore"):
        h = s.astype(np.float16)
        up = np.nextafter(s, np.inf).astype(np.float16)
        dn = np.nextafter(s, -np.inf).astype(np.float16)
    tie = (up != dn) & (e != 0)
    return np.where(tie & (e > 0), up, np.where(tie & (e < 0), dn, h))


def tcnn_hash_encode_half(x: torch.Tensor, params: torch.Tensor, levels) -> torch.Tensor:
    """tiny-cuda-nn's `kernel_grid` (include/tiny-cuda-nn/encodings/grid.h, HEAD -- the reference pins no version,
    README.md:21) in the precision tcnn is built with on every GPU the reference targets (TCNN_HALF_PRECISION, T = __half),
    i.e. what HashEncoding(implementation="tcnn") returns [UPSTREAM-RECALL]:
      * the grid is the HALF copy of the fp32 master parameters (`params` is cast to T before the forward pass);
      * pos_fract: pos = fmaf(scale, x, 0.5f); cell = floorf(pos); w = pos - cell  -- fp32, as tcnn_hash_indices;
      * per corner idx = 0..7: weight = 1.f; for dim in x, y, z: weight *= (idx >> dim & 1) ? w[dim] : 1 - w[dim]  (fp32),
        result = fma((T)weight, grid_val, result)  -- a half-precision fused multiply-add per feature (__hfma2), result
        starting at zero, corners in index order;
      * the encoded position is stored as T.
    -> [N, 2L] float32 holding the half values, level-major.  Every rounding above is reproduced exactly (the hfma through
    _round_f16_of_sum), so a kernel doing the same arithmetic matches bit for bit."""
    rows, w = tcnn_hash_indices(x, levels)
    tab = params.detach().reshape(-1, 2).to(torch.float32).numpy().astype(np.float16).astype(np.float64)
    rows_np, w_np = rows.numpy(), w.numpy().astype(np.float32)
    N, L = x.shape[0], len(levels)
    res = np.zeros((N, L, 2), dtype=np.float64)
    one = np.float32(1.0)
    for k in range(8):
        wk = np.ones((N, L), dtype=np.float32)
        for d in range(3):
            wd = w_np[..., d]
            wk = (wk * (wd if (k >> d) & 1 else (one - wd))).astype(np.float32)
        wh = wk.astype(np.float16).astype(np.float64)
        val = tab[rows_np[..., k]]                                  # [N, L, 2]
        res = _round_f16_of_sum(wh[..., None] * val, res).astype(np.float64)
    return torch.from_numpy(res.astype(np.float32).reshape(N, 2 * L))


def unpack_tcnn_mlp(params: torch.Tensor, in_dim: int, width: int, n_hidden_layers: int, out_dim: int):
    """tcnn FullyFusedMLP parameter vector -> torch-layout [out,in] weight matrices (no biases).
    Layout: first layer [width, pad16(in_dim)], (n_hidden_layers - 1) x [width, width], last [pad16(out_dim), width],
    each row-major, concatenated; padded input columns / output rows are dropped."""
    pad = lambda n: -(-n // 16) * 16
    ws, o = [], 0
    shapes = [(width, pad(in_dim))] + [(width, width)] * (n_hidden_layers - 1) + [(pad(out_dim), width)]
    for r, c in shapes:
        ws.append(params[o:o + r * c].reshape(r, c))
        o += r * c
    assert o == params.numel(), (o, params.numel())
    ws[0] = ws[0][:, :in_dim]
    ws[-1] = ws[-1][:out_dim]
    return [w_.contiguous() for w_ in ws]


# --------------------------------------------------------------------------
# L0.2-L0.5 small pieces                                  [UPSTREAM nerfstudio]
# --------------------------------------------------------------------------

def mlp_forward(x: torch.Tensor, weights: List[torch.Tensor], biases: List[torch.Tensor],
                out_activation: Optional[str] = None, autocast: Optional[torch.dtype] = None) -> torch.Tensor:
    """nerfstudio MLP.pytorch_fwd: ReLU on all but the last layer.  autocast: see _linear."""
    n = len(weights)
    for i, (w, b) in enumerate(zip(weights, biases)):
        x = _linear(x, w, b, autocast)
        if i < n - 1:
            x = F.relu(x)
    if out_activation == "sigmoid":
        x = torch.sigmoid(x)
    elif out_activation == "relu":
        x = F.relu(x)
    return x


SH_C = dict(
    c0=0.28209479177387814, c1=0.4886025119029199, c2a=1.0925484305920792, c2b=0.9461746957575601,
    c2c=0.31539156525251999, c2d=0.5462742152960396, c3a=0.5900435899266435, c3b=2.890611442640554,
    c3c=0.4570457994644658, c3d=0.3731763325901154, c3e=1.445305721320277,
)


def sh16(d: torch.Tensor) -> torch.Tensor:
    """components_from_spherical_harmonics(levels=4).  `d` is used AS GIVEN: the torch
    SHEncoding receives get_normalized_directions(d) = (d+1)/2 (laplace_field.py:379)
    and does not map it back to [-1,1] (tcnn does) -- a torch/tcnn divergence."""
    x, y, z = d[..., 0], d[..., 1], d[..., 2]
    xx, yy, zz = x ** 2, y ** 2, z ** 2
    c = SH_C
    comps = [
        torch.full_like(x, c["c0"]),
        c["c1"] * y, c["c1"] * z, c["c1"] * x,
        c["c2a"] * x * y, c["c2a"] * y * z, c["c2b"] * zz - c["c2c"], c["c2a"] * x * z, c["c2d"] * (xx - yy),
        c["c3a"] * y * (3 * xx - yy), c["c3b"] * x * y * z, c["c3c"] * y * (5 * zz - 1),
        c["c3d"] * z * (5 * zz - 3), c["c3c"] * x * (5 * zz - 1), c["c3e"] * z * (xx - yy),
        c["c3a"] * x * (xx - 3 * yy),
    ]
    return torch.stack(comps, dim=-1)


def contract_inf(x: torch.Tensor) -> torch.Tensor:
    """SceneContraction(order=inf): x if |x|_inf<1 else (2-1/m) * (x/m)."""
    mag = torch.linalg.norm(x, ord=float("inf"), dim=-1)[..., None]
    return torch.where(mag < 1, x, (2 - (1 / mag)) * (x / mag))


def normalized_positions(positions: torch.Tensor, aabb: Optional[torch.Tensor] = None):
    """[REF activenerfacto_field.py:164-172] contraction -> (x+2)/4 -> selector mask; with `aabb` [2,3] the
    spatial_distortion-is-None branch (:168, disable_scene_contraction): [UPSTREAM SceneBox.get_normalized_positions]
    (x - aabb[0]) / (aabb[1] - aabb[0])."""
    if aabb is not None:
        p = (positions - aabb[0]) / (aabb[1] - aabb[0])
    else:
        p = contract_inf(positions)
        p = (p + 2.0) / 4.0
    selector = ((p > 0.0) & (p < 1.0)).all(dim=-1)
    p = p * selector[..., None]
    return p, selector


# --------------------------------------------------------------------------
# L0.9 camera rays                                        [UPSTREAM nerfstudio]
# --------------------------------------------------------------------------

# [UPSTREAM-RECALL] the two constants of camera_utils.radial_and_tangential_undistort's signature (nerfstudio 1.1.0:
# `eps: float = 1e-3, max_iterations: int = 10`); include/unerf.h carries the same pair as UNERF_UNDISTORT_*
UNDISTORT_EPS = 1e-3
UNDISTORT_MAX_ITERATIONS = 10


def radial_and_tangential_undistort(coords: torch.Tensor, distortion_params: torch.Tensor,
                                    eps: float = UNDISTORT_EPS, max_iterations: int = UNDISTORT_MAX_ITERATIONS) -> torch.Tensor:
    """[UPSTREAM-RECALL nerfstudio 1.1.0 cameras/camera_utils.py: radial_and_tangential_undistort +
    _compute_residual_and_jacobian, adapted there from MultiNeRF] -- the WHOLE of upstream's undistortion lives in
    this one function, so a diff against a real install is one function.
    coords [..., 2] distorted image-plane coordinates; distortion_params [6] = (k1, k2, k3, k4, p1, p2).
    Newton's method from the distorted point on the forward OPENCV model
        xd = x d + 2 p1 x y + p2 (r + 2 x^2),  yd = y d + 2 p2 x y + p1 (r + 2 y^2),
        r = x^2 + y^2,  d = 1 + r (k1 + r (k2 + r (k3 + r k4)));
    a fixed `max_iterations` steps, each taken only where |det J| > eps."""
    k1, k2, k3, k4, p1, p2 = (distortion_params[..., i] for i in range(6))
    xd, yd = coords[..., 0], coords[..., 1]
    x, y = xd, yd
    for _ in range(max_iterations):
        r = x * x + y * y
        d = 1.0 + r * (k1 + r * (k2 + r * (k3 + r * k4)))
        fx = d * x + 2 * p1 * x * y + p2 * (r + 2 * x * x) - xd
        fy = d * y + 2 * p2 * x * y + p1 * (r + 2 * y * y) - yd
        d_r = k1 + r * (2.0 * k2 + r * (3.0 * k3 + r * 4.0 * k4))
        d_x = 2.0 * x * d_r
        d_y = 2.0 * y * d_r
        fx_x = d + d_x * x + 2.0 * p1 * y + 6.0 * p2 * x
        fx_y = d_y * x + 2.0 * p1 * x + 2.0 * p2 * y
        fy_x = d_x * y + 2.0 * p2 * y + 2.0 * p1 * x
        fy_y = d + d_y * y + 2.0 * p2 * x + 6.0 * p1 * y
        denominator = fy_x * fx_y - fx_x * fy_y
        x_numerator = fx * fy_y - fy * fx_y
        y_numerator = fy * fx_x - fx * fy_x
        ok = torch.abs(denominator) > eps
        x = x + torch.where(ok, x_numerator / denominator, torch.zeros_like(denominator))
        y = y + torch.where(ok, y_numerator / denominator, torch.zeros_like(denominator))
    return torch.stack([x, y], dim=-1)


CAMERA_PERSPECTIVE, CAMERA_FISHEYE, CAMERA_EQUIRECTANGULAR, CAMERA_ORTHOPHOTO = 1, 2, 3, 8   # nerfstudio CameraType values


def generate_rays(c2w: torch.Tensor, fx: float, fy: float, cx: float, cy: float, H: int, W: int, distortion=None,
                  camera_type: int = CAMERA_PERSPECTIVE):
    """Cameras.generate_rays(keep_shape=True) for ONE camera [UPSTREAM-RECALL nerfstudio 1.1.0
    Cameras._generate_rays_from_coords].  camera_type (CameraType value; the reference's parsers pass
    CAMERA_MODEL_TO_TYPE[meta["camera_model"]] through, dataparsers/sparse_mipnerf360/sparse_mipnerf360_dataparser.py:237-239):
    PERSPECTIVE (u, v, -1); FISHEYE theta = clip(|(u, v)|, 0, pi) -> (u sin(theta) / theta, v sin(theta) / theta, -cos(theta));
    EQUIRECTANGULAR theta = -pi u, phi = pi (0.5 - v) -> (-sin(theta) sin(phi), cos(phi), -cos(theta) sin(phi)), no lens
    undistortion ("do not apply distortion for equirectangular images"); ORTHOPHOTO (0, 0, -1) with the origin moved to
    c2w (u, v, 0, 1).  distortion: the camera's `distortion_params` (k1, k2, k3, k4, p1, p2) as the
    reference's dataparsers pass them (dataparsers/sparse_mipnerf360/sparse_mipnerf360_dataparser.py:113-125, 248-274),
    or None; when any is non-zero the coordinate stack (pixel centre and its +1 x / y neighbours, y already negated) is
    undistorted before the rotation, as upstream does under `mask.any() and (distortion_params != 0).any()`.
    Returns origins [H,W,3], directions [H,W,3], pixel_area [H,W,1]."""
    c2w = c2w.to(torch.float32)
    ii, jj = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    y = ii + 0.5
    x = jj + 0.5
    fx, fy, cx, cy = (torch.tensor(v, dtype=torch.float32) for v in (fx, fy, cx, cy))
    coord = torch.stack([(x - cx) / fx, -(y - cy) / fy], -1)
    coord_x = torch.stack([(x - cx + 1) / fx, -(y - cy) / fy], -1)
    coord_y = torch.stack([(x - cx) / fx, -(y - cy + 1) / fy], -1)
    cs = torch.stack([coord, coord_x, coord_y], dim=0)
    assert camera_type in (CAMERA_PERSPECTIVE, CAMERA_FISHEYE, CAMERA_EQUIRECTANGULAR, CAMERA_ORTHOPHOTO), camera_type
    if distortion is not None and camera_type != CAMERA_EQUIRECTANGULAR:
        dp = torch.as_tensor(distortion, dtype=torch.float32).reshape(6)
        if bool((dp != 0).any()):
            cs = radial_and_tangential_undistort(cs, dp)
    if camera_type == CAMERA_FISHEYE:
        theta = torch.clip(torch.sqrt(torch.sum(cs ** 2, dim=-1)), 0.0, math.pi)
        st = torch.sin(theta)
        ds = torch.stack([cs[..., 0] * st / theta, cs[..., 1] * st / theta, -torch.cos(theta)], dim=-1)
    elif camera_type == CAMERA_EQUIRECTANGULAR:
        theta = -torch.pi * cs[..., 0]
        phi = torch.pi * (0.5 - cs[..., 1])
        ds = torch.stack([-torch.sin(theta) * torch.sin(phi), torch.cos(phi), -torch.cos(theta) * torch.sin(phi)], dim=-1)
    elif camera_type == CAMERA_ORTHOPHOTO:
        ds = torch.cat([torch.zeros_like(cs), -torch.ones_like(cs[..., :1])], dim=-1)
    else:
        ds = torch.cat([cs, -torch.ones_like(cs[..., :1])], dim=-1)  # [3,H,W,3]
    rot = c2w[:3, :3]
    ds = torch.sum(ds[..., None, :] * rot, dim=-1)
    norm = torch.maximum(torch.linalg.vector_norm(ds, dim=-1, keepdim=True), torch.tensor([1e-7]))
    ds = ds / norm
    directions = ds[0]
    dx = torch.sqrt(torch.sum((directions - ds[1]) ** 2, dim=-1))
    dy = torch.sqrt(torch.sum((directions - ds[2]) ** 2, dim=-1))
    pixel_area = (dx * dy)[..., None]
    if camera_type == CAMERA_ORTHOPHOTO:   # grids = (u, v, 0, 1); origins = c2w @ grids
        g = torch.cat([cs[0], torch.zeros_like(cs[0][..., :1]), torch.ones_like(cs[0][..., :1])], dim=-1)
        origins = torch.matmul(c2w[:3, :4], g[..., None])[..., 0].contiguous()
    else:
        origins = c2w[:3, 3].expand(H, W, 3).contiguous()
    return origins, directions, pixel_area


# --------------------------------------------------------------------------
# L0.6 samplers                                           [UPSTREAM nerfstudio]
# --------------------------------------------------------------------------

def spacing_fn(x: torch.Tensor) -> torch.Tensor:
    return torch.where(x < 1, x / 2, 1 - 1 / (2 * x))


def spacing_fn_inv(x: torch.Tensor) -> torch.Tensor:
    return torch.where(x < 0.5, 2 * x, 1 / (2 - 2 * x))


def spacing_to_euclidean(bins: torch.Tensor, near, far, uniform: bool = False) -> torch.Tensor:
    """near / far: python floats (NearFarCollider planes) or per-ray tensors [R,1] (RayBundle.nears / fars, e.g. from an
    obb_box intersection).  uniform: [UPSTREAM nerfstudio 1.1.0 ray_samplers.UniformSampler, selected by
    NerfactoModelConfig.proposal_initial_sampler="uniform" -- the reference's few-view runs, README.md:153] spacing_fn =
    spacing_fn_inv = identity, so SpacedSampler's `spacing_to_euclidean_fn` is x far + (1 - x) near."""
    near, far = torch.as_tensor(near, dtype=torch.float32), torch.as_tensor(far, dtype=torch.float32)
    if uniform:
        return bins * far + (1 - bins) * near
    s_near = spacing_fn(near)
    s_far = spacing_fn(far)
    return spacing_fn_inv(bins * s_far + (1 - bins) * s_near)


def intersect_obb(origins: torch.Tensor, directions: torch.Tensor, R: torch.Tensor, T: torch.Tensor, S: torch.Tensor,
                  max_bound: float = 1e10, invalid_value: float = 1e10):
    """[UPSTREAM-RECALL nerfstudio.utils.math.intersect_obb / intersect_aabb, nerfstudio 1.1.0] ray / oriented-box
    intersection as `Cameras.generate_rays(..., obb_box=box)` uses it to set RayBundle.nears / fars (which the
    NearFarCollider then leaves alone): rays are moved into the box frame with inverse([R|T]), slab test against
    +-S/2, t clamped to [0, max_bound], rays that miss get invalid_value for both.  -> nears [N,1], fars [N,1]"""
    H = torch.eye(4)
    H[:3, :3], H[:3, 3] = R, T
    Hw2b = torch.inverse(H)
    o = torch.cat([origins, torch.ones_like(origins[..., :1])], dim=-1)
    o = torch.matmul(Hw2b, o.T).T[..., :3]
    d = torch.matmul(Hw2b[:3, :3], directions.T).T
    lo, hi = -S / 2, S / 2
    tx_min = (lo - o) / d
    tx_max = (hi - o) / d
    t_min = torch.stack((tx_min, tx_max)).amin(dim=0).amax(dim=-1)
    t_max = torch.stack((tx_min, tx_max)).amax(dim=0).amin(dim=-1)
    t_min = torch.clamp(t_min, min=0, max=max_bound)
    t_max = torch.clamp(t_max, min=0, max=max_bound)
    cond = t_max <= t_min
    t_min = torch.where(cond, torch.full_like(t_min, invalid_value), t_min)
    t_max = torch.where(cond, torch.full_like(t_max, invalid_value), t_max)
    return t_min[..., None], t_max[..., None]


def initial_spacing_bins(num_samples: int) -> torch.Tensor:
    return torch.linspace(0.0, 1.0, num_samples + 1)


def pdf_u(num_samples: int) -> torch.Tensor:
    nb = num_samples + 1
    u = torch.linspace(0.0, 1.0 - (1.0 / nb), steps=nb)
    return u + 1.0 / (2 * nb)


def get_weights(density: torch.Tensor, deltas: torch.Tensor) -> torch.Tensor:
    """RaySamples.get_weights; verbatim twin in the reference at laplace_model.py:47-62.
    density, deltas: [R,S]."""
    dd = deltas * density
    alphas = 1 - torch.exp(-dd)
    trans = torch.cumsum(dd[..., :-1], dim=-1)
    trans = torch.cat([torch.zeros_like(trans[..., :1]), trans], dim=-1)
    trans = torch.exp(-trans)
    return torch.nan_to_num(alphas * trans)


def pdf_resample(weights: torch.Tensor, spacing_bins: torch.Tensor, num_samples: int,
                 histogram_padding: float = 0.01, eps: float = 1e-5) -> torch.Tensor:
    """PDFSampler.generate_ray_samples, eval branch.  weights [R,n], spacing_bins [R,n+1]
    -> new spacing bins [R, num_samples+1]."""
    w = weights + histogram_padding
    wsum = torch.sum(w, dim=-1, keepdim=True)
    padding = torch.relu(eps - wsum)
    w = w + padding / w.shape[-1]
    wsum = wsum + padding
    pdf = w / wsum
    cdf = torch.min(torch.ones_like(pdf), torch.cumsum(pdf, dim=-1))
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], dim=-1)
    nb = num_samples + 1
    u = pdf_u(num_samples).expand(*cdf.shape[:-1], nb).contiguous()
    existing = spacing_bins
    inds = torch.searchsorted(cdf, u, side="right")
    below = torch.clamp(inds - 1, 0, existing.shape[-1] - 1)
    above = torch.clamp(inds, 0, existing.shape[-1] - 1)
    cdf_g0 = torch.gather(cdf, -1, below)
    bins_g0 = torch.gather(existing, -1, below)
    cdf_g1 = torch.gather(cdf, -1, above)
    bins_g1 = torch.gather(existing, -1, above)
    t = torch.clip(torch.nan_to_num((u - cdf_g0) / (cdf_g1 - cdf_g0), 0), 0, 1)
    return bins_g0 + t * (bins_g1 - bins_g0)


@dataclass
class GridMLP:
    """hash grid + small MLP (weights in torch nn.Linear layout [out,in])."""
    table: torch.Tensor
    scalings: torch.Tensor
    log2_T: int
    weights: List[torch.Tensor]
    biases: List[torch.Tensor]
    tcnn_levels: Optional[list] = None   # set: `table` is a tcnn-layout parameter vector (tcnn_grid_levels)
    aabb: Optional[torch.Tensor] = None  # [2,3]: scene-box normalisation instead of the contraction
    grid_half: bool = False              # tcnn layout only: tcnn's own half-precision arithmetic (tcnn_hash_encode_half)


def grid_encode(x: torch.Tensor, g: "GridMLP") -> torch.Tensor:
    if g.tcnn_levels is not None and g.grid_half:
        return tcnn_hash_encode_half(x, g.table, g.tcnn_levels)
    if g.tcnn_levels is not None:
        return tcnn_hash_encode(x, g.table, g.tcnn_levels)
    return hash_encode(x, g.table, g.scalings, g.log2_T)


def sample_positions(origins, directions, euclid_bins):
    """Frustums.get_positions(): o + d * (start+end)/2.  [R,3],[R,3],[R,n+1] -> [R,n,3]"""
    starts, ends = euclid_bins[..., :-1], euclid_bins[..., 1:]
    return origins[:, None, :] + directions[:, None, :] * (starts + ends)[..., None] / 2


def density_field(positions: torch.Tensor, net: GridMLP, average_init_density: float) -> torch.Tensor:
    """HashMLPDensityField.get_density on explicit positions [R,n,3] -> [R,n]."""
    shp = positions.shape[:-1]
    p, sel = normalized_positions(positions, net.aabb)
    h = grid_encode(p.reshape(-1, 3), net)
    out = mlp_forward(h, net.weights, net.biases).view(*shp)
    return average_init_density * torch.exp(out) * sel


def proposal_sample(origins, directions, near, far, prop_nets: List[GridMLP], num_prop: Tuple[int, ...],
                    num_nerf: int, average_init_density: float, uniform: bool = False):
    """ProposalNetworkSampler.generate_ray_samples at eval (anneal = 1, no jitter).
    Returns (final spacing bins [R,num_nerf+1], weights_list, spacing_bins_list)."""
    R = origins.shape[0]
    weights_list, bins_list = [], []
    bins = initial_spacing_bins(num_prop[0])[None].expand(R, -1).contiguous()
    weights = None
    n_iter = len(prop_nets)
    for lvl in range(n_iter + 1):
        if lvl > 0:
            n_new = num_prop[lvl] if lvl < n_iter else num_nerf
            bins = pdf_resample(weights, bins, n_new)
        if lvl < n_iter:
            eb = spacing_to_euclidean(bins, near, far, uniform)
            pos = sample_positions(origins, directions, eb)
            dens = density_field(pos, prop_nets[lvl], average_init_density)
            weights = get_weights(dens, eb[..., 1:] - eb[..., :-1])
            weights_list.append(weights)
            bins_list.append(bins)
    return bins, weights_list, bins_list


# --------------------------------------------------------------------------
# L0.8 renderers                                          [UPSTREAM nerfstudio]
# --------------------------------------------------------------------------

BACKGROUND_COLORS = {"white": (1.0, 1.0, 1.0), "black": (0.0, 0.0, 0.0)}


def render_rgb(rgb: torch.Tensor, weights: torch.Tensor, background="last_sample") -> torch.Tensor:
    """[UPSTREAM nerfstudio 1.1.0 RGBRenderer.forward / combine_rgb] at eval.  rgb [R,S,3], weights [R,S].
    background (NerfactoModelConfig.background_color): "last_sample" (default) blends rgb[..., -1, :] (1 - acc);
    "random" returns the composited colour without blending ("as if the background color was black"); "white" /
    "black" (or a 3-vector) blend a constant colour.  nan_to_num before, clamp to [0,1] after."""
    rgb = torch.nan_to_num(rgb)
    comp = torch.sum(weights[..., None] * rgb, dim=-2)
    acc = torch.sum(weights, dim=-1, keepdim=True)
    if isinstance(background, str) and background == "random":
        return torch.clamp(comp, 0.0, 1.0)
    if isinstance(background, str) and background == "last_sample":
        bg = rgb[..., -1, :]
    else:
        bg = torch.tensor(BACKGROUND_COLORS[background] if isinstance(background, str) else background,
                          dtype=torch.float32).expand(comp.shape)
    comp = comp + bg * (1.0 - acc)
    return torch.clamp(comp, 0.0, 1.0)


def render_accumulation(weights: torch.Tensor) -> torch.Tensor:
    return torch.sum(weights, dim=-1, keepdim=True)


def render_depth_median(weights: torch.Tensor, steps: torch.Tensor) -> torch.Tensor:
    cw = torch.cumsum(weights, dim=-1)
    split = torch.ones((*weights.shape[:-1], 1)) * 0.5
    idx = torch.searchsorted(cw, split, side="left")
    idx = torch.clamp(idx, 0, steps.shape[-1] - 1)
    return torch.gather(steps, dim=-1, index=idx)


def median_margin(weights: torch.Tensor) -> torch.Tensor:
    """Test diagnostic, not part of any output: how far the weight CDF of each ray stays from the 0.5 that
    render_depth_median searches for, min_j |cumsum(w)_j - 0.5| -> [R,1].  A ray whose margin is within the rounding of
    the CDF is a TIE: which of two neighbouring samples becomes the median depends on summation order, so two correct
    implementations may differ there by a whole sample step (the tests allow depth differences on such rays only)."""
    return (torch.cumsum(weights, dim=-1) - 0.5).abs().min(dim=-1, keepdim=True).values


def _note_margin(diagnostics: Optional[dict], weights: torch.Tensor) -> None:
    if diagnostics is not None:
        diagnostics.setdefault("median_margin", []).append(median_margin(weights))


def render_depth_expected(weights: torch.Tensor, steps: torch.Tensor) -> torch.Tensor:
    """clip bounds are the min/max of `steps` over the WHOLE chunk passed in (upstream quirk)."""
    depth = torch.sum(weights * steps, dim=-1, keepdim=True) / (torch.sum(weights, -1, keepdim=True) + 1e-10)
    return torch.clip(depth, steps.min(), steps.max())


def render_uncertainty(betas: torch.Tensor, weights: torch.Tensor) -> torch.Tensor:
    return torch.sum(weights * betas, dim=-1, keepdim=True)


# --------------------------------------------------------------------------
# counter-based RNG shared with the HIP kernels (this build's own definition)
# --------------------------------------------------------------------------

def _hash32(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.uint32)
    with np.errstate(over="ignore"):
        x = x ^ (x >> np.uint32(16))
        x = x * np.uint32(0x21F0AAAD)
        x = x ^ (x >> np.uint32(15))
        x = x * np.uint32(0x735A2D97)
        x = x ^ (x >> np.uint32(15))
    return x


GOLDEN = np.uint32(0x9E3779B9)


def mc_base(seed: int, pass_idx: int, sample_idx: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        key = _hash32(np.array([seed], dtype=np.uint32) + np.uint32(pass_idx) * GOLDEN)
        return _hash32(_hash32(sample_idx.astype(np.uint32)) + key)


MASK_LCG_A, MASK_LCG_C = np.uint32(25173), np.uint32(13849)


def _mask_step(x: np.ndarray) -> np.ndarray:
    """twin of unerf_mask_step: each 16-bit half steps as x -> 25173 x + 13849 (mod 2^16)"""
    x = x.astype(np.uint32)
    with np.errstate(over="ignore"):
        lo = ((x & np.uint32(0xFFFF)) * MASK_LCG_A + MASK_LCG_C) & np.uint32(0xFFFF)
        hi = ((x >> np.uint32(16)) * MASK_LCG_A + MASK_LCG_C) & np.uint32(0xFFFF)
    return (lo | (hi << np.uint32(16))).astype(np.uint32)


def _mask_word0(seed: int, sample_idx: np.ndarray, stream: int, n_pairs: int) -> np.ndarray:
    """[N, n_pairs] pass-0 mask words (twin of unerf_mc_pre / unerf_mc_base_h / unerf_mask_word0): pair j takes the base
    b_h = hash32(hash32(sample) + key + h GOLDEN) of h = bit 1 of j and the odd 24-bit constants of (stream, j & ~2):
    w = (b_h & 0xFFFFFF) A + (b_h >> 8) B  (mod 2^32)."""
    with np.errstate(over="ignore"):
        key = _hash32(np.array([seed], dtype=np.uint32))
        pre = _hash32(sample_idx.astype(np.uint32)) + key
        b = np.stack([_hash32(pre + np.uint32(h) * GOLDEN) for h in (0, 1)], axis=0)            # [2, N]
        j = np.arange(n_pairs, dtype=np.uint32)
        jc = j & ~np.uint32(2)
        A = (_hash32(np.uint32(0xA5A50000) + np.uint32(64 * stream) + jc) & np.uint32(0xFFFFFE)) | np.uint32(1)
        B = (_hash32(np.uint32(0x5A5A0000) + np.uint32(64 * stream) + jc) & np.uint32(0xFFFFFE)) | np.uint32(1)
        bb = b[(j >> np.uint32(1)) & np.uint32(1)].T                                               # [N, n_pairs]
        w = (bb & np.uint32(0xFFFFFF)) * A[None, :] + (bb >> np.uint32(8)) * B[None, :]
    return w.astype(np.uint32)


def mc_keep_mask(seed: int, pass_idx: int, sample_idx: np.ndarray, stream: int, n_units: int,
                 p_drop: float) -> np.ndarray:
    """[N, n_units] bool keep-mask.  stream 0 = density trunk, 1 = last colour layer, 2 = second colour layer, 3 = the
    head's inputs.  Mask word of unit pair j: pass 0 = _mask_word0 (a multilinear hash of the sample's base hash), pass
    k = _mask_step of pass k-1 (a 16-bit LCG per half).
    Low 16 bits -> unit 2j, high 16 bits -> unit 2j+1; keep iff the half read as a signed 16-bit number is below
    round((1-p)*65536) - 32768, i.e. iff (u16 ^ 0x8000) < round((1-p)*65536); p = 0 keeps every unit.
    (twin of unerf_mask_word0 / unerf_mask_step / unerf_keep_lo / unerf_keep_hi in csrc/unerf_common.hpp)"""
    assert n_units % 2 == 0 and n_units <= 128     # 64 pairs per stream: the word constants are indexed 64 stream + pair
    thr = np.uint32(int(round((1.0 - p_drop) * 65536.0)))
    r = _mask_word0(seed, sample_idx, stream, n_units // 2)
    for _ in range(pass_idx):
        r = _mask_step(r)
    lo = ((r & np.uint32(0xFFFF)) ^ np.uint32(0x8000)) < thr
    hi = ((r >> np.uint32(16)) ^ np.uint32(0x8000)) < thr
    return np.stack([lo, hi], axis=-1).reshape(r.shape[0], n_units)


def _xorshift32(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.uint32)
    x = x ^ (x << np.uint32(13))
    x = x ^ (x >> np.uint32(17))
    x = x ^ (x << np.uint32(5))
    return x


def normal_noise(seed: int, draw: int, sample_idx: np.ndarray) -> np.ndarray:
    """Standard normal per (draw, sample), fp32 -- twin of the kernel's generator used when no explicit noise tensor is
    supplied (laplace depth draws; csrc: unerf_depth_stream_seed / unerf_xorshift32 / unerf_normal_pair_from_state).
    Every sample owns one xorshift32 stream seeded by the counter hash of its global index (0 -> GOLDEN) and stepped
    once per draw PAIR; the 16-bit halves of the state are the two uniforms of a Box-Muller pair, the even draw takes
    the cosine branch, the odd one the sine."""
    x = mc_base(seed, 0, sample_idx)
    x = np.where(x == 0, GOLDEN, x).astype(np.uint32)
    for _ in range(draw >> 1):
        x = _xorshift32(x)
    u1 = ((x >> np.uint32(16)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 65536.0)
    u2 = ((x & np.uint32(0xFFFF)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 65536.0)
    rad = np.sqrt(np.float32(-2.0) * np.log(u1)).astype(np.float32)
    ang = np.float32(2.0 * math.pi) * u2
    return (rad * (np.sin(ang) if (draw & 1) else np.cos(ang))).astype(np.float32)


# --------------------------------------------------------------------------
# L1/L2 the reference's own fields and models
# --------------------------------------------------------------------------

@dataclass
class FieldParams:
    """Weights of one nerfacto-style field.  Layout mirrors the reference state-dict
    (SURVEY.md 8b); all Linear weights in torch [out,in] layout."""
    grid: GridMLP                      # table + trunk (2 Linear layers; laplace: see below)
    head_w: List[torch.Tensor]         # colour head Linear weights (3 layers: 63->64->64->3)
    head_b: List[torch.Tensor]
    appearance: torch.Tensor           # [32] constant eval embedding (mean or zeros)
    # laplace-only heads (laplace_field.py:139-160): base Linear(32,64) is grid.weights[0]
    hidden_w: Optional[torch.Tensor] = None   # mlp_hidden 64->15
    hidden_b: Optional[torch.Tensor] = None
    density_w: Optional[torch.Tensor] = None  # mlp_density 64->1
    density_b: Optional[torch.Tensor] = None
    average_init_density: float = 1.0
    beta_min: float = 0.01
    geo_feat_dim: int = 15
    sh_remap: bool = False             # tcnn SphericalHarmonics maps its [0,1] input back to [-1,1]
    density_activation: str = "exp"    # laplace only: "exp" (trunc_exp) | "softplus" (laplace_model.py:151)


def _color_inputs(directions: torch.Tensor, S: int, geo: torch.Tensor, appearance: torch.Tensor,
                  sh_remap: bool = False):
    """[UPSTREAM NerfactoField.get_outputs; re-implemented by the reference at
    laplace_field.py:378-398, 451-459]  directions [R,3], geo [R,S,15] -> h [R*S,63]"""
    R = directions.shape[0]
    dn = (directions + 1.0) / 2.0
    if sh_remap:   # tcnn SphericalHarmonics: x * 2 - 1 on its [0,1] input
        dn = dn * 2.0 - 1.0
    d = sh16(dn)[:, None, :].expand(R, S, 16)
    app = appearance.view(1, 1, -1).expand(R, S, -1)
    return torch.cat([d, geo, app], dim=-1).reshape(R * S, -1)


def active_field(origins, directions, euclid_bins, fp: FieldParams, autocast: Optional[torch.dtype] = None):
    """[REF activenerfacto_field.py:162-215]  -> density [R,S], rgb [R,S,3], beta [R,S]
    autocast: low-precision Linear layers (see _linear) -- the reference's default implementation="tcnn" runs both MLPs
    as fp16 FullyFusedMLPs (activenerfacto_field.py:89, 148-156)"""
    R, S = euclid_bins.shape[0], euclid_bins.shape[1] - 1
    pos = sample_positions(origins, directions, euclid_bins)
    p, sel = normalized_positions(pos, fp.grid.aabb)
    feat = grid_encode(p.reshape(-1, 3), fp.grid)
    h = mlp_forward(feat, fp.grid.weights, fp.grid.biases, autocast=autocast).view(R, S, -1)
    g = fp.geo_feat_dim
    dens_pre, geo, unc_pre = h[..., 0], h[..., 1:1 + g], h[..., 1 + g]
    density = fp.average_init_density * torch.exp(dens_pre) * sel
    beta = F.softplus(unc_pre) + fp.beta_min
    rgb = mlp_forward(_color_inputs(directions, S, geo, fp.appearance, fp.sh_remap), fp.head_w, fp.head_b, "sigmoid",
                      autocast=autocast)
    return density, rgb.view(R, S, 3), beta


def _linear(x, w, b, autocast: Optional[torch.dtype] = None):
    """nn.Linear, optionally the way torch.autocast runs it (mcdropout_models.py:86-92 forces autocast at eval: fp16 on
    a GPU, bf16 on the CPU): operands rounded to the low-precision type, fp32 accumulation, result rounded to that type.
    Used only to MEASURE how far the reference's autocast arithmetic is from its fp32 semantics (the semantics this
    build matches, DESIGN.md section 1 "Precision")."""
    if autocast is None:
        return F.linear(x, w, b)
    r = lambda t: t.to(autocast).to(torch.float32)
    return r(F.linear(r(x), r(w), r(b)))


def mcdropout_field(origins, directions, euclid_bins, fp: FieldParams, keep_trunk: Optional[torch.Tensor],
                    keep_head: Optional[torch.Tensor], p_drop: float, keep_head0: Optional[torch.Tensor] = None,
                    autocast: Optional[torch.dtype] = None, keep_in: Optional[torch.Tensor] = None):
    """[REF mcdropout_fields.py:110-174 + utils.py:6-43]
    trunk = Linear(32,64),ReLU,Dropout,Linear(64,16); head = Linear(63,64),ReLU,[Dropout,]Linear(64,64),ReLU,
    Dropout,Linear(64,3),Sigmoid.  keep_* are bool masks [R*S,64] (None = no Dropout module at that site):
    keep_trunk -- density_dropout_layers; keep_head0 -- rgb_dropout_layers contains 1 (in front of Linear 1);
    keep_head -- rgb_dropout_layers contains -1 / 2 (in front of the last Linear); keep_in [R*S,63] --
    rgb_dropout_layers contains 0 (in front of Linear 0: on the head's inputs [SH16 | geo15 | appearance32])."""
    R, S = euclid_bins.shape[0], euclid_bins.shape[1] - 1
    scale = 1.0 / (1.0 - p_drop)
    pos = sample_positions(origins, directions, euclid_bins)
    p, sel = normalized_positions(pos, fp.grid.aabb)
    feat = grid_encode(p.reshape(-1, 3), fp.grid)
    h = F.relu(_linear(feat, fp.grid.weights[0], fp.grid.biases[0], autocast))
    if keep_trunk is not None:
        h = h * keep_trunk.to(h.dtype) * scale
    out = _linear(h, fp.grid.weights[1], fp.grid.biases[1], autocast).view(R, S, -1)
    g = fp.geo_feat_dim
    density = fp.average_init_density * torch.exp(out[..., 0]) * sel
    geo = out[..., 1:1 + g]
    x = _color_inputs(directions, S, geo, fp.appearance, fp.sh_remap)
    if keep_in is not None:
        x = x * keep_in.to(x.dtype) * scale
    x = F.relu(_linear(x, fp.head_w[0], fp.head_b[0], autocast))
    if keep_head0 is not None:
        x = x * keep_head0.to(x.dtype) * scale
    x = F.relu(_linear(x, fp.head_w[1], fp.head_b[1], autocast))
    if keep_head is not None:
        x = x * keep_head.to(x.dtype) * scale
    rgb = torch.sigmoid(_linear(x, fp.head_w[2], fp.head_b[2], autocast))
    return density, rgb.view(R, S, 3)


def sample_laplace(weight_samples: torch.Tensor, activation: str, x: torch.Tensor, out_dim: int,
                   autocast: Optional[torch.dtype] = None):
    """[REF laplace_field.py:528-568] with the randn draw lifted out: `weight_samples` [n,P] are
    the already-formed mu + randn*std rows (weight [out,in] row-major, then bias).
    Sequential accumulation order as the reference loop."""
    n, P = weight_samples.shape
    in_dim = x.shape[-1]
    mu = 0.0
    mu2 = 0.0
    for s in range(n):
        w = weight_samples[s, : out_dim * in_dim].view(out_dim, in_dim)
        b = weight_samples[s, out_dim * in_dim:]
        pred = _linear(x, w, b, autocast)
        pred = {"exp": torch.exp, "softplus": F.softplus, "sigmoid": torch.sigmoid}[activation](pred)
        mu = mu + pred
        mu2 = mu2 + pred ** 2
    mu = mu / n
    mu2 = mu2 / n
    return mu, mu2 - mu ** 2


def laplace_weight_samples(mu_q: torch.Tensor, diag_ggn: torch.Tensor, prior_prec: float, eps: float,
                           noise: torch.Tensor) -> torch.Tensor:
    """[REF laplace_field.py:538-547] noise = the randn(n,P) draw."""
    std = 1 / torch.sqrt(diag_ggn + prior_prec + eps)
    return mu_q.view(1, -1) + noise * std.view(1, -1)


def laplace_field(origins, directions, euclid_bins, fp: FieldParams, ws_density: torch.Tensor,
                  ws_rgb: torch.Tensor, autocast: Optional[torch.dtype] = None):
    """[REF laplace_field.py:279-362, 365-485, 487-525] is_inference=True,
    use_deterministic_density=False.  Quirks kept: base_mlp is a bare Linear (no ReLU,
    utils.py:22-23); returned mu_d is NOT selector-masked (laplace_field.py:356-362).
    -> mu_d [R,S], var_d [R,S], mu_rgb [R,S,3], var_rgb [R,S] (relu, channel-mean)"""
    R, S = euclid_bins.shape[0], euclid_bins.shape[1] - 1
    pos = sample_positions(origins, directions, euclid_bins)
    p, _sel = normalized_positions(pos, fp.grid.aabb)
    feat = grid_encode(p.reshape(-1, 3), fp.grid)
    # autocast is an EMULATION switch for measuring the f16 kernels against: the reference itself runs these layers in
    # fp32 (`.float()` at laplace_field.py:305 and :460, no autocast wrapper around the Laplace model)
    hb = _linear(feat, fp.grid.weights[0], fp.grid.biases[0], autocast)
    geo = _linear(hb, fp.hidden_w, fp.hidden_b, autocast).view(R, S, -1)
    mu_d, var_d = sample_laplace(ws_density, fp.density_activation, hb, 1, autocast)
    x = _color_inputs(directions, S, geo, fp.appearance, fp.sh_remap)
    x = F.relu(_linear(x, fp.head_w[0], fp.head_b[0], autocast))
    x = F.relu(_linear(x, fp.head_w[1], fp.head_b[1], autocast))
    mu_rgb, var_rgb = sample_laplace(ws_rgb, "sigmoid", x, 3, autocast)
    var_rgb = F.relu(var_rgb).mean(dim=-1)
    return mu_d.view(R, S), var_d.view(R, S), mu_rgb.view(R, S, 3), var_rgb.view(R, S)


def laplace_field_deterministic(origins, directions, euclid_bins, fp: FieldParams):
    """[REF laplace_field.py:317-345, 462-465] is_inference=False path (plain forward)."""
    R, S = euclid_bins.shape[0], euclid_bins.shape[1] - 1
    pos = sample_positions(origins, directions, euclid_bins)
    p, sel = normalized_positions(pos, fp.grid.aabb)
    feat = grid_encode(p.reshape(-1, 3), fp.grid)
    hb = F.linear(feat, fp.grid.weights[0], fp.grid.biases[0])
    geo = F.linear(hb, fp.hidden_w, fp.hidden_b).view(R, S, -1)
    act = F.softplus if fp.density_activation == "softplus" else torch.exp
    density = act(F.linear(hb, fp.density_w, fp.density_b)).view(R, S) * sel
    x = _color_inputs(directions, S, geo, fp.appearance, fp.sh_remap)
    x = F.relu(F.linear(x, fp.head_w[0], fp.head_b[0]))
    x = F.relu(F.linear(x, fp.head_w[1], fp.head_b[1]))
    rgb = torch.sigmoid(F.linear(x, fp.head_w[2], fp.head_b[2]))
    return density, rgb.view(R, S, 3)


def laplace_ggn_diag(scene: "NerfScene", origins, directions):
    """[REF laplace_model.py:343-400 compute_hessian_naive, one batch]  Diagonal of the generalised
    Gauss-Newton matrix of the summed-MSE loss w.r.t. the two last layers (mlp_density: 64+1, mlp_rgb_ll:
    3*64+3 parameters, torch parameter order: weight row-major, then bias).  The reference obtains it from
    one GGN-vector product per unit vector (backpack); the loss Hessian w.r.t. the rendered pixel is 2 I,
    so diag = 2 * sum_{ray, channel} (d pred_rgb / d theta)^2 -- formed here with plain autograd, one
    backward per rendered value.  Forward = the is_inference=False path (`laplace_model.py:208-226`) with
    the eval-mode RGB renderer (eval_setup puts the pipeline in eval mode).  -> (ggn_density [65], ggn_rgb [195])"""
    import copy
    fp = copy.copy(scene.field)
    fp.density_w = scene.field.density_w.clone().requires_grad_(True)
    fp.density_b = scene.field.density_b.clone().requires_grad_(True)
    fp.head_w = list(scene.field.head_w)
    fp.head_b = list(scene.field.head_b)
    fp.head_w[2] = scene.field.head_w[2].clone().requires_grad_(True)
    fp.head_b[2] = scene.field.head_b[2].clone().requires_grad_(True)
    params = [fp.density_w, fp.density_b, fp.head_w[2], fp.head_b[2]]
    with torch.no_grad():
        eb, _, _ = _sample(scene, origins, directions)
    density, rgb = laplace_field_deterministic(origins, directions, eb, fp)
    w = get_weights(density, eb[..., 1:] - eb[..., :-1])
    pred = render_rgb(rgb, w, scene.background)
    diag = [torch.zeros_like(q) for q in params]
    flat = pred.reshape(-1)
    for i in range(flat.numel()):
        g = torch.autograd.grad(flat[i], params, retain_graph=True, allow_unused=True)
        for d, gi in zip(diag, g):
            if gi is not None:
                d += 2.0 * gi.detach() ** 2
    return (torch.cat([diag[0].reshape(-1), diag[1].reshape(-1)]),
            torch.cat([diag[2].reshape(-1), diag[3].reshape(-1)]))


@dataclass
class NerfScene:
    """Everything a nerfacto-family model needs at eval."""
    field: FieldParams
    prop_nets: List[GridMLP]
    near: float = 0.05
    far: float = 1000.0
    num_prop: Tuple[int, ...] = (256, 96)
    num_nerf: int = 48
    prop_average_init_density: float = 0.01
    uniform_spacing: bool = False          # proposal_initial_sampler="uniform" (UniformSampler) instead of "piecewise"
    background: object = "last_sample"     # background_color: "last_sample" | "random" | "white" | "black" | (r,g,b)


def _sample(scene: NerfScene, origins, directions, nears=None, fars=None):
    near = scene.near if nears is None else nears
    far = scene.far if fars is None else fars
    bins, wl, bl = proposal_sample(origins, directions, near, far, scene.prop_nets, scene.num_prop,
                                   scene.num_nerf, scene.prop_average_init_density, scene.uniform_spacing)
    eb = spacing_to_euclidean(bins, near, far, scene.uniform_spacing)
    return eb, wl, bl


def _prop_depths(scene, wl, bl, nears=None, fars=None):
    out = {}
    for i, (w, b) in enumerate(zip(wl, bl)):
        eb = spacing_to_euclidean(b, scene.near if nears is None else nears, scene.far if fars is None else fars,
                                  scene.uniform_spacing)
        out[f"prop_depth_{i}"] = render_depth_median(w, (eb[..., :-1] + eb[..., 1:]) / 2)
    return out


def active_compose(eb, density, rgb, beta, background="last_sample", diagnostics: Optional[dict] = None) -> Dict[str, torch.Tensor]:
    """[REF activenerfacto_model.py:94-127] everything get_outputs does after the field call: eb [R,S+1] Euclidean
    bin edges, density / beta [R,S], rgb [R,S,3].  Pinned to the reference's own code by
    tests/golden/nerf_model_glue.npz (fake-self run of ActiveNerfactoModel.get_outputs)."""
    deltas = eb[..., 1:] - eb[..., :-1]
    steps = (eb[..., :-1] + eb[..., 1:]) / 2
    w = get_weights(density, deltas)
    _note_margin(diagnostics, w)
    out = {
        "rgb": render_rgb(rgb, w, background),
        "accumulation": render_accumulation(w),
        "depth": render_depth_median(w, steps),
        "expected_depth": render_depth_expected(w, steps),
        "density": density,
    }
    beta = torch.nan_to_num(beta, 0.0) if torch.isnan(beta).any() else beta
    out["rgb_var"] = render_uncertainty(beta, w ** 2)
    out["rgb_std"] = out["rgb_var"].sqrt()
    out["depth_var"] = torch.sum(w * (steps - out["depth"]) ** 2, dim=-1, keepdim=True) + 1e-5
    out["depth_std"] = out["depth_var"].sqrt()
    return out


def active_outputs(scene: NerfScene, origins, directions, nears=None, fars=None,
                   autocast: Optional[torch.dtype] = None, diagnostics: Optional[dict] = None) -> Dict[str, torch.Tensor]:
    """[REF activenerfacto_model.py:83-152] one chunk of rays [R,3]; nears / fars [R,1]: per-ray planes of the bundle
    (obb_box), else the collider's constants.  autocast: the MAIN field's Linear layers in low precision (the proposal
    networks stay fp32 here as they do in the kernels)."""
    eb, wl, bl = _sample(scene, origins, directions, nears, fars)
    density, rgb, beta = active_field(origins, directions, eb, scene.field, autocast)
    out = active_compose(eb, density, rgb, beta, scene.background, diagnostics)
    out.update(_prop_depths(scene, wl, bl, nears, fars))
    return out


def nerfacto_pass_outputs(scene: NerfScene, origins, directions, eb, wl, bl, density, rgb, nears=None, fars=None,
                          diagnostics: Optional[dict] = None):
    """[UPSTREAM NerfactoModel.get_outputs] rgb/accumulation/depth/expected_depth/prop_depth_i."""
    deltas = eb[..., 1:] - eb[..., :-1]
    steps = (eb[..., :-1] + eb[..., 1:]) / 2
    w = get_weights(density, deltas)
    _note_margin(diagnostics, w)
    out = {
        "rgb": render_rgb(rgb, w, scene.background),
        "accumulation": render_accumulation(w),
        "depth": render_depth_median(w, steps),
        "expected_depth": render_depth_expected(w, steps),
    }
    out.update(_prop_depths(scene, wl, bl, nears, fars))
    return out


def nerfacto_outputs(scene: NerfScene, origins, directions, nears=None, fars=None) -> Dict[str, torch.Tensor]:
    """[UPSTREAM nerfstudio 1.1.0 NerfactoModel.get_outputs at eval] plain nerfacto, the member model of the reference's
    NeRF ensembles (ensemble_utils.py:149-150): NerfactoField = the mc-dropout field's graph without Dropout modules."""
    eb, wl, bl = _sample(scene, origins, directions, nears, fars)
    density, rgb = mcdropout_field(origins, directions, eb, scene.field, None, None, 0.0)
    return nerfacto_pass_outputs(scene, origins, directions, eb, wl, bl, density, rgb, nears, fars)


def mcdropout_outputs(scene: NerfScene, origins, directions, K: int, seed: int, p_drop: float,
                      ray_offset: int = 0, drop_sites: int = 5, autocast: Optional[torch.dtype] = None,
                      nears=None, fars=None, ray_ids: Optional[np.ndarray] = None,
                      diagnostics: Optional[dict] = None) -> Dict[str, torch.Tensor]:
    """[REF mcdropout_models.py:94-131] K stochastic passes of one chunk + mean / unbiased std.
    Masks come from the shared counter RNG keyed by the global sample index
    (ray_offset+r)*S+s, so chunking does not change them (ray_ids [R]: the rays' global indices when they are not
    a contiguous run)."""
    R = origins.shape[0]
    S = scene.num_nerf
    eb, wl, bl = _sample(scene, origins, directions, nears, fars)  # deterministic: identical in every pass
    rid = np.arange(R, dtype=np.int64) + ray_offset if ray_ids is None else np.asarray(ray_ids, dtype=np.int64)
    sidx = (rid[:, None] * S + np.arange(S)[None, :]).reshape(-1)
    outs = []
    passes: Optional[dict] = {} if diagnostics is not None else None
    for k in range(K):
        # drop_sites bits: 1 trunk (mask stream 0), 2 head hidden-0 (stream 2), 4 head hidden-1 (stream 1),
        # 8 head inputs (stream 3, 63 of its 64 units)
        fp = scene.field
        nH, nHC, nIN = fp.grid.weights[0].shape[0], fp.head_w[1].shape[0], fp.head_w[0].shape[1]   # 64, 64, 63 for nerfacto
        ev = lambda n: n + (n & 1)                                                                 # words gate unit PAIRS
        kt = torch.from_numpy(mc_keep_mask(seed, k, sidx, 0, ev(nH), p_drop)[:, :nH]) if drop_sites & 1 else None
        kh0 = torch.from_numpy(mc_keep_mask(seed, k, sidx, 2, ev(nHC), p_drop)[:, :nHC]) if drop_sites & 2 else None
        kh = torch.from_numpy(mc_keep_mask(seed, k, sidx, 1, ev(nHC), p_drop)[:, :nHC]) if drop_sites & 4 else None
        kin = torch.from_numpy(mc_keep_mask(seed, k, sidx, 3, ev(nIN), p_drop)[:, :nIN]) if drop_sites & 8 else None
        density, rgb = mcdropout_field(origins, directions, eb, scene.field, kt, kh, p_drop, keep_head0=kh0, autocast=autocast,
                                       keep_in=kin)
        outs.append(nerfacto_pass_outputs(scene, origins, directions, eb, wl, bl, density, rgb, nears, fars, passes))
    if diagnostics is not None:   # the smallest margin of the K passes: a tie in ANY pass moves the mean of the medians
        diagnostics.setdefault("median_margin", []).append(torch.stack(passes["median_margin"]).min(dim=0).values)
    res = {}
    for key in outs[0].keys():
        el = torch.stack([o[key] for o in outs], dim=0)
        res[key] = el.mean(dim=0)
        if key in ("rgb", "depth", "expected_depth"):
            res[key + "_std"] = el.std(dim=0).mean(dim=-1)[..., None]
    return res


def laplace_compose(eb, mu_d, var_d, mu_rgb, var_rgb, depth_noise: Optional[torch.Tensor],
                    use_deterministic_density: bool = False, background="last_sample",
                    diagnostics: Optional[dict] = None) -> Dict[str, torch.Tensor]:
    """[REF laplace_model.py:471-530] everything get_outputs_unc does after the field call: weights from mu_d,
    rgb / rgb_var from those weights, then (unless use_deterministic_density) D density draws
    relu(mu_d + max(sqrt(var_d), 1e-10) * noise), their mean weights, and depth / expected depth / accumulation
    from THOSE.  Pinned to the reference's own code by tests/golden/nerf_model_glue.npz."""
    deltas = eb[..., 1:] - eb[..., :-1]
    steps = (eb[..., :-1] + eb[..., 1:]) / 2
    w = get_weights(mu_d, deltas)
    rgb = render_rgb(mu_rgb, w, background)
    rgb_var = render_uncertainty(var_rgb, w ** 2)
    if use_deterministic_density:
        wm = w
    else:
        sd = torch.maximum(var_d.sqrt(), torch.tensor([1e-10]))
        sd = torch.nan_to_num(sd, nan=1e-10) if torch.isnan(sd).any() else sd
        sampled = F.relu(mu_d[None] + sd[None] * depth_noise)
        sw = torch.stack([get_weights(sampled[i], deltas) for i in range(sampled.shape[0])], dim=0)
        wm = sw.mean(dim=0)
    _note_margin(diagnostics, wm)
    depth = render_depth_median(wm, steps)
    depth_var = torch.sum(wm * (steps - depth) ** 2, dim=-1, keepdim=True) + 1e-5
    return {
        "rgb": rgb, "rgb_std": rgb_var.sqrt(), "accumulation": render_accumulation(wm), "depth": depth,
        "depth_std": depth_var.sqrt(), "expected_depth": render_depth_expected(wm, steps),
    }


def laplace_outputs(scene: NerfScene, origins, directions, ws_density, ws_rgb,
                    depth_noise: Optional[torch.Tensor], use_deterministic_density: bool = False,
                    nears=None, fars=None, autocast: Optional[torch.dtype] = None,
                    diagnostics: Optional[dict] = None) -> Dict[str, torch.Tensor]:
    """[REF laplace_model.py:456-556] is_inference=True.
    use_deterministic_density=False: density = sampled-head mean (NOT selector-masked), depth from the mean of the
    weights of D Normal(mu_d, sigma_d) density draws; depth_noise [D,R,S] = the standard-normal draw behind them.
    use_deterministic_density=True (laplace_field.py:501-506): density = the plain mean head, selector-masked
    (is_inference=False branch of get_density), colour still sampled, depth from the ordinary weights."""
    eb, wl, bl = _sample(scene, origins, directions, nears, fars)
    mu_d, var_d, mu_rgb, var_rgb = laplace_field(origins, directions, eb, scene.field, ws_density, ws_rgb, autocast)
    if use_deterministic_density:
        mu_d, _ = laplace_field_deterministic(origins, directions, eb, scene.field)
    out = laplace_compose(eb, mu_d, var_d, mu_rgb, var_rgb, depth_noise, use_deterministic_density, scene.background,
                          diagnostics)
    out.update(_prop_depths(scene, wl, bl, nears, fars))
    return out


def render_camera(chunk_fn, origins_hw, directions_hw, chunk: int = 1 << 15) -> Dict[str, torch.Tensor]:
    """[UPSTREAM Model.get_outputs_for_camera_ray_bundle; in-repo twin laplace_model.py:269-297]
    row-major chunk loop; chunk_fn(origins[R,3], dirs[R,3], ray_offset) -> dict."""
    H, W = origins_hw.shape[:2]
    o = origins_hw.reshape(-1, 3)
    d = directions_hw.reshape(-1, 3)
    lists: Dict[str, List[torch.Tensor]] = {}
    for i in range(0, H * W, chunk):
        out = chunk_fn(o[i:i + chunk], d[i:i + chunk], i)
        for k, v in out.items():
            lists.setdefault(k, []).append(v)
    return {k: torch.cat(v).view(H, W, -1) for k, v in lists.items()}


def ensemble_aggregate(outputs_list: List[Dict[str, torch.Tensor]]) -> Dict[str, torch.Tensor]:
    """[REF ensemble_pipeline.py:159-189]"""
    outputs = {}
    keys0 = outputs_list[0].keys()
    for k in keys0:
        el = torch.stack([o[k] for o in outputs_list], dim=0)
        outputs[k] = el.mean(dim=0)
        if "rgb_std" in keys0 and "depth_std" in keys0:
            if k in ("rgb", "depth"):
                alea = torch.stack([o[k + "_var"] for o in outputs_list], dim=0)
                outputs[k + "_var_alea"] = alea.mean(dim=0).mean(dim=-1).unsqueeze(-1)
                outputs[k + "_var_epi"] = el.var(dim=0).mean(dim=-1).unsqueeze(-1)
                outputs[k + "_var"] = outputs[k + "_var_epi"] + outputs[k + "_var_alea"]
                outputs[k + "_std"] = outputs[k + "_var"].sqrt()
        else:
            if k in ("rgb", "depth", "expected_depth"):
                outputs[k + "_std"] = el.std(dim=0).mean(dim=-1).unsqueeze(-1)
    return outputs


# --------------------------------------------------------------------------
# scene construction from a plain tensor dict (format: uncertainty-nerf-gs_amd/synthetic.py)
# --------------------------------------------------------------------------

def scene_from_tensors(t: dict) -> NerfScene:
    """Build the oracle-side scene from the same weight dict the device path is built from."""
    def grid(d):
        if d.get("w0") is None:      # use_linear=True proposal network: one Linear on the grid features
            ws, bs = [], []
        else:
            ws, bs = [d["w0"]], [d["b0"]]
        if "w1" in d and d.get("w1") is not None and not d.get("_laplace", False):
            ws.append(d["w1"])
            bs.append(d["b1"])
        return GridMLP(d["table"], d["scalings"], int(d["log2T"]), ws, bs, tcnn_levels=d.get("tcnn_levels"),
                       aabb=t.get("aabb"),   # scene box [2,3]: disable_scene_contraction
                       grid_half=d.get("tcnn_levels") is not None and t.get("grid_precision", "f32") == "f16")

    f = t["field"]
    lap = t["kind"] == "laplace"
    fd = dict(f)
    fd["_laplace"] = lap
    fp = FieldParams(grid=grid(fd), head_w=list(f["head_w"]), head_b=list(f["head_b"]), appearance=f["appearance"],
                     average_init_density=float(f.get("average_init_density", 1.0)),
                     beta_min=float(f.get("beta_min", 0.01)), sh_remap=bool(f.get("sh_remap", False)),
                     geo_feat_dim=int(f["w1"].shape[0]) - {"active": 2, "mcdropout": 1, "laplace": 0}[t["kind"]])
    if lap:
        fp.hidden_w, fp.hidden_b = f["w1"], f["b1"]
        fp.density_w, fp.density_b = f["density_w"], f["density_b"]
    return NerfScene(field=fp, prop_nets=[grid(p) for p in t["props"]], near=float(t["near"]), far=float(t["far"]),
                     num_prop=tuple(t["num_prop"]), num_nerf=int(t["num_nerf"]),
                     prop_average_init_density=float(t["prop_average_init_density"]),
                     uniform_spacing=t.get("proposal_initial_sampler", "piecewise") == "uniform",
                     background=t.get("background_color", "last_sample"))
