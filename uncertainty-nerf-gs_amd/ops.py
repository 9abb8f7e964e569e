"""torch-tensor wrappers around the libunerf C ABI -- one function per entry point.

torch is plumbing here: it owns device memory and the current HIP stream; every number is
produced by the HIP kernels in csrc/.  All tensors must be fp32 (or int32/int64 where
stated), contiguous and on a HIP device; nothing is silently copied to or from the host.
"""
from __future__ import annotations

import ctypes as C
import threading
import os
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch

from . import lib as _l


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ctx(device):
    device = torch.device(device)
    if device.type != "cuda":
        raise _l.UnerfError(f"expected a HIP device, got {device} (libunerf has no CPU path)")
    return torch.cuda.device(device)


class KernelTimer:
    """Optional per-entry-point timing with HIP events recorded on the launch stream (torch's current
    stream is the stream handed to libunerf).  bench.py uses it for the roofline figure."""

    def __init__(self, prealloc: int = 0):
        self.events = {}
        # event pairs made up front (hipEventCreate is the expensive half of an instrumented call: inside a ~1 ms frame of
        # ~15 entry points it showed); _run takes from here and falls back to making its own
        self.pool = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(prealloc)]

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, pairs in self.events.items():
            ms = [a.elapsed_time(b) for a, b in pairs]
            out[name] = {"launches": len(ms), "total_ms": float(sum(ms)), "avg_ms": float(sum(ms) / max(len(ms), 1))}
        return out


TIMER: Optional[KernelTimer] = None


def _run(name: str, rc_fn):
    """rc_fn() launches one libunerf entry point and returns its status code."""
    if TIMER is None:
        _l.check(rc_fn(), name)
        return
    a, b = TIMER.pool.pop() if TIMER.pool else (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    a.record()
    rc = rc_fn()
    b.record()
    _l.check(rc, name)
    TIMER.events.setdefault(name, []).append((a, b))


def _p(t: Optional[torch.Tensor], dtype=torch.float32, name: str = "tensor") -> Optional[int]:
    if t is None:
        return None
    if not t.is_cuda:
        raise _l.UnerfError(f"{name}: expected a HIP device tensor (libunerf has no CPU path)")
    if t.dtype != dtype:
        raise _l.UnerfError(f"{name}: dtype {t.dtype}, expected {dtype}")
    if not t.is_contiguous():
        raise _l.UnerfError(f"{name}: must be contiguous")
    return t.data_ptr()


def _host12(m: torch.Tensor):
    flat = [float(v) for v in m.detach().cpu().to(torch.float32).reshape(-1)[:12]]
    return (C.c_float * 12)(*flat)


# ------------------------------------------------------------------ rays ---------------

def _host_distortion(distortion):
    """camera `distortion_params` (k1, k2, k3, k4, p1, p2; nerfstudio's order) -> 6 host floats, or None"""
    if distortion is None:
        return None
    vals = [float(v) for v in torch.as_tensor(distortion).detach().cpu().to(torch.float32).reshape(-1)]
    if len(vals) != 6:
        raise _l.UnerfError(f"distortion: expected 6 parameters (k1, k2, k3, k4, p1, p2), got {len(vals)}")
    return (C.c_float * 6)(*vals)


def generate_rays(c2w: torch.Tensor, fx: float, fy: float, cx: float, cy: float, H: int, W: int, device,
                  ray_start: int = 0, count: Optional[int] = None, pixel_area: bool = False, distortion=None,
                  camera_type: int = _l.CAMERA_PERSPECTIVE):
    """-> origins [count,3], directions [count,3], (pixel_area [count,1] | None)
    distortion: the camera's 6 OPENCV lens parameters (include/unerf.h: unerf_generate_rays) or None
    camera_type: nerfstudio CameraType value (lib.CAMERA_*: perspective, fisheye, equirectangular, orthophoto)"""
    lib = _l.load()
    count = H * W - ray_start if count is None else count
    o = torch.empty(count, 3, device=device, dtype=torch.float32)
    d = torch.empty(count, 3, device=device, dtype=torch.float32)
    pa = torch.empty(count, 1, device=device, dtype=torch.float32) if pixel_area else None
    dist = _host_distortion(distortion)
    with _ctx(o.device):
        _run("generate_rays", lambda: lib.unerf_generate_rays(_host12(c2w), fx, fy, cx, cy, dist, int(camera_type), H, W, ray_start, count,
                                                              _p(o), _p(d), _p(pa), _stream()))
    return o, d, pa


def world_to_box(R, T) -> torch.Tensor:
    """inverse([R|T]) [3,4] of an OrientedBox (R [3,3], T [3]) -- the transform intersect_obb applies to the rays"""
    H = torch.eye(4, dtype=torch.float64)
    H[:3, :3] = torch.as_tensor(R, dtype=torch.float64).cpu()
    H[:3, 3] = torch.as_tensor(T, dtype=torch.float64).cpu().reshape(3)
    return torch.inverse(H)[:3].to(torch.float32).contiguous()


def ray_box_bins(origins, directions, w2b: torch.Tensor, S, near: float, far: float, sbins_row: torch.Tensor,
                 want_planes: bool = False, spacing: int = 0):
    """per-ray first-level spacing bins [R,n+1] for an oriented crop box (include/unerf.h: unerf_ray_box_bins);
    w2b = world_to_box(R, T), S = the box's edge lengths.  -> bins, (nears [R,1], fars [R,1]) | None"""
    lib = _l.load()
    R, n = origins.shape[0], sbins_row.numel() - 1
    out = torch.empty(R, n + 1, device=origins.device, dtype=torch.float32)
    nf = torch.empty(2, R, 1, device=origins.device, dtype=torch.float32) if want_planes else None
    half = (C.c_float * 3)(*[float(v) / 2 for v in torch.as_tensor(S).reshape(-1)[:3]])
    with _ctx(origins.device):
        _run("ray_box_bins", lambda: lib.unerf_ray_box_bins(_p(origins), _p(directions), R, _host12(w2b), half, near, far,
                                                            spacing, _p(sbins_row), n, _p(out), _p(nf[0]) if want_planes else None,
                                                            _p(nf[1]) if want_planes else None, _stream()))
    return out, ((nf[0], nf[1]) if want_planes else None)


def ray_planes_bins(nears, fars, near: float, far: float, sbins_row: torch.Tensor, spacing: int = 0) -> torch.Tensor:
    """per-ray first-level spacing bins [R,n+1] for a bundle that carries its own planes (RayBundle.nears / fars)"""
    lib = _l.load()
    nears, fars = nears.reshape(-1).contiguous(), fars.reshape(-1).contiguous()
    R, n = nears.shape[0], sbins_row.numel() - 1
    out = torch.empty(R, n + 1, device=nears.device, dtype=torch.float32)
    with _ctx(nears.device):
        _run("ray_planes_bins", lambda: lib.unerf_ray_planes_bins(_p(nears), _p(fars), R, near, far, spacing, _p(sbins_row), n, _p(out),
                                                                  _stream()))
    return out


# ------------------------------------------------------------- hash grid ---------------

def hashgrid_fwd(xyz: torch.Tensor, table: torch.Tensor, scalings: torch.Tensor, log2T: int,
                 return_indices: bool = False):
    lib = _l.load()
    N, L = xyz.shape[0], scalings.numel()
    out = torch.empty(N, 2 * L, device=xyz.device, dtype=torch.float32)
    idx = torch.empty(N, L, 8, device=xyz.device, dtype=torch.int32) if return_indices else None
    with _ctx(xyz.device):
        _run("hashgrid_fwd", lambda: lib.unerf_hashgrid_fwd(_p(xyz, name="xyz"), _p(table, name="table"), _p(scalings), N, L, log2T,
                                        _p(out), _p(idx, torch.int32), _stream()))
    return (out, idx) if return_indices else out


# ---- tiny-cuda-nn HashGrid layout (include/unerf.h: unerf_tcnn_level) -------------------------------

def tcnn_grid_levels(num_levels: int, base_res: int, per_level_scale: float, log2_hashmap_size: int):
    """Level records of a tcnn `HashGrid` encoding, computed the way tcnn does (float32 log2f / exp2f):
    -> list of (scale, res, offset, size, dense) with offset/size in rows of 2 features.
    nerfstudio's HashEncoding(implementation="tcnn") passes per_level_scale = exp((ln max_res - ln min_res)/(L-1))."""
    import numpy as np
    log2_pls = np.log2(np.float32(per_level_scale)).astype(np.float32)
    out, offset = [], 0
    for l in range(num_levels):
        scale = np.float32(np.exp2(np.float32(l) * log2_pls).astype(np.float32) * np.float32(base_res) - np.float32(1.0))
        res = int(np.ceil(scale)) + 1
        cube = res ** 3
        size = min((cube + 7) // 8 * 8, 1 << log2_hashmap_size)
        out.append((float(scale), res, offset, size, 1 if cube <= size else 0))
        offset += size
    return out


def tcnn_levels_ctypes(levels):
    return (_l.TcnnLevel * len(levels))(*[_l.TcnnLevel(*lv) for lv in levels])


def tcnn_levels_tensor(levels, device) -> torch.Tensor:
    """the same records as a device buffer (5 x 4 bytes per level) for unerf_density_net / unerf_field_params"""
    import struct
    raw = b"".join(struct.pack("<fIIII", *lv) for lv in levels)
    return torch.frombuffer(bytearray(raw), dtype=torch.int32).clone().to(device)


def hashgrid_fwd_tcnn(xyz: torch.Tensor, params: torch.Tensor, levels, return_indices: bool = False):
    """tcnn-layout lookup: params = flat fp32 `tcnn_encoding.params`, levels from tcnn_grid_levels"""
    lib = _l.load()
    N, L = xyz.shape[0], len(levels)
    out = torch.empty(N, 2 * L, device=xyz.device, dtype=torch.float32)
    idx = torch.empty(N, L, 8, device=xyz.device, dtype=torch.int32) if return_indices else None
    lv = tcnn_levels_ctypes(levels)
    with _ctx(xyz.device):
        _run("hashgrid_fwd_tcnn", lambda: lib.unerf_hashgrid_fwd_tcnn(_p(xyz, name="xyz"), _p(params, name="params"), lv, N, L,
                                                                     _p(out), _p(idx, torch.int32), _stream()))
    return (out, idx) if return_indices else out


def hashgrid_fwd_tcnn_half(xyz: torch.Tensor, params: torch.Tensor, levels) -> torch.Tensor:
    """tcnn-layout lookup in tcnn's own half arithmetic (include/unerf.h: unerf_hashgrid_fwd_tcnn_half).  params: the
    fp32 master vector (cast to half here, round to nearest even, as tcnn does before its forward pass) or a half tensor.
    -> [N, 2L] fp32 holding the half features"""
    lib = _l.load()
    N, L = xyz.shape[0], len(levels)
    ph = params.detach().reshape(-1, 2).to(torch.float16).contiguous()
    out = torch.empty(N, 2 * L, device=xyz.device, dtype=torch.float32)
    lv = tcnn_levels_ctypes(levels)
    with _ctx(xyz.device):
        _run("hashgrid_fwd_tcnn_half", lambda: lib.unerf_hashgrid_fwd_tcnn_half(_p(xyz, name="xyz"), _p(ph, torch.float16, "params"),
                                                                               lv, N, L, _p(out), _stream()))
    return out


# --------------------------------------------------------- parameter packs --------------

_HASH_P1, _HASH_P2 = 2654435761, 805459861


# A level of a proposal grid gets a dense x-paired copy (4 gather instructions per sample instead of 8) when that copy
# is at most this large.  48 MiB takes every level up to resolution 128 (the copy of a 128^3 level is 33.5 MB): the
# proposal kernels are bound by their texture-address unit, which pays per gather INSTRUCTION once neighbouring pixels
# share cache lines, so the bigger copies win although they no longer fit the L2 (same box, 1080p: first pass
# 8.17 -> 7.38 ms per frame, second 3.44 -> 3.22, with the round-1 cap of 6 MiB on the left).  Same values, same bits.
DENSE_LEVEL_BYTES = 48 << 20


def build_dense_pairs(table: torch.Tensor, scalings: torch.Tensor, log2T: int, max_level_bytes: int = DENSE_LEVEL_BYTES,
                      max_levels: int = 8):
    """Dense re-indexing of the coarse (prefix) levels of a hash grid (include/unerf.h, unerf_density_net):
    cell (x,y,z), x fastest, holds float4 = (table[hash(x,y,z)], table[hash(x+1,y,z)]).  Pure data movement:
    the kernel reads exactly the values the hashed lookup would.  -> (dense [cells,4] | None, offs, dims)"""
    T = 1 << log2T
    table = table.detach().to("cpu", torch.float32)
    chunks, offs, dims, cur = [], [], [], 0
    for l in range(min(scalings.numel(), max_levels)):
        dim = int(scalings[l].item()) + 1
        if dim ** 3 * 16 > max_level_bytes:
            break
        ax = torch.arange(dim, dtype=torch.int64)
        z, y, x = torch.meshgrid(ax, ax, ax, indexing="ij")           # x fastest after flatten
        hy, hz = y * _HASH_P1, z * _HASH_P2
        i0 = ((x ^ hy ^ hz) & (T - 1)).reshape(-1) + l * T
        i1 = ((torch.clamp(x + 1, max=dim - 1) ^ hy ^ hz) & (T - 1)).reshape(-1) + l * T
        chunks.append(torch.cat([table[i0], table[i1]], dim=-1))
        offs.append(cur)
        dims.append(dim)
        cur += dim ** 3
    if not chunks:
        return None, [], []
    return torch.cat(chunks, dim=0).contiguous(), offs, dims


def _aabb6(aabb):
    vals = [0.0] * 6 if aabb is None else [float(v) for v in aabb]
    assert len(vals) == 6
    return (C.c_float * 6)(*vals)


@dataclass
class DensityNetDev:
    """Device-resident proposal network (hash grid + Linear-ReLU-Linear), weights transposed."""
    table: torch.Tensor
    scalings: torch.Tensor
    log2T: int
    w0t: torch.Tensor
    b0: torch.Tensor
    w1t: torch.Tensor
    b1: torch.Tensor
    dense: Optional[torch.Tensor] = None
    dense_off: Tuple[int, ...] = ()
    dense_dim: Tuple[int, ...] = ()
    use_dense: bool = True
    tcnn_levels: Optional[torch.Tensor] = None   # device records: `table` is then a tcnn-layout parameter vector
    aabb: Optional[Tuple[float, ...]] = None     # 6 floats: scene-box normalisation instead of the contraction
    # tcnn layout only: "f16" = `table` is the HALF copy of the parameters ([rows, 2] float16) and the lookup runs in
    # tcnn's own half arithmetic (unerf_density_net.grid_half); "f32" = fp32 rows and blend
    grid_precision: str = "f32"

    @classmethod
    def from_torch(cls, table, scalings, log2T, w0, b0, w1, b1, device, tcnn_levels=None, grid_precision="f32"):
        f = lambda t: t.detach().to(device=device, dtype=torch.float32).contiguous()
        if grid_precision not in ("f32", "f16") or (grid_precision == "f16" and tcnn_levels is None):
            raise _l.UnerfError(f"grid_precision={grid_precision!r}: 'f32', or 'f16' with a tcnn-layout grid")
        if w0 is None:   # use_linear=True: one Linear on the grid features (unerf_density_net.hidden = 0)
            w0t, b0d = torch.zeros(0, device=device), torch.zeros(0, device=device)
        else:
            w0t, b0d = f(w0.t()), f(b0)
        if tcnn_levels is not None:
            sc = torch.zeros(len(tcnn_levels))
            tab = table.detach().reshape(-1, 2)
            tab = tab.to(device=device, dtype=torch.float16).contiguous() if grid_precision == "f16" else f(tab)
            return cls(tab, f(sc), int(log2T), w0t, b0d, f(w1.t()), f(b1),
                       tcnn_levels=tcnn_levels_tensor(tcnn_levels, device), grid_precision=grid_precision)
        dense, offs, dims = build_dense_pairs(table, scalings.detach().cpu(), int(log2T),
                                              max_level_bytes=int(os.environ.get("UNERF_DENSE_LEVEL_BYTES", DENSE_LEVEL_BYTES)))
        return cls(f(table), f(scalings), int(log2T), w0t, b0d, f(w1.t()), f(b1),
                   None if dense is None else f(dense), tuple(offs), tuple(dims))

    def cstruct(self) -> _l.DensityNet:
        nd = len(self.dense_off) if (self.use_dense and self.dense is not None and self.tcnn_levels is None) else 0
        offs = (C.c_int * 8)(*(list(self.dense_off[:nd]) + [0] * (8 - nd)))
        dims = (C.c_int * 8)(*(list(self.dense_dim[:nd]) + [0] * (8 - nd)))
        half = self.grid_precision == "f16"
        lin = self.b0.numel() == 0
        return _l.DensityNet(_p(self.table, torch.float16 if half else torch.float32, "table"), _p(self.scalings),
                             self.scalings.numel(), self.log2T, None if lin else _p(self.w0t),
                             None if lin else _p(self.b0), _p(self.w1t), _p(self.b1), self.b0.numel(),
                             _p(self.dense) if nd else None, nd, offs, dims, _p(self.tcnn_levels, torch.int32),
                             0 if self.aabb is None else 1, _aabb6(self.aabb), 1 if half else 0)


_warned_any_width = set()


def _warn_any_width(H, HC, G, Fp, L):
    key = (H, HC, G, Fp, L)
    if key not in _warned_any_width:
        _warned_any_width.add(key)
        import warnings
        warnings.warn(f"libunerf: field widths hidden={H}, hidden_color={HC}, geo_feat_dim={G}, features_per_level={Fp}, "
                      f"num_levels={L} differ from nerfacto's 64 / 64 / 15 / 2 / 16: rendering with the any-width kernel "
                      "(correct, 10-20 x slower than the matrix kernels)", stacklevel=3)


@dataclass
class FieldDev:
    """Device-resident main field.  `head0` is the first colour Linear (64x63) split as
    [SH16 | geo15 | appearance32]: the appearance block is folded into the bias with the
    constant eval embedding (NerfactoField.get_outputs eval branch; laplace_field.py:386-398)."""
    mode: int
    table: torch.Tensor
    scalings: torch.Tensor
    log2T: int
    w0t: torch.Tensor
    b0: torch.Tensor
    w1t: torch.Tensor
    b1: torch.Tensor
    h0t: torch.Tensor
    hb0: torch.Tensor
    h1t: torch.Tensor
    hb1: torch.Tensor
    h2t: torch.Tensor
    hb2: torch.Tensor
    average_init_density: float = 1.0
    beta_min: float = 0.01
    sh_remap: int = 0
    K: int = 0
    seed: int = 0
    p_drop: float = 0.2
    ws_density: Optional[torch.Tensor] = None
    ws_rgb: Optional[torch.Tensor] = None
    lap_mask_density: int = 0     # 1: use_deterministic_density (ws_density = copies of the mean row, selector-masked)
    mfma_blob: Optional[torch.Tensor] = None
    lap_blob: Optional[torch.Tensor] = None
    use_mfma: bool = True
    tcnn_levels: Optional[torch.Tensor] = None   # device records: `table` is then a tcnn-layout parameter vector
    mfma16_blob: Optional[torch.Tensor] = None   # split-f16 operands of the dense layers (pack_field_mfma16)
    lap16_blob: Optional[torch.Tensor] = None
    # "f16x2": split-f16 matrix kernels (fp32-equivalent, default); "fp32": exact fp32-input MFMA; "f16": ONE f16 product
    # per MAC with fp32 accumulation -- the reference's own eval precision (forced autocast, mcdropout_models.py:86-92;
    # tcnn FullyFusedMLP, activenerfacto_field.py:89), unerf_field_params.f16_single
    precision: str = "f16x2"
    packed_drop_scale: float = 1.0               # the inverted-dropout scale folded into mfma16_blob at pack time
    drop_sites: int = 0                          # UNERF_DROP_* bits (0 = reference default: trunk + last head layer)
    packed_drop_sites: int = 0                   # ... and the layers it was folded into
    packed_lap_softplus: int = -1                # the density activation lap16_blob's density rows were scaled for
    packed_mode: int = -1                        # the mode mfma16_blob was laid out for (MCDROPOUT: folded trunk-out slabs)
    lap_softplus: int = 0                        # LAPLACE: density_activation "softplus" instead of trunc_exp
    # LAPLACE, per-chunk last-layer samples (laplace_model.py:432-443: sample_laplace runs inside every eval chunk):
    # ws_density / ws_rgb [sets, n, P] and the blobs [sets, LAP_BLOB_FLOATS]; ray g uses set g // lap_chunk_rays.
    # 0: one set [n, P] for every ray
    lap_chunk_rays: int = 0
    aabb: Optional[Tuple[float, ...]] = None     # 6 floats: scene-box normalisation instead of the contraction
    # the first colour layer UNFOLDED -- [63][64] transposed weights, bias without the appearance term, the eval embedding:
    # read only when drop_sites contains DROP_HEADIN (dropout on the head's inputs; VALU kernel, include/unerf.h)
    h0_full_t: Optional[torch.Tensor] = None
    hb0_raw: Optional[torch.Tensor] = None
    app_embed: Optional[torch.Tensor] = None
    # tcnn layout only: "f16" = `table` is the HALF copy of the parameters ([rows, 2] float16) and the lookup runs in
    # tcnn's own half arithmetic (unerf_field_params.grid_half) -- what the reference's default implementation="tcnn"
    # computes; "f32" = fp32 rows and blend (also the only form of the torch layout)
    grid_precision: str = "f32"
    # widths other than nerfacto's 64 / 64 / 15 / 2 / (32): the any-width kernel (0 = the defaults; include/unerf.h)
    hidden: int = 0
    hidden_color: int = 0
    geo_dim: int = 0
    feat_per_level: int = 0
    app_dim: int = 0

    @property
    def any_width(self) -> bool:
        return bool(self.hidden)     # from_torch sets all five together

    @classmethod
    def from_torch(cls, mode, table, scalings, log2T, w0, b0, w1, b1, head_w, head_b, appearance, device,
                   tcnn_levels=None, **kw):
        f = lambda t: t.detach().to(device=device, dtype=torch.float32).contiguous()
        gp = kw.get("grid_precision", "f32")
        if gp not in ("f32", "f16") or (gp == "f16" and tcnn_levels is None):
            raise _l.UnerfError(f"grid_precision={gp!r}: 'f32', or 'f16' with a tcnn-layout grid")
        if tcnn_levels is not None:
            table, scalings = table.reshape(-1, 2), torch.zeros(len(tcnn_levels))
            kw["tcnn_levels"] = tcnn_levels_tensor(tcnn_levels, device)
        h0 = head_w[0].detach().to(torch.float32).contiguous()
        lap = mode == _l.FIELD_LAPLACE
        # widths (include/unerf.h: unerf_field_params.hidden ...): anything but nerfacto's runs the any-width kernel
        H, HC, AD = int(w0.shape[0]), int(head_w[1].shape[0]), int(appearance.numel())
        G = int(w1.shape[0]) - (0 if lap else (2 if mode == _l.FIELD_ACTIVE else 1))
        Fp = int(table.shape[-1]) if tcnn_levels is None else 2
        L = int(scalings.numel())
        if h0.shape != (HC, 16 + G + AD) or w1.shape[1] != H or w0.shape[1] != L * Fp or head_w[2].shape != (3, HC):
            raise _l.UnerfError(f"FieldDev: inconsistent layer shapes (trunk {tuple(w0.shape)} -> {tuple(w1.shape)}, head "
                                f"{tuple(h0.shape)} / {tuple(head_w[1].shape)} / {tuple(head_w[2].shape)}, appearance {AD}, grid {L} x {Fp})")
        headin = mode == _l.FIELD_MCDROPOUT and bool(int(kw.get("drop_sites", 0)) & _l.DROP_HEADIN)
        generic = (H, HC, G, Fp, L) != (64, 64, 15, 2, 16) or (AD != 32 and headin)
        if generic:
            _warn_any_width(H, HC, G, Fp, L)
            kw.update(hidden=H, hidden_color=HC, geo_dim=G, feat_per_level=Fp, app_dim=AD)
        # appearance block folded into the bias; accumulated in float64 so the result does not depend on the
        # memory layout the weights arrived in (a strided view takes a different CPU matmul path)
        hb0 = (head_b[0].detach().double() + h0[:, 16 + G:].double() @ appearance.detach().double()).to(torch.float32)
        kw["packed_drop_scale"] = cls._drop_scale(mode, int(kw.get("K", 0)), float(kw.get("p_drop", 0.2)))
        kw["packed_drop_sites"] = cls._sites(int(kw.get("drop_sites", 0)))
        blob = blob16 = None
        if not generic:
            blob = f(pack_field_mfma(w0, b0, w1, b1, h0[:, :31], hb0, head_w[1], head_b[1], head_w[2], head_b[2],
                                     geo_first_unit=0 if lap else 1))
            blob16 = pack_field_mfma16(w0, b0, w1, b1, h0[:, :31], hb0, head_w[1], head_b[1], head_w[2], head_b[2],
                                       geo_first_unit=0 if lap else 1, drop_scale=kw["packed_drop_scale"],
                                       drop_sites=kw["packed_drop_sites"],
                                       fold_trunk=mode == _l.FIELD_MCDROPOUT and bool(_l.load().unerf_build_flags() & _l.BUILD_TRUNK_FOLD))
        kw["packed_mode"] = mode
        kw["mfma16_blob"] = None if blob16 is None else f(blob16)   # None (weights beyond the f16 range): exact kernels
        lap_blob = None
        if lap and kw.get("ws_density") is not None and kw["ws_density"].dim() == 3 and not kw.get("lap_chunk_rays"):
            raise _l.UnerfError("FieldDev: stacked Laplace sample sets [sets, n, P] need lap_chunk_rays (rays per set)")
        if lap and not generic and kw.get("ws_density") is not None and max(kw["ws_density"].shape[-2], kw["ws_rgb"].shape[-2]) <= 32 * LAP_BLOCKS:
            wsd, wsr = kw["ws_density"], kw["ws_rgb"]
            stacked = wsd.dim() == 3
            lap_blob, lap16 = pack_laplace_sets(wsd if stacked else wsd[None], wsr if stacked else wsr[None], device,
                                                exp2_rows=bool(_l.load().unerf_build_flags() & _l.BUILD_LAP_EXP2),
                                                softplus=bool(kw.get("lap_softplus", 0)))
            lap_blob = lap_blob.contiguous() if stacked else lap_blob[0].contiguous()
            kw["lap16_blob"] = None if lap16 is None else (lap16 if stacked else lap16[0].contiguous())
            kw["packed_lap_softplus"] = int(bool(kw.get("lap_softplus", 0)))
        if kw.get("ws_density") is not None:
            kw["ws_density"], kw["ws_rgb"] = f(kw["ws_density"]), f(kw["ws_rgb"])
        tab = table.detach().to(device=device, dtype=torch.float16).contiguous() if gp == "f16" else f(table)
        return cls(mode, tab, f(scalings), int(log2T), f(w0.t()), f(b0), f(w1.t()), f(b1),
                   f(h0[:, :16 + G].t()), f(hb0), f(head_w[1].t()), f(head_b[1]), f(head_w[2].t()), f(head_b[2]),
                   mfma_blob=blob, lap_blob=lap_blob, h0_full_t=f(h0.t()), hb0_raw=f(head_b[0]), app_embed=f(appearance),
                   **kw)

    @staticmethod
    def _drop_scale(mode: int, K: int, p_drop: float) -> float:
        """1/(1-p) when the kernels generate dropout masks (MCDROPOUT, K > 0), else 1"""
        return 1.0 / (1.0 - p_drop) if (mode == _l.FIELD_MCDROPOUT and K > 0) else 1.0

    @staticmethod
    def _sites(drop_sites: int) -> int:
        return drop_sites if drop_sites else (_l.DROP_TRUNK | _l.DROP_HEAD1)

    def cstruct(self) -> _l.FieldParams:
        if self.any_width and self.precision == "f16":
            raise _l.UnerfError("FieldDev.precision='f16': the any-width kernel computes in fp32 (use 'f16x2' or 'fp32')")
        if self.precision not in ("f16x2", "fp32", "f16"):
            raise _l.UnerfError(f"FieldDev.precision={self.precision!r}: expected 'f16x2', 'fp32' or 'f16'")
        h16 = self.precision in ("f16x2", "f16")
        use16 = self.use_mfma and h16 and self.mfma16_blob is not None
        if self.precision == "f16" and (not use16 or (self.mode == _l.FIELD_LAPLACE and self.lap16_blob is None)):
            raise _l.UnerfError("FieldDev.precision='f16' needs the f16 operand blobs (weights inside the f16 range, "
                                "use_mfma=True, at most 128 Laplace samples); use 'f16x2' or 'fp32'")
        if use16 and (abs(self._drop_scale(self.mode, self.K, self.p_drop) - self.packed_drop_scale) > 1e-7
                      or (self.packed_drop_scale != 1.0 and self._sites(self.drop_sites) != self.packed_drop_sites)):
            # the split-f16 operands carry 1/(1-p) inside two weight matrices: K (0 <-> > 0) or p_drop changed since
            # from_torch().  The exact kernels apply the scale at run time, so the two paths would silently disagree.
            raise _l.UnerfError(
                f"FieldDev: K={self.K}, p_drop={self.p_drop} do not match the dropout scale {self.packed_drop_scale:.6g} "
                "packed into mfma16_blob; rebuild the FieldDev (from_torch) after changing K or p_drop")
        if (use16 and self.lap16_blob is not None and self.packed_lap_softplus >= 0
                and self.packed_lap_softplus != int(bool(self.lap_softplus))):
            raise _l.UnerfError("FieldDev: lap_softplus changed since from_torch(): lap16_blob's density rows carry (or "
                                "lack) the log2(e) factor of the other activation; rebuild the FieldDev")
        if use16 and self.packed_mode >= 0 and (self.packed_mode == _l.FIELD_MCDROPOUT) != (self.mode == _l.FIELD_MCDROPOUT):
            raise _l.UnerfError("FieldDev: mfma16_blob was laid out for another mode; rebuild the FieldDev (from_torch)")
        half = self.grid_precision == "f16"
        if half and self.tcnn_levels is None:
            raise _l.UnerfError("FieldDev.grid_precision='f16' needs a tcnn-layout grid (tcnn_levels)")
        cs = _l.FieldParams(
            self.mode, _p(self.table, torch.float16 if half else torch.float32, "table"), _p(self.scalings), self.scalings.numel(), self.log2T,
            _p(self.w0t), _p(self.b0), _p(self.w1t), _p(self.b1), self.b1.numel(),
            _p(self.h0t), _p(self.hb0), _p(self.h1t), _p(self.hb1), _p(self.h2t), _p(self.hb2),
            self.average_init_density, self.beta_min, self.sh_remap, self.K, self.seed & 0xFFFFFFFF, self.p_drop,
            _p(self.ws_density), _p(self.ws_rgb), 0 if self.ws_density is None else self.ws_density.shape[-2],
            int(self.lap_mask_density),
            _p(self.mfma_blob) if self.use_mfma else None, _p(self.lap_blob) if self.use_mfma else None,
            _p(self.tcnn_levels, torch.int32),
            _p(self.mfma16_blob) if (self.use_mfma and h16) else None,
            _p(self.lap16_blob) if (self.use_mfma and h16) else None, 0, 0, int(self.drop_sites), int(self.lap_softplus),
            0 if self.aabb is None else 1, _aabb6(self.aabb), 1 if self.precision == "f16" else 0, None,
            _p(self.h0_full_t), _p(self.hb0_raw), _p(self.app_embed),
            int(self.lap_chunk_rays) if (self.ws_density is not None and self.ws_density.dim() == 3) else 0,
            self.ws_density.shape[0] if (self.ws_density is not None and self.ws_density.dim() == 3) else 1,
            0 if self.ws_rgb is None else self.ws_rgb.shape[-2])
        cs.grid_half = 1 if half else 0
        cs.hidden, cs.hidden_color, cs.geo_dim = int(self.hidden), int(self.hidden_color), int(self.geo_dim)
        cs.feat_per_level, cs.app_dim = int(self.feat_per_level), int(self.app_dim)
        return cs


# ---- MFMA operand packing for field_kernel_mfma (csrc/unerf_nerf.hip) -------------------------
# v_mfma_f32_32x32x2_f32 computes D[32x32] += A[32x2] B[2x32]; lane l holds A[i=l&31][k=l>>5],
# B[k=l>>5][j=l&31] and D rows (r&3)+8(r>>2)+4(l>>5) of column l&31 in its 16 accumulator
# registers r.  The kernel keeps samples on the columns (lanes) and layer units on the rows, so a
# layer's accumulators ARE the next layer's B operands with no data movement; the weights are
# pre-arranged here, once, into one 64-float "A fragment" per MFMA in exactly the (step, lane)
# order the kernel consumes them from LDS.

MFMA_FRAGS = 160            # L0: 32, trunk-out: 32, colour-0: 32, colour-1: 64
MFMA_BIAS_OFF = MFMA_FRAGS * 64
MFMA_H2_OFF = MFMA_BIAS_OFF + 7 * 32
MFMA_BLOB_FLOATS = MFMA_H2_OFF + 2 * 2 * 3 * 16 + 4
MFMA16_BLOB_FLOATS = MFMA_BLOB_FLOATS + 4 * 64 * 8 // 2     # include/unerf.h: UNERF_MFMA16_BLOB_FLOATS


def _mfma_unit(r, h):
    """accumulator register r of a lane in half h holds output row (unit) ..."""
    return (r & 3) + 8 * (r >> 2) + 4 * h


def pack_field_mfma(w0, b0, w1, b1, h0, hb0, h1, hb1, h2, hb2, geo_first_unit: int = 1) -> torch.Tensor:
    """torch-layout ([out,in]) CPU tensors -> flat fp32 blob [MFMA_BLOB_FLOATS].
    w0 [64,32]; w1 [out1<=32,64]; h0 [64,31] (cols: SH16 | geo15, appearance already folded into hb0);
    h1 [64,64]; h2 [3,64].  geo_first_unit: trunk-out row of geo feature 0 (1 when row 0 is the density
    logit -- active / mc-dropout; 0 for the laplace `mlp_hidden`, whose 15 rows are all geo)."""
    f = lambda t: t.detach().to("cpu", torch.float32)
    w0, b0, w1, b1, h0, hb0, h1, hb1, h2, hb2 = map(f, (w0, b0, w1, b1, h0, hb0, h1, hb1, h2, hb2))
    out1 = w1.shape[0]
    assert w0.shape == (64, 32) and w1.shape[1] == 64 and out1 <= 32 and h0.shape == (64, 31)
    assert h1.shape == (64, 64) and h2.shape == (3, 64)
    lane = torch.arange(64)
    i, h = lane & 31, lane >> 5
    blob = torch.zeros(MFMA_BLOB_FLOATS)
    fr = blob[:MFMA_BIAS_OFF].view(MFMA_FRAGS, 64)
    w1p = torch.zeros(32, 64)
    w1p[:out1] = w1
    b1p = torch.zeros(32)
    b1p[:out1] = b1
    for blk in range(2):
        for s in range(16):
            fr[blk * 16 + s] = w0[32 * blk + i, 16 * h + s]                      # L0: input 16h+s
    for bi in range(2):
        for r in range(16):
            fr[32 + bi * 16 + r] = w1p[i, 32 * bi + _mfma_unit(r, h)]              # trunk-out
    for blk in range(2):
        for s in range(8):                                                         # geo rows of trunk-out
            g = _mfma_unit(s, h) - geo_first_unit                                      # geo feature index of this row
            ok = (g >= 0) & (g < 15)
            fr[64 + blk * 16 + s] = torch.where(ok, h0[32 * blk + i, (16 + g).clamp(16, 30)], torch.zeros(64))
        for s in range(8, 16):                                                     # SH components 8h+s-8
            fr[64 + blk * 16 + s] = h0[32 * blk + i, 8 * h + (s - 8)]
    for blk in range(2):
        for bi in range(2):
            for r in range(16):
                fr[96 + blk * 32 + bi * 16 + r] = h1[32 * blk + i, 32 * bi + _mfma_unit(r, h)]
    bias = blob[MFMA_BIAS_OFF:MFMA_H2_OFF].view(7, 2, 16)
    r16 = torch.arange(16)
    for k, vec in enumerate((b0[:32], b0[32:], b1p, hb0[:32], hb0[32:], hb1[:32], hb1[32:])):
        for hh in range(2):
            bias[k, hh] = vec[_mfma_unit(r16, hh)]
    hh2 = blob[MFMA_H2_OFF:MFMA_H2_OFF + 192].view(2, 2, 3, 16)
    for blk in range(2):
        for hh in range(2):
            for c in range(3):
                hh2[blk, hh, c] = h2[c, 32 * blk + _mfma_unit(r16, hh)]
    blob[MFMA_H2_OFF + 192:MFMA_H2_OFF + 195] = hb2
    return blob


# ---- split-f16 operand packing for field_kernel_mfma16 --------------------------------------------
# v_mfma_f32_32x32x16_f16 (gfx950): D[32x32] += A[32x16] B[16x32] with fp32 accumulation; lane l holds
# A[row l&31][k = 8(l>>5) + e], B[k = 8(l>>5) + e][col l&31], e = 0..7 (one 16-byte register quad each).
# Every fp32 weight w is stored as two halves hi = f16(w), lo = f16(w - hi) (22 mantissa bits together) and
# the kernel splits the activations the same way; hi*hi + hi*lo + lo*hi is accumulated in fp32 (the dropped
# lo*lo term is 2^-22 relative).  Three f16 MFMAs at 16x the fp32-MFMA rate replace eight fp32 MFMAs.
# The accumulator registers 8s..8s+7 of a lane in half g hold layer units 16s + 4g + (e&3) + 8(e>>2): that is
# the k order of the next layer's B operand, so it is the k order the A fragments are packed in.
F16_OPERAND_LIMIT = 6.0e4  # |operand| must stay below f16 max (65504) for the hi half to be finite
MF16_SLABS = 20            # L0: 4 (step, block), trunk-out: 4 steps, colour-0: 4 (geo|SH, block), colour-1: 8
MF16_SLAB_FLOATS = 512     # 64 lanes x 8 halves x (hi, lo) = 2 KiB
# the 64 -> 3 colour layer as four MORE single-operand slabs (hi halves only, [step][lane][8 halves], rows 0..2 = r, g, b,
# rows 3..31 zero) behind the fp32 tail: read by the "f16" form only (unerf_field_params.f16_single), which runs that layer
# on the matrix pipe too
# (offset MFMA_BLOB_FLOATS, MFMA16_BLOB_FLOATS - MFMA_BLOB_FLOATS floats)


def _mf16_unit(s, g, e):
    """layer unit held by accumulator register 8s+e of a lane in half g"""
    return 16 * s + 4 * g + (e & 3) + 8 * (e >> 2)


def _split_f16(w: torch.Tensor):
    hi = w.to(torch.float16)
    lo = (w - hi.to(torch.float32)).to(torch.float16)
    return hi, lo


def pack_field_mfma16(w0, b0, w1, b1, h0, hb0, h1, hb1, h2, hb2, geo_first_unit: int = 1,
                      drop_scale: float = 1.0, drop_sites: int = 5, fold_trunk: bool = False) -> torch.Tensor:
    """Same arguments and same blob size as pack_field_mfma; the first 10240 floats hold the 20 split-f16
    A-operand slabs [slab][hi|lo][lane][8 halves], the bias rows and the rgb layer follow (fp32).
    drop_scale = 1/(1-p) when MC-dropout masks are applied in front of the trunk-out and rgb layers: the
    inverted-dropout scale is folded into those two weight matrices (the kernel then only zeroes units).
    fold_trunk (MCDROPOUT with UNERF_BUILD_TRUNK_FOLD; needs <= 16 trunk-out rows): the second operand of the four
    trunk-out slabs holds rows 0..15 = W_hi, rows 16..31 = W_lo, so ONE MFMA against the activations' hi halves yields
    both W_hi a_hi (rows 0..15) and W_lo a_hi (rows 16..31 = the same lane's registers 8..15)."""
    f = lambda t: t.detach().to("cpu", torch.float32)
    w0f, w1f, h0f, h1f = map(f, (w0, w1, h0, h1))
    h2 = f(h2)
    # the scale sits in the layer BEHIND each active dropout site (include/unerf.h: UNERF_DROP_*)
    if drop_sites & 1:
        w1f = w1f * float(drop_scale)
    if drop_sites & 2:
        h1f = h1f * float(drop_scale)
    out1 = w1f.shape[0]
    assert w0f.shape == (64, 32) and w1f.shape[1] == 64 and out1 <= 32 and h0f.shape == (64, 31) and h1f.shape == (64, 64)
    if max(float(abs(w).max()) for w in (w0f, w1f, h0f, h1f)) >= F16_OPERAND_LIMIT:
        return None   # outside the f16 operand range: the caller stays on the exact fp32 kernels
    lane = torch.arange(64)
    row, g = (lane & 31)[:, None], (lane >> 5)[:, None]
    e = torch.arange(8)[None, :]
    w1p = torch.zeros(32, 64)
    w1p[:out1] = w1f
    slabs = torch.zeros(MF16_SLABS, 64, 8)
    for s in range(2):                                   # L0: half g feeds inputs 16g + (8s + e)
        for b in range(2):
            slabs[2 * s + b] = w0f[32 * b + row, 16 * g + 8 * s + e]
    for s in range(4):                                   # trunk-out: 64 hidden units, accumulator order
        slabs[4 + s] = w1p[row, 32 * (s >> 1) + _mf16_unit(s & 1, g, e)]
    for b in range(2):
        gi = _mf16_unit(0, g, e) - geo_first_unit          # colour-0, geo rows of the trunk output (regs 0..7)
        ok = (gi >= 0) & (gi < 15)
        slabs[8 + b] = torch.where(ok, h0f[32 * b + row, (16 + gi).clamp(16, 30)], torch.zeros(64, 8))
        slabs[10 + b] = h0f[32 * b + row, 8 * g + e]       # colour-0, SH components 8g + e
    for s in range(4):                                   # colour-1
        for b in range(2):
            slabs[12 + 2 * s + b] = h1f[32 * b + row, 32 * (s >> 1) + _mf16_unit(s & 1, g, e)]
    hi, lo = _split_f16(slabs)
    if fold_trunk:   # second operand of the trunk-out slabs: rows 0..15 = W_hi, rows 16..31 = W_lo of row - 16
        assert out1 <= 16, "fold_trunk: the trunk output must fit 16 MFMA rows"
        for s in range(4):
            lo[4 + s] = torch.where(row < 16, hi[4 + s], lo[4 + s][(lane & 15) + 32 * (lane >> 5)])
    frag = torch.stack([hi, lo], dim=1).contiguous()      # [slab][hi|lo][lane][8]
    head = frag.view(torch.int16).reshape(-1).view(torch.float32)
    assert head.numel() == MF16_SLABS * MF16_SLAB_FLOATS == MFMA_BIAS_OFF
    h2s = h2 * (float(drop_scale) if drop_sites & 4 else 1.0)
    tail = pack_field_mfma(w0, b0, w1, b1, h0, hb0, h1, hb1, h2s, hb2, geo_first_unit)[MFMA_BIAS_OFF:]
    if float(h2s.abs().max()) >= F16_OPERAND_LIMIT:
        return None
    c2 = torch.zeros(4, 64, 8)                            # colour-2: rows 0..2 of a 32-row block, accumulator order
    h2p = torch.zeros(32, 64)
    h2p[:3] = h2s
    for s in range(4):
        c2[s] = h2p[row, 32 * (s >> 1) + _mf16_unit(s & 1, g, e)]
    c2f = c2.to(torch.float16).contiguous().view(torch.int16).reshape(-1).view(torch.float32)
    assert c2f.numel() == MFMA16_BLOB_FLOATS - MFMA_BLOB_FLOATS
    return torch.cat([head, tail, c2f])


# sampled last layers of the Laplace field as MFMA A fragments: rows = weight samples (padded to 128
# per output channel), K = the 64 hidden units in accumulator-register order.
LAP_BLOCKS = 4                                   # 128 rows per output channel
LAP_FRAGS = (1 + 3) * LAP_BLOCKS * 32            # density + 3 colour channels
LAP_BIAS_OFF = LAP_FRAGS * 64
LAP_BLOB_FLOATS = LAP_BIAS_OFF + (1 + 3) * LAP_BLOCKS * 32
LAP_PAD_BIAS = -1e30                             # exp / sigmoid of a padded row contributes exactly 0


def pack_laplace_heads(ws_density: torch.Tensor, ws_rgb: torch.Tensor) -> torch.Tensor:
    """ws_density [n,65], ws_rgb [n,195] (rows = mu + randn*std, torch [out,in] row-major then bias;
    laplace_field.py:538-547) -> flat fp32 blob [LAP_BLOB_FLOATS].  Head q: 0 = density, 1..3 = r,g,b."""
    wd = ws_density.detach().to("cpu", torch.float32)
    wr = ws_rgb.detach().to("cpu", torch.float32)
    n = wd.shape[0]
    assert wd.shape == (n, 65) and wr.shape == (n, 195) and 1 <= n <= 32 * LAP_BLOCKS
    heads_w = [wd[:, :64]] + [wr[:, c * 64:(c + 1) * 64] for c in range(3)]
    heads_b = [wd[:, 64]] + [wr[:, 192 + c] for c in range(3)]
    lane = torch.arange(64)
    i, h = lane & 31, lane >> 5
    r16 = torch.arange(16)
    blob = torch.zeros(LAP_BLOB_FLOATS)
    fr = blob[:LAP_BIAS_OFF].view(4, LAP_BLOCKS, 2, 16, 64)
    bias = blob[LAP_BIAS_OFF:].view(4, LAP_BLOCKS, 2, 16)
    for q in range(4):
        W = torch.zeros(32 * LAP_BLOCKS, 64)
        W[:n] = heads_w[q]
        bv = torch.full((32 * LAP_BLOCKS,), LAP_PAD_BIAS)
        bv[:n] = heads_b[q]
        for b in range(LAP_BLOCKS):
            for bi in range(2):
                for r in range(16):
                    fr[q, b, bi, r] = W[32 * b + i, 32 * bi + _mfma_unit(r, h)]
            for hh in range(2):
                bias[q, b, hh] = bv[32 * b + _mfma_unit(r16, hh)]
    return blob


LOG2E = 1.4426950408889634


def pack_laplace_heads16(ws_density: torch.Tensor, ws_rgb: torch.Tensor, exp2_rows: bool = False,
                         softplus: bool = False) -> torch.Tensor:
    """Split-f16 form of pack_laplace_heads (same size, same bias layout): fragments
    [head q][block b][k-step s][hi|lo][lane][8 halves], k order = accumulator order of the 64 hidden units.
    exp2_rows (UNERF_BUILD_LAP_EXP2): the rows (weights and bias) carry the base change of the activation behind
    them, so that the kernel's epilogue is the bare hardware exp2 -- density rows x log2(e) (exp(x) = 2^(x log2 e); left
    alone when the density activation is softplus), colour rows x -log2(e) (sigmoid(x) = 1 / (1 + 2^(-x log2 e)))."""
    wd = ws_density.detach().to("cpu", torch.float32)
    wr = ws_rgb.detach().to("cpu", torch.float32)
    if exp2_rows:
        wd = (wd.double() * (1.0 if softplus else LOG2E)).to(torch.float32)
        wr = (wr.double() * -LOG2E).to(torch.float32)
        ws_density, ws_rgb = wd, wr
    n = wd.shape[0]
    assert wd.shape == (n, 65) and wr.shape == (n, 195) and 1 <= n <= 32 * LAP_BLOCKS
    if max(float(abs(wd).max()), float(abs(wr).max())) >= F16_OPERAND_LIMIT:
        return None
    heads_w = [wd[:, :64]] + [wr[:, c * 64:(c + 1) * 64] for c in range(3)]
    lane = torch.arange(64)
    row, g = (lane & 31)[:, None], (lane >> 5)[:, None]
    e = torch.arange(8)[None, :]
    slabs = torch.zeros(4, LAP_BLOCKS, 4, 64, 8)
    for q in range(4):
        W = torch.zeros(32 * LAP_BLOCKS, 64)
        W[:n] = heads_w[q]
        for b in range(LAP_BLOCKS):
            for st in range(4):
                slabs[q, b, st] = W[32 * b + row, 32 * (st >> 1) + _mf16_unit(st & 1, g, e)]
    hi, lo = _split_f16(slabs)
    frag = torch.stack([hi, lo], dim=3).contiguous()       # [q][b][s][hi|lo][lane][8]
    head = frag.view(torch.int16).reshape(-1).view(torch.float32)
    assert head.numel() == LAP_BIAS_OFF
    tail = pack_laplace_heads(ws_density, ws_rgb)[LAP_BIAS_OFF:].clone().view(4, LAP_BLOCKS, 2, 16)
    if exp2_rows:   # padded rows: bias -1e30 in front of exp2 (-> 0); in front of 1 / (1 + exp2(.)) the sign flips with the rows
        tail[1:][tail[1:] == LAP_PAD_BIAS] = -LAP_PAD_BIAS
    return torch.cat([head, tail.reshape(-1)])


_LAP_MAPS: Dict[str, Tuple[torch.Tensor, ...]] = {}


def _lap_maps(device):
    """Row / column index maps of the two Laplace blob layouts (the loops of pack_laplace_heads / pack_laplace_heads16
    as gather indices), built once per device."""
    key = str(device)
    if key not in _LAP_MAPS:
        lane = torch.arange(64)
        i, h = lane & 31, lane >> 5
        b, bi, r = torch.arange(LAP_BLOCKS), torch.arange(2), torch.arange(16)
        rows32 = ((32 * b)[:, None, None, None] + i[None, None, None, :]).expand(LAP_BLOCKS, 2, 16, 64)
        cols32 = ((32 * bi)[None, :, None, None] + _mfma_unit(r[None, None, :, None], h[None, None, None, :])).expand(LAP_BLOCKS, 2, 16, 64)
        brow = (32 * b)[:, None, None] + _mfma_unit(r[None, None, :], bi[None, :, None])                    # [B,2,16]
        st, e = torch.arange(4), torch.arange(8)
        rows16 = ((32 * b)[:, None, None, None] + i[None, None, :, None]).expand(LAP_BLOCKS, 4, 64, 8)
        cols16 = (32 * (st >> 1)[None, :, None, None]
                  + _mf16_unit((st & 1)[None, :, None, None], h[None, None, :, None], e[None, None, None, :])).expand(LAP_BLOCKS, 4, 64, 8)
        _LAP_MAPS[key] = tuple(t.contiguous().to(device) for t in (rows32, cols32, brow, rows16, cols16))
    return _LAP_MAPS[key]


def pack_laplace_sets(ws_density: torch.Tensor, ws_rgb: torch.Tensor, device, exp2_rows: bool = False,
                      softplus: bool = False) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """T sets of sampled last layers at once, packed ON `device` (one gather per blob; the per-chunk draws of
    NerfactoLaplaceModel make 64 sets per 1080p frame): ws_density [T,n,65], ws_rgb [T,n,195] ->
    (lap_blob [T, LAP_BLOB_FLOATS], lap16_blob [T, LAP_BLOB_FLOATS] | None).  Row t equals
    pack_laplace_heads(ws_density[t], ws_rgb[t]) / pack_laplace_heads16(..., exp2_rows, softplus) bit for bit
    (tests/test_mfma_pack_cpu.py); None when an operand is outside the f16 range."""
    wd = ws_density.detach().to(device=device, dtype=torch.float32)
    wr = ws_rgb.detach().to(device=device, dtype=torch.float32)
    T, n, nr = wd.shape[0], wd.shape[1], wr.shape[1]      # the colour head may carry its own number of rows
    assert wd.shape == (T, n, 65) and wr.shape == (T, nr, 195) and 1 <= n <= 32 * LAP_BLOCKS and 1 <= nr <= 32 * LAP_BLOCKS
    rows32, cols32, brow, rows16, cols16 = _lap_maps(wd.device)
    rows = 32 * LAP_BLOCKS

    def stack(d, c, pad_d, pad_c):
        """-> W [T,4,rows,64] (padding rows 0), B [T,4,rows] (padding rows pad_*)"""
        W = torch.zeros(T, 4, rows, 64, device=wd.device)
        B = torch.empty(T, 4, rows, device=wd.device)
        B[:, 0], B[:, 1:] = pad_d, pad_c
        W[:, 0, :n], B[:, 0, :n] = d[:, :, :64], d[:, :, 64]
        for ch in range(3):
            W[:, 1 + ch, :nr], B[:, 1 + ch, :nr] = c[:, :, ch * 64:(ch + 1) * 64], c[:, :, 192 + ch]
        return W, B

    W, B = stack(wd, wr, LAP_PAD_BIAS, LAP_PAD_BIAS)
    blob = torch.cat([W[:, :, rows32, cols32].reshape(T, -1), B[:, :, brow].reshape(T, -1)], dim=1)
    assert blob.shape[1] == LAP_BLOB_FLOATS
    if exp2_rows:   # the base change of the activation behind each row (pack_laplace_heads16)
        wd = (wd.double() * (1.0 if softplus else LOG2E)).to(torch.float32)
        wr = (wr.double() * -LOG2E).to(torch.float32)
    if max(float(wd.abs().max()), float(wr.abs().max())) >= F16_OPERAND_LIMIT:
        return blob, None
    W, B = stack(wd, wr, LAP_PAD_BIAS, -LAP_PAD_BIAS if exp2_rows else LAP_PAD_BIAS)
    hi, lo = _split_f16(W[:, :, rows16, cols16])                       # [T,4,B,4,64,8]
    frag = torch.stack([hi, lo], dim=4).contiguous()                   # [T][q][b][s][hi|lo][lane][8]
    head = frag.view(torch.int16).reshape(T, -1).view(torch.float32)
    assert head.shape[1] == LAP_BIAS_OFF
    return blob, torch.cat([head, B[:, :, brow].reshape(T, -1)], dim=1).contiguous()


# ------------------------------------------------------- frame-path scratch -------------

class Workspace:
    """Scratch buffers of the frame path (render.render_rays / render_camera): the per-launch-group temporaries that never
    leave it -- proposal densities, resampled bins, the field kernel's per-sample rows (6.4 GB per 2^20-ray group at K = 8).
    Taken from torch's caching allocator per call they are the same requests every group, but the allocator's block
    splitting needs several frames to settle: the FOURTH frame of a process asked hipMalloc for another 6 GiB
    (memory_reserved 7.9 -> 13.9 GiB), which costs 0.4 ms on some boxes and 118 ms on others -- one 163 ms frame in the
    timed five of a default bench run.  One flat buffer per tag, grown when a larger request arrives, handed out as views:
    every use is ordered on the stream that runs the frame, and nothing that is returned to a caller aliases one.
    UNERF_WORKSPACE=0: allocate per call instead (NerfSceneDev.workspace is then None)."""

    def __init__(self):
        self._bufs: Dict[Tuple, torch.Tensor] = {}
        self._streams: Dict[int, "torch.cuda.Stream"] = {}   # every stream a view was handed out on (see release)

    def get(self, tag: str, shape, device, dtype=torch.float32) -> torch.Tensor:
        n = 1
        for d in shape:
            n *= int(d)
        if torch.device(device).type == "cuda":
            st = torch.cuda.current_stream(device)
            self._streams.setdefault(st.cuda_stream, st)
        key = (tag, dtype, str(device))
        buf = self._bufs.get(key)
        if buf is None or buf.numel() < n:
            self._bufs.pop(key, None)          # release the smaller buffer before asking for the larger one
            buf = None
            buf = torch.empty(max(n, 1), device=device, dtype=dtype)
            self._bufs[key] = buf
        return buf[:n].view(*shape)

    def nbytes(self) -> int:
        return sum(b.numel() * b.element_size() for b in self._bufs.values())

    def release(self) -> None:
        """drop every buffer (scene teardown; a scene that is rebuilt keeps its arena instead: models.invalidate()).
        Each buffer goes back to the allocator with every stream recorded on it that a view of this arena was ever handed
        out on (`get` notes the caller's current stream: the frame's stream, and the sampling / shading side streams of
        render_camera(overlap=True) when they take scratch), so the block is not reused under work still queued there."""
        for b in self._bufs.values():
            if b.is_cuda:
                for st in self._streams.values():
                    if st.device == b.device:
                        b.record_stream(st)
        self._bufs.clear()
        self._streams.clear()


def _scratch(workspace: Optional["Workspace"], tag: str, shape, device) -> torch.Tensor:
    if workspace is None:
        return torch.empty(*shape, device=device, dtype=torch.float32)
    return workspace.get(tag, shape, device)


# ------------------------------------------------------- proposal sampling -------------

def proposal_density(origins, directions, sbins, net: DensityNetDev, near: float, far: float,
                     average_init_density: float, n: Optional[int] = None, ray_offset: int = 0,
                     image_width: int = 0, spacing: int = 0, workspace: Optional[Workspace] = None) -> torch.Tensor:
    """sbins: [n+1] shared row or [R,n+1] per ray -> density [R,n].  image_width > 0: the rays are pixels
    [ray_offset, ray_offset + R) of a row-major image (8x8-pixel-patch schedule, same results).
    workspace: the result is a view of that scratch arena (valid until the next call with it)."""
    lib = _l.load()
    R = origins.shape[0]
    stride = 0 if sbins.dim() == 1 else sbins.shape[1]
    n = sbins.shape[-1] - 1 if n is None else n
    out = _scratch(workspace, f"prop_density_{n}", (R, n), origins.device)
    cs = net.cstruct()
    with _ctx(origins.device):
        _run(f"proposal_density_{n}", lambda: lib.unerf_proposal_density(_p(origins), _p(directions), _p(sbins), stride, R, n, near, far,
                                            spacing, C.byref(cs), average_init_density, _p(out), ray_offset, image_width,
                                            _stream()))
    return out


def weights_pdf_resample(density, sbins, u, near: float, far: float, histogram_padding: float = 0.01,
                         eps: float = 1e-5, want_prop_depth: bool = True, want_weights: bool = False,
                         clip_minmax: Optional[torch.Tensor] = None, ray_offset: int = 0, chunk_rays: int = 1 << 15,
                         spacing: int = 0, workspace: Optional[Workspace] = None):
    """-> (new sbins [R,m+1], prop_depth [R,1] | None, weights [R,n] | None); workspace: the new bins are a view of it"""
    lib = _l.load()
    R, n = density.shape
    m = u.numel() - 1
    stride = 0 if sbins.dim() == 1 else sbins.shape[1]
    out = _scratch(workspace, f"pdf_bins_{m}", (R, m + 1), density.device)
    pd = torch.empty(R, 1, device=density.device, dtype=torch.float32) if want_prop_depth else None
    w = torch.empty(R, n, device=density.device, dtype=torch.float32) if want_weights else None
    with _ctx(density.device):
        _run(f"weights_pdf_resample_{n}", lambda: lib.unerf_weights_pdf_resample(_p(density), _p(sbins), stride, R, n, near, far, spacing, _p(u), m,
                                                histogram_padding, eps, _p(out), _p(pd), _p(w), _p(clip_minmax),
                                                ray_offset, chunk_rays, _stream()))
    return out, pd, w


def new_clip_buffer(num_rays: int, chunk_rays: int, device) -> torch.Tensor:
    nchunks = (num_rays + chunk_rays - 1) // chunk_rays
    buf = torch.empty(nchunks, 2, device=device, dtype=torch.float32)
    buf[:, 0] = float("inf")
    buf[:, 1] = 0.0
    return buf


# ------------------------------------------------------------ main field ---------------

def field_gather(origins, directions, sbins, field: FieldDev, near: float, far: float,
                 out: Optional[torch.Tensor] = None, spacing: int = 0) -> torch.Tensor:
    """level-major hash-grid lookup of the final samples -> feature planes [L, R*S, 2]"""
    lib = _l.load()
    R, S = sbins.shape[0], sbins.shape[1] - 1
    L = field.scalings.numel()
    if out is None:
        out = torch.empty(L, R * S, 2, device=origins.device, dtype=torch.float32)
    with _ctx(origins.device):
        _run("field_gather", lambda: lib.unerf_field_gather(_p(origins), _p(directions), _p(sbins), R, S, near, far,
                                                            spacing, _p(field.table), _p(field.scalings), L, field.log2T,
                                                            _p(out), _stream()))
    return out


def supports_planes(field: FieldDev) -> bool:
    """the sample-major plane layout is written by the ACTIVE / MCDROPOUT matrix kernels"""
    return field.mode != _l.FIELD_LAPLACE and field.use_mfma and (field.mfma_blob is not None or field.mfma16_blob is not None)


def supports_packed(field: "FieldDev") -> bool:
    """unerf_field_params.packed_out: the ACTIVE / MCDROPOUT kernels can leave one 16-byte row per sample"""
    return field.mode in (_l.FIELD_ACTIVE, _l.FIELD_MCDROPOUT) and not field.any_width


def field_fwd(origins, directions, sbins, field: FieldDev, near: float, far: float, ray_offset: int = 0,
              features: Optional[torch.Tensor] = None, image_width: int = 0, euclidean_bins: bool = False,
              sample_major: bool = False, spacing: int = 0, nonfinite_flag: Optional[torch.Tensor] = None,
              packed: bool = False, workspace: Optional[Workspace] = None):
    """-> density [B,R,S], rgb [B,R,S,3], aux, aux2 (see include/unerf.h).  image_width > 0 tells the kernel that
    rays [ray_offset, ray_offset+R) are consecutive pixels of a row-major image (8x4-pixel tiles: same results).
    euclidean_bins: `sbins` holds Euclidean bin edges (a caller-made RaySamples) instead of spacing-domain bins.
    sample_major: the outputs are planes density [B,S,R], rgb [B,S,3,R], aux [S,R] (composite_*_planes read them).
    packed (ACTIVE / MCDROPOUT, ray-major): -> None, rows [B,R,S,4] = (sigma, r, g, b), aux, None -- one 16-byte store
    per sample; composite_var / composite_moments take the rows as `rgb` with density=None.
    workspace: the four outputs are views of that scratch arena (valid until the next call with it)."""
    if euclidean_bins:
        near = -1.0
    lib = _l.load()
    R, S = sbins.shape[0], sbins.shape[1] - 1
    B = max(field.K, 1) if field.mode == _l.FIELD_MCDROPOUT else 1
    dev = origins.device
    if packed and (sample_major or not supports_packed(field)):
        raise _l.UnerfError("field_fwd: packed rows are written by the ACTIVE / MCDROPOUT kernels in the ray-major layout only")
    new = lambda tag, *shape: _scratch(workspace, "field_" + tag, shape, dev)
    has_aux = field.mode != _l.FIELD_MCDROPOUT
    if sample_major:
        density, rgb, aux = new("density", B, S, R), new("rgb", B, S, 3, R), (new("aux", S, R) if has_aux else None)
    elif packed:
        density, rgb, aux = None, new("rgb", B, R, S, 4), (new("aux", R, S) if has_aux else None)
    else:
        density, rgb, aux = new("density", B, R, S), new("rgb", B, R, S, 3), (new("aux", R, S) if has_aux else None)
    aux2 = new("aux2", R, S) if field.mode == _l.FIELD_LAPLACE else None
    cs = field.cstruct()
    cs.image_width = int(image_width)
    cs.sample_major = 1 if sample_major else 0
    cs.packed_out = 1 if packed else 0
    cs.overflow_flag = _p(nonfinite_flag, torch.int32)      # set by the f16 matrix kernels (either form) on operand overflow
    with _ctx(dev):
        _run("field_fwd", lambda: lib.unerf_field_fwd(_p(origins), _p(directions), _p(sbins), R, S, near, far, spacing, ray_offset,
                                     C.byref(cs), _p(features), _p(density), _p(rgb), _p(aux), _p(aux2), _stream()))
    return density, rgb, aux, aux2


def laplace_depth_weights(density_mu, density_var, sbins, near: float, far: float, noise: Optional[torch.Tensor],
                          D: int = 100, seed: int = 0, ray_offset: int = 0, spacing: int = 0) -> torch.Tensor:
    lib = _l.load()
    R, S = density_mu.shape
    out = torch.empty(R, S, device=density_mu.device, dtype=torch.float32)
    with _ctx(out.device):
        _run("laplace_depth_weights", lambda: lib.unerf_laplace_depth_weights(_p(density_mu), _p(density_var), _p(sbins), R, S, near, far,
                                                 spacing, _p(noise), D, seed & 0xFFFFFFFF, ray_offset, _p(out), _stream()))
    return out


def laplace_ggn_diag(origins, directions, sbins, field: FieldDev, density_mean: torch.Tensor, rgb_mean: torch.Tensor,
                     near: float, far: float, ggn_density: torch.Tensor, ggn_rgb: torch.Tensor, spacing: int = 0,
                     background=None) -> None:
    """One batch of NerfactoLaplaceModel.compute_hessian_naive (laplace_model.py:343-400): adds the batch's
    diagonal GGN of the summed-MSE loss to ggn_density [65] / ggn_rgb [195] in place.  density_mean [65] and
    rgb_mean [195] are the flattened mean last layers (weight row-major, then bias)."""
    lib = _l.load()
    if field.mode != _l.FIELD_LAPLACE:
        raise _l.UnerfError("laplace_ggn_diag needs a LAPLACE field")
    R, S = sbins.shape[0], sbins.shape[1] - 1
    dev = origins.device
    cs = field.cstruct()
    dm = density_mean.detach().to(device=dev, dtype=torch.float32).contiguous()
    rm = rgb_mean.detach().to(device=dev, dtype=torch.float32).contiguous()
    if dm.numel() != 65 or rm.numel() != 195:
        raise _l.UnerfError("laplace_ggn_diag: density_mean must have 65 and rgb_mean 195 entries")
    cs.ws_density, cs.ws_rgb, cs.n_lap, cs.n_lap_rgb, cs.lap_chunk_rays, cs.lap_sets = _p(dm), _p(rm), 1, 1, 0, 1
    cs.mfma_blob = _p(field.mfma_blob)
    nbytes = lib.unerf_laplace_ggn_workspace_bytes(R, S)
    ws = torch.empty((nbytes + 3) // 4, device=dev, dtype=torch.float32)
    bg_mode, bg_rgb = _background(background)
    with _ctx(dev):
        _run("laplace_ggn_diag", lambda: lib.unerf_laplace_ggn_diag(_p(origins), _p(directions), _p(sbins), R, S, near, far,
                                                                   spacing, C.byref(cs), bg_mode, bg_rgb, _p(ws), nbytes, _p(ggn_density),
                                                                   _p(ggn_rgb), _stream()))


def background_of(name) -> Tuple[int, Optional[Tuple[float, float, float]]]:
    """config.background_color -> (UNERF_BG_* mode, constant colour | None).  [UPSTREAM nerfstudio 1.1.0 RGBRenderer:
    "last_sample" blends the last sample's colour, "random" returns the composited colour unblended at eval, "white" /
    "black" (utils.colors.COLORS_DICT) blend a constant; a 3-vector is a constant colour.]"""
    if name is None or (isinstance(name, str) and name == "last_sample"):
        return _l.BG_LAST_SAMPLE, None
    if isinstance(name, str):
        if name == "random":
            return _l.BG_NONE, None
        colors = {"white": (1.0, 1.0, 1.0), "black": (0.0, 0.0, 0.0), "red": (1.0, 0.0, 0.0), "green": (0.0, 1.0, 0.0),
                  "blue": (0.0, 0.0, 1.0)}
        if name not in colors:
            raise _l.UnerfError(f"background_color={name!r}: expected last_sample, random, white, black")
        return _l.BG_COLOR, colors[name]
    vals = tuple(float(v) for v in torch.as_tensor(name).reshape(-1))
    if len(vals) != 3:
        raise _l.UnerfError("background colour must have 3 components")
    return _l.BG_COLOR, vals


def _background(background):
    """(mode, rgb) as produced by background_of (None = last_sample) -> (int, host float[3] | None) for the C ABI"""
    if background is None:
        return _l.BG_LAST_SAMPLE, None
    mode, rgb = background
    return int(mode), (None if rgb is None else (C.c_float * 3)(*[float(v) for v in rgb]))


def _packed_shape(density, rgb):
    if density is not None:
        return density.shape
    if rgb.dim() != 4 or rgb.shape[-1] != 4:
        raise _l.UnerfError(f"composite: density=None needs packed rows [B,R,S,4], got {tuple(rgb.shape)}")
    return rgb.shape[:3]


def composite_var(density, rgb, sbins, near: float, far: float, beta=None, weights_alt=None, clip_minmax=None,
                  ray_offset: int = 0, chunk_rays: int = 1 << 15, spacing: int = 0, background=None,
                  nonfinite_flag: Optional[torch.Tensor] = None) -> torch.Tensor:
    """density [B,R,S] -> out [B,R,8] = rgb3, accumulation, depth, expected_depth, rgb_var, depth_var.
    density=None: `rgb` holds the packed rows [B,R,S,4] = (sigma, r, g, b) of field_fwd(packed=True).
    nonfinite_flag: int32 device tensor (1 element) that receives |= 1 when a NaN density / colour is read"""
    lib = _l.load()
    B, R, S = _packed_shape(density, rgb)
    out = torch.empty(B, R, 8, device=rgb.device, dtype=torch.float32)
    bg_mode, bg_rgb = _background(background)
    with _ctx(out.device):
        _run("composite_var", lambda: lib.unerf_composite_var(_p(density), _p(rgb), _p(beta), _p(weights_alt), _p(sbins), B, R, S, near,
                                         far, spacing, _p(clip_minmax), ray_offset, chunk_rays, bg_mode, bg_rgb,
                                         _p(nonfinite_flag, torch.int32), _p(out), _stream()))
    return out


def composite_moments(density, rgb, sbins, near: float, far: float, clip_minmax=None, ray_offset: int = 0,
                      chunk_rays: int = 1 << 15, spacing: int = 0, background=None, nonfinite_flag=None):
    """density [B<=16,R,S], rgb [B,R,S,3] -> (mean [R,8], var [R,8]) over the B passes (fused composite + moments);
    density=None: `rgb` holds the packed rows [B,R,S,4] of field_fwd(packed=True)"""
    lib = _l.load()
    B, R, S = _packed_shape(density, rgb)
    mean = torch.empty(R, 8, device=rgb.device, dtype=torch.float32)
    var = torch.empty(R, 8, device=rgb.device, dtype=torch.float32)
    bg_mode, bg_rgb = _background(background)
    with _ctx(mean.device):
        _run("composite_moments", lambda: lib.unerf_composite_moments(_p(density), _p(rgb), _p(sbins), B, R, S, near, far,
                                                                      spacing, _p(clip_minmax), ray_offset, chunk_rays,
                                                                      bg_mode, bg_rgb, _p(nonfinite_flag, torch.int32),
                                                                      _p(mean), _p(var), _stream()))
    return mean, var


def composite_var_planes(density, rgb, sbins, near: float, far: float, beta=None, clip_minmax=None,
                         ray_offset: int = 0, chunk_rays: int = 1 << 15, spacing: int = 0, background=None,
                         nonfinite_flag=None) -> torch.Tensor:
    """planes density [B,S,R], rgb [B,S,3,R], beta [S,R] -> out [B,R,8] (channels as composite_var)"""
    lib = _l.load()
    B, S, R = density.shape
    out = torch.empty(B, R, 8, device=density.device, dtype=torch.float32)
    bg_mode, bg_rgb = _background(background)
    with _ctx(out.device):
        _run("composite_var", lambda: lib.unerf_composite_var_planes(_p(density), _p(rgb), _p(beta), _p(sbins), B, R, S, near, far,
                                                                    spacing, _p(clip_minmax), ray_offset, chunk_rays, bg_mode,
                                                                    bg_rgb, _p(nonfinite_flag, torch.int32), _p(out), _stream()))
    return out


def composite_moments_planes(density, rgb, sbins, near: float, far: float, clip_minmax=None, ray_offset: int = 0,
                             chunk_rays: int = 1 << 15, spacing: int = 0, background=None, nonfinite_flag=None):
    """planes density [B>=2,S,R], rgb [B,S,3,R] -> (mean [R,8], var [R,8]) over the B passes"""
    lib = _l.load()
    B, S, R = density.shape
    mean = torch.empty(R, 8, device=density.device, dtype=torch.float32)
    var = torch.empty(R, 8, device=density.device, dtype=torch.float32)
    bg_mode, bg_rgb = _background(background)
    with _ctx(mean.device):
        _run("composite_moments", lambda: lib.unerf_composite_moments_planes(_p(density), _p(rgb), _p(sbins), B, R, S, near, far,
                                                                             spacing, _p(clip_minmax), ray_offset, chunk_rays,
                                                                             bg_mode, bg_rgb, _p(nonfinite_flag, torch.int32),
                                                                             _p(mean), _p(var), _stream()))
    return mean, var


def moments(x: torch.Tensor, want_var: bool = True):
    """x [K,N,C] -> mean [N,C], var [N,C] (unbiased) | None"""
    lib = _l.load()
    K, N, Cc = x.shape
    mean = torch.empty(N, Cc, device=x.device, dtype=torch.float32)
    var = torch.empty(N, Cc, device=x.device, dtype=torch.float32) if want_var else None
    with _ctx(x.device):
        _run("moments", lambda: lib.unerf_moments(_p(x), K, N, Cc, _p(mean), _p(var), _stream()))
    return mean, var


# ---------------------------------------------------------------- splats ---------------

def splat_project(means3d, scales, glob_scale: float, quats, viewmat: torch.Tensor, fx, fy, cx, cy, H: int, W: int,
                  block_width: int = 16, clip_thresh: float = 0.01, raw: bool = False, opacity_logits=None,
                  antialiased: bool = False):
    """gsplat.project_gaussians signature -> (xys, depths, radii, conics, compensation, num_tiles_hit, cov3d).
    raw=True: `scales` are the model's log-scales and `quats` its unnormalised quaternions; torch.exp and the division by
    quats.norm() of activesplatfacto_model.py:221-223 happen inside the kernel (unerf_splat_project_raw).
    opacity_logits [N] (raw only): the tuple gains an 8th entry, opacities = sigmoid(logits) [* compensation if
    antialiased], and num_tiles_hit is the TIGHT count (tiles the alpha >= 1/255 ellipse reaches); bin it with
    splat_bin_sort(..., tight=(conics, opacities))."""
    lib = _l.load()
    if opacity_logits is not None and not raw:
        raise ValueError("opacity_logits (tight tile counts) go with raw=True")
    N, dev = means3d.shape[0], means3d.device
    xys = torch.empty(N, 2, device=dev)
    depths = torch.empty(N, device=dev)
    radii = torch.empty(N, device=dev, dtype=torch.int32)
    conics = torch.empty(N, 3, device=dev)
    comp = torch.empty(N, device=dev)
    tiles = torch.empty(N, device=dev, dtype=torch.int32)
    cov3d = torch.empty(N, 6, device=dev)
    opac = torch.empty(N, device=dev) if opacity_logits is not None else None
    outs = lambda: (_p(xys), _p(depths), _p(radii, torch.int32), _p(conics), _p(comp), _p(tiles, torch.int32), _p(cov3d),
                    _stream())
    with _ctx(dev):
        if raw:
            _run("splat_project", lambda: lib.unerf_splat_project_raw(
                _p(means3d), _p(scales), glob_scale, _p(quats), _host12(viewmat), fx, fy, cx, cy, H, W, block_width,
                clip_thresh, N, _p(opacity_logits), 1 if antialiased else 0, _p(opac), *outs()))
        else:
            _run("splat_project", lambda: lib.unerf_splat_project(
                _p(means3d), _p(scales), glob_scale, _p(quats), _host12(viewmat), fx, fy, cx, cy, H, W, block_width,
                clip_thresh, N, *outs()))
    if opac is not None:
        return xys, depths, radii, conics, comp, tiles, cov3d, opac
    return xys, depths, radii, conics, comp, tiles, cov3d


def splat_sh_colors(degree: int, means3d, cam_pos: torch.Tensor, sh_coeffs, log_unc=None, beta_min: float = 0.01):
    lib = _l.load()
    N, dev = means3d.shape[0], means3d.device
    colors = torch.empty(N, 3, device=dev)
    beta = torch.empty(N, device=dev) if log_unc is not None else None
    cp = (C.c_float * 3)(*[float(v) for v in cam_pos.detach().cpu().reshape(-1)[:3]])
    with _ctx(dev):
        _run("splat_sh_colors", lambda: lib.unerf_splat_sh_colors(degree, _p(means3d), cp, _p(sh_coeffs), _p(log_unc), beta_min, N,
                                           _p(colors), _p(beta), _stream()))
    return colors, beta


def splat_sh_colors_split(degree: int, means3d, cam_pos: torch.Tensor, features_dc, features_rest, log_unc=None,
                          beta_min: float = 0.01):
    """splat_sh_colors on gauss_params.features_dc [N,3] / features_rest [N,15,3] as stored (no concatenation)."""
    lib = _l.load()
    N, dev = means3d.shape[0], means3d.device
    colors = torch.empty(N, 3, device=dev)
    beta = torch.empty(N, device=dev) if log_unc is not None else None
    cp = (C.c_float * 3)(*[float(v) for v in cam_pos.detach().cpu().reshape(-1)[:3]])
    with _ctx(dev):
        _run("splat_sh_colors", lambda: lib.unerf_splat_sh_colors_split(
            degree, _p(means3d), cp, _p(features_dc), _p(features_rest), _p(log_unc), beta_min, N, _p(colors), _p(beta),
            _stream()))
    return colors, beta


def splat_shade_inputs(degree: int, means3d, cam_pos: torch.Tensor, features_dc, features_rest, log_unc, beta_min: float,
                       opacity_logits, compensation, depths):
    """-> (rows [N,C], opacities [N] | None): rows = [rgb, beta, depth] (C = 5) with `log_unc`, [rgb, depth] (C = 4)
    without; opacities = sigmoid(opacity_logits) [* compensation] (None when opacity_logits is None: the projection
    made them).  One launch for the SH colours, beta, the channel concatenation and the opacity activation of one frame
    (unerf_splat_shade_inputs)."""
    lib = _l.load()
    N, dev = means3d.shape[0], means3d.device
    Cn = 5 if log_unc is not None else 4
    rows = torch.empty(N, Cn, device=dev)
    opac = torch.empty(N, device=dev) if opacity_logits is not None else None
    cp = (C.c_float * 3)(*[float(v) for v in cam_pos.detach().cpu().reshape(-1)[:3]])
    with _ctx(dev):
        _run("splat_sh_colors", lambda: lib.unerf_splat_shade_inputs(
            degree, _p(means3d), cp, _p(features_dc), _p(features_rest), _p(log_unc), beta_min, _p(opacity_logits),
            _p(compensation), _p(depths), N, Cn, _p(rows), _p(opac), _stream()))
    return rows, opac


class SplatCount:
    """The asynchronous half of gsplat's compute_cumulative_intersects: the inclusive scan of num_tiles_hit is launched
    on the caller's stream, and its last element (the number of intersections, which sizes every later buffer) travels to
    pinned host memory on a SIDE stream.  `wait()` blocks on that copy only -- kernels the caller queued on its own stream
    in between (SH colours: independent of the count) keep the GPU busy while the host learns the number, instead of the
    device idling through a full stream synchronisation and the launch latency of everything behind it."""

    _side: Dict = {}
    # A small RING of pinned int32 words per device: a count may be started and awaited later (pipelined frames, ensemble
    # members, threads), so two can be in flight; each takes the next word, and wait() caches its value on first return --
    # a word is reused only RING counts later (a count still unawaited by then is refused: it would read another's value)
    RING = 8
    _pinned: Dict = {}
    _next: Dict = {}
    _owner: Dict = {}

    _events: Dict = {}
    # slot allocation is the one piece of state threads share: two threads counting on one device would otherwise take the
    # same ring word (and its event pair) and each could read the other's count
    _lock = threading.Lock()

    def __init__(self, num_tiles_hit: torch.Tensor, defer_copy: bool = False):
        """defer_copy: only the scan is queued (and the point behind it marked on the caller's stream); the caller queues its
        independent kernels next and then calls start_copy() -- the host-side set-up of the side-stream copy (~50 us of
        Python and HIP calls) then runs while those kernels execute instead of in front of them."""
        lib = _l.load()
        self.N, self.dev = num_tiles_hit.shape[0], num_tiles_hit.device
        self.cum = torch.empty(self.N, device=self.dev, dtype=torch.int32)
        self._value: Optional[int] = None
        self._copy_started = False
        with _ctx(self.dev):
            ws0 = torch.empty(int(lib.unerf_splat_sort_workspace_bytes(self.N, 0)), device=self.dev, dtype=torch.uint8)
            _run("splat_count", lambda: lib.unerf_splat_count_intersects(_p(num_tiles_hit, torch.int32), self.N,
                                                                         _p(self.cum, torch.int32), _p(ws0, torch.uint8),
                                                                         ws0.numel(), _stream()))
            key = self.dev.index if self.dev.index is not None else torch.cuda.current_device()
            self._key = key
            import weakref
            with SplatCount._lock:
                if key not in SplatCount._side:
                    SplatCount._side[key] = torch.cuda.Stream(device=self.dev)
                if key not in SplatCount._pinned:
                    SplatCount._pinned[key] = torch.empty(SplatCount.RING, dtype=torch.int32, pin_memory=True)
                    SplatCount._next[key], SplatCount._owner[key] = 0, [None] * SplatCount.RING
                    # one (scan done, copy done) event pair per ring word, made once: an event is reusable once awaited
                    SplatCount._events[key] = [(torch.cuda.Event(), torch.cuda.Event()) for _ in range(SplatCount.RING)]
                slot = SplatCount._next[key]
                prev = SplatCount._owner[key][slot]
                prev = prev() if prev is not None else None
                if prev is not None and prev._value is None:
                    raise _l.UnerfError(f"SplatCount: {SplatCount.RING} counts started on this device without wait(): await "
                                        "them before starting more")
                SplatCount._owner[key][slot] = weakref.ref(self)
                SplatCount._next[key] = (slot + 1) % SplatCount.RING
            self._host = SplatCount._pinned[key][slot:slot + 1]
            self._ready, self._done = SplatCount._events[key][slot]
            self._ready.record(torch.cuda.current_stream())
            self._ws0 = ws0     # keeps the scan's scratch alive until the count is known
        if not defer_copy:
            self.start_copy()

    def start_copy(self) -> None:
        """queue the read-back of the count on the side stream (behind the scan only)"""
        if self._copy_started:
            return
        self._copy_started = True
        side = SplatCount._side[self._key]
        with _ctx(self.dev):
            with torch.cuda.stream(side):
                side.wait_event(self._ready)
                self._host.copy_(self.cum[-1:], non_blocking=True)
                self._done.record(side)
            self.cum.record_stream(side)

    def wait(self) -> int:
        if self._value is None:
            self.start_copy()
            self._done.synchronize()
            self._value = int(self._host[0])
        return self._value


def splat_bin_sort(xys, depths, radii, num_tiles_hit, H: int, W: int, block_width: int = 16,
                   want_isect_ids: bool = True, count: Optional[SplatCount] = None,
                   tight: Optional[Tuple[torch.Tensor, torch.Tensor]] = None):
    """-> (num_intersects, cum_tiles_hit, isect_ids_sorted | None, gaussian_ids_sorted, tile_bins [tiles,2])
    One host read-back (num_intersects) sizes the buffers, as gsplat's compute_cumulative_intersects does; `count`: a
    SplatCount started earlier (so that the read-back overlaps other work), else it is made and awaited here.
    The 64-bit isect ids are gsplat's by-product; the rasteriser does not read them (want_isect_ids=False
    skips their gather + 8-byte store per intersection).
    tight = (conics, opacities): num_tiles_hit is a tight count of splat_project(..., opacity_logits=...)."""
    lib = _l.load()
    N, dev = xys.shape[0], xys.device
    tbx, tby = (W + block_width - 1) // block_width, (H + block_width - 1) // block_width
    count = SplatCount(num_tiles_hit) if count is None else count
    cum = count.cum
    I = count.wait()
    with _ctx(dev):
        ws = torch.empty(int(lib.unerf_splat_sort_workspace_bytes(N, I)), device=dev, dtype=torch.uint8)
        ids = torch.empty(max(I, 1), device=dev, dtype=torch.int64) if want_isect_ids else None
        gids = torch.empty(max(I, 1), device=dev, dtype=torch.int32)
        bins = torch.empty(tbx * tby, 2, device=dev, dtype=torch.int32)
        _run("splat_bin_sort", lambda: lib.unerf_splat_bin_sort(_p(xys), _p(depths), _p(radii, torch.int32), _p(cum, torch.int32), N, I, H,
                                          W, block_width, _p(tight[0]) if tight else None, _p(tight[1]) if tight else None,
                                          _p(ids, torch.int64), _p(gids, torch.int32),
                                          _p(bins, torch.int32), _p(ws, torch.uint8), ws.numel(), _stream()))
    return I, cum, (ids[:I] if ids is not None else None), gids[:I], bins


def splat_rasterize(gaussian_ids_sorted, tile_bins, xys, conics, colors, opacities, H: int, W: int,
                    background: Optional[torch.Tensor] = None, block_width: int = 16, want_final_idx: bool = False,
                    stop_idx: Optional[torch.Tensor] = None, cull: bool = True,
                    chan_max: Optional[Tuple[int, torch.Tensor]] = None):
    """colors [N,C] -> (out_img [H,W,C], final_T [H,W], final_idx | None).  stop_idx [H,W] int32: the final_idx of an
    earlier pass with the same geometry (bounded second pass); cull=False: gsplat's schedule without wave-level culling
    (same bits either way).  chan_max = (channel, one zeroed device float): receives max(out_img[..., channel]) for
    splat_alpha_normalize(..., max_ready=that float)."""
    lib = _l.load()
    dev, Cn = xys.device, colors.shape[1]
    out = torch.empty(H, W, Cn, device=dev)
    fT = torch.empty(H, W, device=dev)
    fidx = torch.empty(H, W, device=dev, dtype=torch.int32) if want_final_idx else None
    if gaussian_ids_sorted.numel() == 0:
        gaussian_ids_sorted = torch.zeros(1, device=dev, dtype=torch.int32)
    with _ctx(dev):
        _run(f"splat_rasterize_c{Cn}", lambda: lib.unerf_splat_rasterize(_p(gaussian_ids_sorted, torch.int32), _p(tile_bins, torch.int32), _p(xys),
                                           _p(conics), _p(colors), _p(opacities), _p(background), Cn, H, W,
                                           block_width, _p(stop_idx, torch.int32), 0 if cull else _l.RASTER_NO_CULL,
                                           chan_max[0] if chan_max else -1, _p(chan_max[1]) if chan_max else None,
                                           _p(out), _p(fT), _p(fidx, torch.int32), _stream()))
    return out, fT, fidx


def splat_alpha_normalize(img: torch.Tensor, ch: int, final_T: torch.Tensor, max_ready: Optional[torch.Tensor] = None) -> None:
    """in place on channel `ch` of img [H,W,C].  max_ready: the device float a splat_rasterize(chan_max=(ch, float)) call
    that produced img left the channel's maximum in; None: the maximum is computed here."""
    lib = _l.load()
    H, W, Cn = img.shape
    scratch = torch.empty(1, device=img.device) if max_ready is None else max_ready
    with _ctx(img.device):
        _run("splat_alpha_normalize", lambda: lib.unerf_splat_alpha_normalize(_p(img), Cn, ch, _p(final_T), H * W, _p(scratch),
                                                                              0 if max_ready is None else 1, _stream()))


def splat_normalize_outputs(img: torch.Tensor, ch: int, final_T: torch.Tensor, max_ready: torch.Tensor, rgb: bool = False,
                            acc: bool = False, sq_ch: Optional[int] = None, sqrt: bool = False):
    """splat_alpha_normalize(img, ch, final_T, max_ready) + the frame's elementwise outputs in the same pass:
    -> (rgb [H,W,3] = clamp(img[..., :3], max=1) | None, accumulation [H,W,1] = 1 - final_T | None,
        img[..., sq_ch] ** 2 [H,W,1] | None, sqrt(normalised channel) [H,W,1] | None) -- the bits of the torch calls they replace"""
    lib = _l.load()
    H, W, Cn = img.shape
    dev = img.device
    rgb_o = torch.empty(H, W, 3, device=dev) if rgb else None
    acc_o = torch.empty(H, W, 1, device=dev) if acc else None
    sq_o = torch.empty(H, W, 1, device=dev) if sq_ch is not None else None
    sqrt_o = torch.empty(H, W, 1, device=dev) if sqrt else None
    with _ctx(dev):
        _run("splat_alpha_normalize", lambda: lib.unerf_splat_normalize_outputs(
            _p(img), Cn, ch, _p(final_T), H * W, _p(max_ready), _p(rgb_o), _p(acc_o), -1 if sq_ch is None else sq_ch, _p(sq_o), _p(sqrt_o),
            _stream()))
    return rgb_o, acc_o, sq_o, sqrt_o


def splat_depth_sqdiff(xys, depths, depth_img: torch.Tensor, ch: int) -> torch.Tensor:
    lib = _l.load()
    H, W, Cn = depth_img.shape
    out = torch.empty(xys.shape[0], device=xys.device)
    with _ctx(xys.device):
        _run("splat_depth_sqdiff", lambda: lib.unerf_splat_depth_sqdiff(_p(xys), _p(depths), _p(depth_img), Cn, ch, H, W, xys.shape[0], _p(out),
                                              _stream()))
    return out
