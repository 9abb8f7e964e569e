"""Frame-level NeRF pipelines on the HIP kernels.

One call renders a batch of rays the way the reference renders one eval chunk
(`Model.get_outputs_for_camera_ray_bundle` -> `forward` -> `get_outputs`), but with a launch
granularity chosen for MI355X: `rays_per_launch` (default 2^20: a 1080p frame is two launch groups; the K = 8 sample
buffers of one group are 6.4 GB) rays go through the seven kernels at once -- 288 GB of HBM make the reference's
32768-ray chunking unnecessary (2^18 -> 2^20 is worth 1.5-2 % of a frame: fewer partially filled last rounds; measured
in profiles/r3_12_exp_rays_per_launch.json); the only
place the reference chunk size is observable (DepthRenderer("expected") clips to the chunk's
min/max sample position) is reproduced exactly through `chunk_rays`.

Kernel sequence per launch group:
  sampling stage : proposal_density(256) -> weights_pdf_resample(->96) -> proposal_density(96)
                   -> weights_pdf_resample(->48) -> field_gather (level-major hash-grid lookup)
  shading stage  : field_fwd (hash grid + MLPs on the f16 matrix pipe: "f16" / split-f16 "f16x2" / exact "fp32", see
                   ops.FieldDev.precision) -> [laplace_depth_weights] -> composite_var | composite_moments over K

By default the hash-grid lookup is fused into `field_fwd` and everything runs on the caller's
stream.  Two measured alternatives are kept as options (numbers: DESIGN.md section 4.3):
`scene.split_gather` (level-major gather kernel, one 4-MiB level table = one XCD L2 at a time) and
`render_camera(overlap=True)` (sampling stage of launch group g+1 on a second HIP stream underneath
the shading stage of group g).  On MI355X neither beats the fused single-stream form for these
shapes: the stages contend for the same CUs' issue slots and register file.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import lib as _l
from . import ops


def _linspace_bins(n: int) -> torch.Tensor:
    return torch.linspace(0.0, 1.0, n + 1)


def _pdf_u(m: int) -> torch.Tensor:
    nb = m + 1
    u = torch.linspace(0.0, 1.0 - (1.0 / nb), steps=nb)
    return u + 1.0 / (2 * nb)


@dataclass
class NerfSceneDev:
    """Device-resident nerfacto-family scene: main field + proposal networks + sampler constants
    (NerfactoModelConfig defaults)."""
    field: ops.FieldDev
    props: List[ops.DensityNetDev]
    near: float = 0.05
    far: float = 1000.0
    num_prop: Tuple[int, ...] = (256, 96)
    num_nerf: int = 48
    prop_average_init_density: float = 0.01
    chunk_rays: int = 1 << 15
    # proposal_initial_sampler: lib.SPACING_PIECEWISE (nerfacto default) | lib.SPACING_UNIFORM (README.md:153 of the reference)
    spacing: int = 0
    # RGBRenderer background as (UNERF_BG_* mode, colour | None) from ops.background_of(config.background_color);
    # None = "last_sample"
    background: Optional[Tuple] = None
    split_gather: bool = False  # True: level-major gather kernel + feature planes instead of the fused lookup
    # field outputs as sample-major planes + lane-per-ray composite (ACTIVE / MCDROPOUT).  Measured alternative, off by
    # default: the plane stores cut the field kernel's write traffic to the algorithmic bytes, but the lane-per-ray
    # composite is latency-bound and costs more than the stores gain; UNERF_SAMPLE_MAJOR=1 turns it on
    sample_major: bool = field(default_factory=lambda: os.environ.get("UNERF_SAMPLE_MAJOR", "0") == "1")
    # OverflowGuard: re-render launch groups whose f16 operands overflowed with fp32 kernels (UNERF_OVERFLOW_GUARD=0: off)
    overflow_guard: bool = field(default_factory=lambda: os.environ.get("UNERF_OVERFLOW_GUARD", "1") != "0")
    overflow_rerenders: int = 0      # how many launch groups that has happened to (diagnostic)
    # field outputs as packed (sigma, r, g, b) rows (include/unerf.h: packed_out)
    packed_out: bool = field(default_factory=lambda: os.environ.get("UNERF_PACKED_OUT", "1") != "0")
    # scratch arena of the frame path's per-launch-group temporaries (ops.Workspace; UNERF_WORKSPACE=0: per-call allocations)
    workspace: Optional[ops.Workspace] = field(
        default_factory=lambda: ops.Workspace() if os.environ.get("UNERF_WORKSPACE", "1") != "0" else None)
    _const: Dict[str, torch.Tensor] = field(default_factory=dict)

    @property
    def device(self):
        return self.field.table.device

    def const(self, name: str, n: int) -> torch.Tensor:
        key = f"{name}{n}"
        if key not in self._const:
            t = _linspace_bins(n) if name == "bins" else _pdf_u(n)
            self._const[key] = t.to(self.device)
        return self._const[key]


class OverflowGuard:
    """The f16 matrix kernels (precision "f16x2" / "f16") carry activations as f16 operands: a hidden unit or logit at or
    beyond 65504 becomes hi = inf, lo = -inf and poisons its sample with NaN -- which the renderers' nan_to_num would
    turn into a plausible pixel.  Trained nerfacto fields sit orders of magnitude below that, so nothing is clamped in the
    hot loops; instead every composite call ORs a "saw a NaN density / colour" bit into one device word per launch group
    (unerf_composite_*: nonfinite_flag), the words of a frame are read back ONCE at its end, and a group whose word is
    set is rendered again with the exact-fp32 kernels, which have no such limit.  A NaN the fp32 kernels produce too
    (inf density x selector 0, as in the reference) survives the re-render unchanged."""

    def __init__(self, scene: "NerfSceneDev", n_groups: int):
        f = getattr(scene, "field", None)
        on = (f is not None and f.use_mfma and f.precision != "fp32" and f.mfma16_blob is not None
              and getattr(scene, "overflow_guard", True))
        self.scene = scene
        self.flags = torch.zeros(max(n_groups, 1), dtype=torch.int32, device=scene.device) if on else None
        self.rerendered: List[int] = []

    def flag(self, g: int) -> Optional[torch.Tensor]:
        return None if self.flags is None else self.flags[g:g + 1]

    def offenders(self) -> List[int]:
        """launch groups that saw a NaN -- one device -> host read per frame"""
        if self.flags is None:
            return []
        return [i for i, f in enumerate(self.flags.tolist()) if f]     # one small device -> host copy (the frame's sync)

    def redo(self, g: int, render_group):
        """render_group() again with the exact-fp32 kernels"""
        f = self.scene.field
        saved, f.precision = f.precision, "fp32"
        try:
            out = render_group()
        finally:
            f.precision = saved
        self.rerendered.append(g)
        self.scene.overflow_rerenders += 1
        return out


def sample_rays(scene: NerfSceneDev, origins: torch.Tensor, directions: torch.Tensor, clip: Optional[torch.Tensor],
                ray_offset: int = 0, want_prop_depth: bool = True, image_width: int = 0,
                init_bins: Optional[torch.Tensor] = None, workspace: Optional[ops.Workspace] = None):
    """ProposalNetworkSampler at eval.  -> (final spacing bins [R,S+1], [prop_depth_0, prop_depth_1])
    workspace (the frame path only): the bins are a view of that scratch arena, not a tensor of their own
    init_bins [R, num_prop[0]+1]: per-ray first-level bins (a bundle with its own nears / fars: `crop_bins`), else
    the shared uniform row"""
    sb = scene.const("bins", scene.num_prop[0]) if init_bins is None else init_bins
    prop_depths = []
    n_iter = len(scene.props)
    for lvl in range(n_iter):
        dens = ops.proposal_density(origins, directions, sb, scene.props[lvl], scene.near, scene.far,
                                    scene.prop_average_init_density, ray_offset=ray_offset, image_width=image_width,
                                    spacing=scene.spacing, workspace=workspace)
        m = scene.num_prop[lvl + 1] if lvl + 1 < n_iter else scene.num_nerf
        last = lvl + 1 == n_iter
        sb, pd, _ = ops.weights_pdf_resample(dens, sb, scene.const("u", m), scene.near, scene.far,
                                             want_prop_depth=want_prop_depth,
                                             clip_minmax=clip if last else None, ray_offset=ray_offset,
                                             chunk_rays=scene.chunk_rays, spacing=scene.spacing, workspace=workspace)
        prop_depths.append(pd)
    return sb, prop_depths


def _unpack(out: torch.Tensor) -> Dict[str, torch.Tensor]:
    return {
        "rgb": out[:, 0:3], "accumulation": out[:, 3:4], "depth": out[:, 4:5], "expected_depth": out[:, 5:6],
        "rgb_var": out[:, 6:7], "depth_var": out[:, 7:8],
    }


def _uses_split(scene: NerfSceneDev) -> bool:
    f = scene.field
    return bool(scene.split_gather and f.mode != _l.FIELD_LAPLACE and f.use_mfma and f.mfma_blob is not None
                and f.tcnn_levels is None)


def crop_bins(scene: NerfSceneDev, origins, directions, obb=None, nears=None, fars=None) -> Optional[torch.Tensor]:
    """First-level bins for rays whose planes are not the collider's: `obb` = (world_to_box [3,4], S [3]) of an
    OrientedBox -- what Cameras.generate_rays(..., obb_box=box) does to the bundle (laplace_model.py:413,
    ensemble_pipeline.py:157) -- or the bundle's own nears / fars [R,1]."""
    row = scene.const("bins", scene.num_prop[0])
    if obb is not None:
        return ops.ray_box_bins(origins, directions, obb[0], obb[1], scene.near, scene.far, row, spacing=scene.spacing)[0]
    if nears is not None and fars is not None:
        return ops.ray_planes_bins(nears, fars, scene.near, scene.far, row, spacing=scene.spacing)
    return None


def sampling_stage(scene: NerfSceneDev, origins, directions, clip, ray_offset: int, image_width: int = 0,
                   init_bins: Optional[torch.Tensor] = None, scratch: bool = True):
    """-> (final spacing bins, prop depths, feature planes | None); the frame path's stage: its temporaries live in
    scene.workspace.  scratch=False (render_camera(overlap=True)): tensors of their own -- there the bins of group g are
    still being read by the shading stream while this stage runs for group g + 1"""
    sb, prop_depths = sample_rays(scene, origins, directions, clip, ray_offset, image_width=image_width,
                                  init_bins=init_bins, workspace=scene.workspace if scratch else None)
    feats = (ops.field_gather(origins, directions, sb, scene.field, scene.near, scene.far, spacing=scene.spacing)
             if _uses_split(scene) else None)
    return sb, prop_depths, feats


def shading_stage(scene: NerfSceneDev, origins, directions, sb, prop_depths, feats, clip, ray_offset: int = 0,
                  depth_noise: Optional[torch.Tensor] = None, depth_draws: int = 100, depth_seed: int = 0,
                  keep_density: bool = False, image_width: int = 0,
                  nonfinite_flag: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
    f = scene.field
    # ACTIVE / MCDROPOUT: the field kernel writes sample-major planes (whole 32-byte sectors per store) and the
    # composite walks them with a lane per ray; LAPLACE keeps the ray-major layout its depth-draw kernel reads
    planes = scene.sample_major and feats is None and ops.supports_planes(f)
    # default (ACTIVE / MCDROPOUT): one 16-byte row (sigma, r, g, b) per sample instead of four scattered dwords
    # (unerf_field_params.packed_out; UNERF_PACKED_OUT=0: the [B,R,S] + [B,R,S,3] layout of the Field-level API)
    packed = scene.packed_out and not planes and ops.supports_packed(f)
    density, rgb, aux, aux2 = ops.field_fwd(origins, directions, sb, f, scene.near, scene.far, ray_offset, features=feats,
                                            image_width=image_width, sample_major=planes, spacing=scene.spacing,
                                            nonfinite_flag=nonfinite_flag, packed=packed, workspace=scene.workspace)
    kw = dict(clip_minmax=clip, ray_offset=ray_offset, chunk_rays=scene.chunk_rays, spacing=scene.spacing,
              background=scene.background, nonfinite_flag=nonfinite_flag)
    res: Dict[str, torch.Tensor] = {}
    if f.mode == _l.FIELD_ACTIVE:
        if planes:
            out = ops.composite_var_planes(density, rgb, sb, scene.near, scene.far, beta=aux, **kw)[0]
        else:
            out = ops.composite_var(density, rgb, sb, scene.near, scene.far, beta=aux, **kw)[0]
        res = _unpack(out)
        res["rgb_std"] = res["rgb_var"].sqrt()
        res["depth_std"] = res["depth_var"].sqrt()
        if keep_density:   # the reference returns density [R,48,1] (activenerfacto_model.py:115,122)
            # (a copy: the kernel's rows live in scene.workspace and the next launch group overwrites them)
            res["density"] = (rgb[0][..., 0] if packed else (density[0].t() if planes else density[0])).clone(
                memory_format=torch.contiguous_format)
    elif f.mode == _l.FIELD_MCDROPOUT:
        if planes and f.K >= 2:
            mean, var = ops.composite_moments_planes(density, rgb, sb, scene.near, scene.far, **kw)
        elif planes and f.K == 1:
            mean = ops.composite_var_planes(density, rgb, sb, scene.near, scene.far, **kw)[0]
            var = torch.full_like(mean, float("nan"))       # torch.std of one pass (unbiased) is NaN
        elif 0 < f.K <= 16:
            mean, var = ops.composite_moments(density, rgb, sb, scene.near, scene.far, **kw)
        elif f.K > 16:
            out = ops.composite_var(density, rgb, sb, scene.near, scene.far, **kw)  # [B,R,8]
            mean, var = ops.moments(out[:, :, :6].contiguous())
        if f.K > 0:
            res = {"rgb": mean[:, 0:3], "accumulation": mean[:, 3:4], "depth": mean[:, 4:5],
                   "expected_depth": mean[:, 5:6]}
            std = var.sqrt()
            res["rgb_std"] = std[:, 0:3].mean(dim=-1)[..., None]
            res["depth_std"] = std[:, 4:5].mean(dim=-1)[..., None]
            res["expected_depth_std"] = std[:, 5:6].mean(dim=-1)[..., None]
        else:
            comp = ops.composite_var_planes if planes else ops.composite_var
            u = _unpack(comp(density, rgb, sb, scene.near, scene.far, **kw)[0])
            res = {k: u[k] for k in ("rgb", "accumulation", "depth", "expected_depth")}
    else:
        # use_deterministic_density (laplace_model.py:486-507 is skipped): depth from the ordinary weights
        walt = None if f.lap_mask_density else ops.laplace_depth_weights(
            density[0], aux, sb, scene.near, scene.far, depth_noise, depth_draws, depth_seed, ray_offset,
            spacing=scene.spacing)
        out = ops.composite_var(density, rgb, sb, scene.near, scene.far, beta=aux2, weights_alt=walt, **kw)[0]
        u = _unpack(out)
        res = {"rgb": u["rgb"], "rgb_std": u["rgb_var"].sqrt(), "accumulation": u["accumulation"],
               "depth": u["depth"], "depth_std": u["depth_var"].sqrt(), "expected_depth": u["expected_depth"]}
    for i, pd in enumerate(prop_depths):
        if pd is not None:
            res[f"prop_depth_{i}"] = pd
    return res


def render_rays(scene: NerfSceneDev, origins: torch.Tensor, directions: torch.Tensor, ray_offset: int = 0,
                total_rays: Optional[int] = None, clip: Optional[torch.Tensor] = None,
                init_bins: Optional[torch.Tensor] = None, **shade_kw) -> Dict[str, torch.Tensor]:
    """Render rays [R,3] with the scene's method (field.mode).  Output keys follow the reference:
      ACTIVE     activenerfacto_model.py:117-127   rgb accumulation depth expected_depth rgb_var rgb_std
                                                   depth_var depth_std prop_depth_i (+density)
      MCDROPOUT  mcdropout_models.py:121-126       means of every key + rgb_std depth_std expected_depth_std
      LAPLACE    laplace_model.py:523-530          rgb rgb_std accumulation depth depth_std expected_depth
    """
    _l.require_gpu()
    R = origins.shape[0]
    if clip is None:
        clip = ops.new_clip_buffer((total_rays or (ray_offset + R)), scene.chunk_rays, origins.device)
    sb, prop_depths, feats = sampling_stage(scene, origins, directions, clip, ray_offset,
                                            image_width=shade_kw.get("image_width", 0), init_bins=init_bins)
    return shading_stage(scene, origins, directions, sb, prop_depths, feats, clip, ray_offset, **shade_kw)


def render_camera(scene: NerfSceneDev, c2w: torch.Tensor, fx: float, fy: float, cx: float, cy: float, H: int, W: int,
                  rays_per_launch: int = 1 << 20, overlap: bool = False, obb=None, distortion=None, camera_type: int = 1,
                  **shade_kw) -> Dict[str, torch.Tensor]:
    """get_outputs_for_camera: generate the H*W rays on device, render them in row-major launch
    groups, return images [H,W,C].  With `overlap`, sampling (group g+1) and shading (group g) run on
    two streams.  obb = (world_to_box [3,4], S [3]): the oriented crop box of `obb_box` (see crop_bins).
    distortion = the camera's `distortion_params` (k1, k2, k3, k4, p1, p2) or None: the rays are bent as
    Cameras.generate_rays bends them (unerf_generate_rays).  camera_type: nerfstudio's CameraType value (perspective 1,
    fisheye 2, equirectangular 3, orthophoto 8: include/unerf.h)."""
    _l.require_gpu()
    total = H * W
    dev = scene.device
    clip = ops.new_clip_buffer(total, scene.chunk_rays, dev)
    # launch groups must not split a reference chunk (the clip bounds are per chunk)
    rpl = max(scene.chunk_rays, (rays_per_launch // scene.chunk_rays) * scene.chunk_rays)
    starts = list(range(0, total, rpl))
    lists: Dict[str, List[torch.Tensor]] = {}
    with torch.cuda.device(dev):
        cur = torch.cuda.current_stream()
        guard = OverflowGuard(scene, len(starts))

        def group(gi: int, flag=None):
            start = starts[gi]
            o, d, _ = ops.generate_rays(c2w, fx, fy, cx, cy, H, W, dev, start, min(rpl, total - start), distortion=distortion,
                                        camera_type=camera_type)
            return render_rays(scene, o, d, ray_offset=start, total_rays=total, clip=clip, image_width=W,
                               init_bins=crop_bins(scene, o, d, obb), nonfinite_flag=flag, **shade_kw)

        if not overlap or len(starts) == 1:
            for gi in range(len(starts)):
                out = group(gi, guard.flag(gi))
                for k, v in out.items():
                    lists.setdefault(k, []).append(v)
            for gi in guard.offenders():
                # (the sampling stage is fp32 in every precision: it reproduces the group's sample positions, so the
                # per-chunk clip bounds it re-accumulates with atomic min / max do not move)
                out = guard.redo(gi, lambda: group(gi))
                for k, v in out.items():
                    lists[k][gi] = v
        else:
            s_samp, s_shade = _streams(dev)
            s_samp.wait_stream(cur)
            s_shade.wait_stream(cur)
            for gi, start in enumerate(starts):
                with torch.cuda.stream(s_samp):
                    o, d, _ = ops.generate_rays(c2w, fx, fy, cx, cy, H, W, dev, start, min(rpl, total - start), distortion=distortion,
                                        camera_type=camera_type)
                    sb, pds, feats = sampling_stage(scene, o, d, clip, start, image_width=W,
                                                    init_bins=crop_bins(scene, o, d, obb), scratch=False)
                    ev = torch.cuda.Event()
                    ev.record(s_samp)
                    for t in (o, d, sb, feats, *pds):   # handed to the other stream: keep the allocator honest
                        if t is not None:
                            t.record_stream(s_shade)
                with torch.cuda.stream(s_shade):
                    s_shade.wait_event(ev)
                    out = shading_stage(scene, o, d, sb, pds, feats, clip, start, image_width=W,
                                        nonfinite_flag=guard.flag(gi), **shade_kw)
                    for v in out.values():
                        v.record_stream(cur)
                for k, v in out.items():
                    lists.setdefault(k, []).append(v)
            cur.wait_stream(s_shade)
            cur.wait_stream(s_samp)
            for gi in guard.offenders():
                for k, v in guard.redo(gi, lambda: group(gi)).items():
                    lists[k][gi] = v
        return {k: torch.cat(v).view(H, W, -1) for k, v in lists.items()}


_STREAMS: Dict[int, Tuple["torch.cuda.Stream", "torch.cuda.Stream"]] = {}


def _streams(dev):
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    if idx not in _STREAMS:
        _STREAMS[idx] = (torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev))
    return _STREAMS[idx]
