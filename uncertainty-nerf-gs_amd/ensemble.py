"""Ensemble aggregation (models/ensemble/ensemble_pipeline.py:144-191), single- and multi-GPU.

The reference renders its M members sequentially on one device and then does
`torch.stack(...).mean(0) / .var(0) / .std(0)` per output key.  Here:

  * `aggregate(outputs_list)`        -- members already on one device: the stack/mean/var is the
                                        `unerf_moments` kernel (two-pass fp32, like torch).
  * `aggregate_distributed(outputs)` -- ONE MEMBER PER RANK (one process per GPU, RCCL over xGMI):
    every rank renders the same camera with its own member, then the per-pixel moments are formed
    with the exact two-pass formula on a pixel slice per rank:
        all_to_all(pixel slices of the member images)  ->  rank g reduces its rows  ->  all_gather(reduced slices)
    This is SURVEY.md 8(e) option A (exact parity with torch.stack(...).var(0); no E[x^2]-E[x]^2
    cancellation).  Payload per GPU at 1080p, 6 floats per pixel: 7/8 x 50 MB sent and received in the exchange,
    every xGMI link carrying 1/8 of an image at once (point-to-point, not ring-bound), then 7/8 x 100 MB of
    reduced slices gathered -- well under a millisecond each at ~150 GB/s per link, next to a ~110 ms member render.

The moment math is injected (`moments_fn`) so that the collective plumbing can be tested with
world_size=2 on the gloo backend without a GPU; the default is the HIP kernel.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch

MomentsFn = Callable[[torch.Tensor], Tuple[torch.Tensor, torch.Tensor]]  # [K,N,C] -> mean [N,C], var [N,C]


def _hip_moments(x: torch.Tensor):
    from . import ops
    return ops.moments(x.contiguous())


def _finish(keys: Sequence[str], mean: Dict[str, torch.Tensor], var: Dict[str, torch.Tensor],
            alea: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """The key loop of ensemble_pipeline.py:159-189, in the members' key order.  Order matters and is
    kept: the reference writes outputs[k] = mean for EVERY member key in sequence, so when members
    themselves emit rgb_var / rgb_std / depth_var / depth_std after rgb / depth (the active-nerfacto
    order, activenerfacto_model.py:117-127) the combined epistemic+aleatoric values written while
    visiting "rgb"/"depth" are subsequently overwritten by the member means of those keys; only the
    *_var_alea / *_var_epi keys survive.  Reproduced as is (parity), flagged in DESIGN.md."""
    out: Dict[str, torch.Tensor] = {}
    has_std = "rgb_std" in keys and "depth_std" in keys
    for k in keys:
        out[k] = mean[k]
        if has_std:
            if k in ("rgb", "depth"):
                out[k + "_var_alea"] = alea[k].mean(dim=-1).unsqueeze(-1)
                out[k + "_var_epi"] = var[k].mean(dim=-1).unsqueeze(-1)
                out[k + "_var"] = out[k + "_var_epi"] + out[k + "_var_alea"]
                out[k + "_std"] = out[k + "_var"].sqrt()
        elif k in ("rgb", "depth", "expected_depth"):
            out[k + "_std"] = var[k].sqrt().mean(dim=-1).unsqueeze(-1)
    return out


def aggregate(outputs_list: List[Dict[str, torch.Tensor]], moments_fn: Optional[MomentsFn] = None):
    """All members on one device.  Every tensor is [H,W,C]."""
    moments_fn = moments_fn or _hip_moments
    keys = [k for k, v in outputs_list[0].items() if torch.is_tensor(v)]
    mean, var = {}, {}
    for k in keys:
        x = torch.stack([o[k] for o in outputs_list], dim=0)
        shp = x.shape[1:]
        m, v = moments_fn(x.reshape(x.shape[0], -1, shp[-1]).contiguous())   # also [M,1,3] for a splat `background`
        mean[k], var[k] = m.view(shp), v.view(shp)
    alea = {k: mean[k + "_var"] for k in ("rgb", "depth") if k + "_var" in mean}
    return _finish(keys, mean, var, alea)


def pixel_slice(num_pixels: int, rank: int, world: int) -> Tuple[int, int]:
    """contiguous, balanced split of the flattened pixel axis (bit-exact bookkeeping: the slices
    tile [0,num_pixels) without gaps or overlap for every world size)."""
    base, rem = divmod(num_pixels, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def _image_keys(member: Dict[str, torch.Tensor]) -> Tuple[List[str], List[str]]:
    """tensor keys split into per-pixel images [H,W,C] and everything else (active-splatfacto's `background` [3],
    activesplatfacto_model.py:363): the reference stacks and averages every key alike (ensemble_pipeline.py:159-162),
    only the images need the pixel-sliced exchange"""
    keys = [k for k, v in member.items() if torch.is_tensor(v)]
    H, W = next(v.shape[:2] for v in (member[k] for k in keys) if v.dim() == 3)
    img = [k for k in keys if member[k].dim() == 3 and tuple(member[k].shape[:2]) == (H, W)]
    return img, [k for k in keys if k not in img]


class _StageClock:
    """per-stage device times of aggregate_distributed (events on the current stream; the collectives of a NCCL / RCCL
    group make the current stream wait for them, so the event behind a collective is behind its data)"""

    def __init__(self, sink: Optional[dict], device):
        self.sink, self.marks = sink, []
        self.on = sink is not None and torch.device(device).type == "cuda"
        self.mark("start")

    def mark(self, name: str) -> None:
        if self.on:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self.marks.append((name, ev))

    def finish(self) -> None:
        if not self.on:
            return
        self.marks[-1][1].synchronize()
        for (_, e0), (name, e1) in zip(self.marks[:-1], self.marks[1:]):
            self.sink[name] = self.sink.get(name, 0.0) + e0.elapsed_time(e1)
        self.sink["calls"] = self.sink.get("calls", 0) + 1


def aggregate_distributed(outputs, group=None, moments_fn: Optional[MomentsFn] = None, stage_ms: Optional[dict] = None):
    """This rank's member outputs -> the ensemble outputs, identical on every rank.
    `outputs` is one member's dict ([H,W,C] per key) or a list of such dicts when a rank holds several
    members (M members over N < M GPUs; every rank must hold the same number).

    Exchange (SURVEY.md 8e option A): ONE all_to_all of pixel slices -- rank g receives rows
    [a_g, b_g) of every member's packed image, (N-1)/N of one packed image sent and received per rank, all xGMI links
    busy at once -- then the exact two-pass moments on the slice (what torch.stack(...).mean(0) / .var(0) compute),
    then ONE all_gather of the reduced slices.  1080p, 6 floats per pixel: 44 MB out, 44 MB in per rank for the
    exchange (the two all_gathers of whole member images this replaces moved 8x that)."""
    import torch.distributed as dist
    moments_fn = moments_fn or _hip_moments
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    members = outputs if isinstance(outputs, (list, tuple)) else [outputs]
    all_keys = [k for k, v in members[0].items() if torch.is_tensor(v)]   # member key order (it matters, see _finish)
    keys, small_keys = _image_keys(members[0])
    widths = [members[0][k].shape[-1] for k in keys]
    H, W = members[0][keys[0]].shape[:2]
    P = H * W
    ml = len(members)
    # one packed [m_local, P, sum(C)] block per rank -> a single collective instead of one per key
    clock = _StageClock(stage_ms, members[0][keys[0]].device)
    packed = torch.stack([torch.cat([m[k].reshape(P, -1) for k in keys], dim=-1) for m in members], dim=0).contiguous()
    Ctot = packed.shape[-1]
    edges = [pixel_slice(P, r, world) for r in range(world)]
    a, b = edges[rank]
    # send block for destination r: this rank's members restricted to r's pixel rows, [m_local * (b_r - a_r), Ctot]
    send = torch.cat([packed[:, ar:br].reshape(-1, Ctot) for ar, br in edges], dim=0).contiguous()
    recv = torch.empty(world * ml * (b - a), Ctot, dtype=packed.dtype, device=packed.device)
    clock.mark("pack")
    dist.all_to_all_single(recv, send, output_split_sizes=[ml * (b - a)] * world,
                           input_split_sizes=[ml * (br - ar) for ar, br in edges], group=group)
    clock.mark("all_to_all")
    stack = recv.view(world * ml, b - a, Ctot)                      # [M, b-a, Ctot], member order = rank-major
    if b > a:
        m, v = moments_fn(stack.contiguous())
    else:
        m = v = torch.zeros(0, Ctot, dtype=packed.dtype, device=packed.device)
    part = torch.cat([m, v], dim=-1).contiguous()                   # [b-a, 2*Ctot]
    # slices differ by at most one row: pad to the largest so all_gather sees equal shapes
    rows = max(br - ar for ar, br in edges)
    padded = torch.zeros(rows, 2 * Ctot, dtype=part.dtype, device=part.device)
    padded[: b - a] = part
    parts = [torch.empty_like(padded) for _ in range(world)]
    clock.mark("moments")
    dist.all_gather(parts, padded, group=group)
    clock.mark("all_gather")
    full = torch.cat([parts[r][: edges[r][1] - edges[r][0]] for r in range(world)], dim=0)
    mean, var, off = {}, {}, 0
    for k, c in zip(keys, widths):
        mean[k] = full[:, off:off + c].reshape(H, W, c)
        var[k] = full[:, Ctot + off:Ctot + off + c].reshape(H, W, c)
        off += c
    if small_keys:   # a few floats per member: gathered whole, reduced on every rank
        flat = torch.stack([torch.cat([m[k].reshape(-1) for k in small_keys]) for m in members], dim=0).contiguous()
        allf = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(allf, flat, group=group)
        sm, sv = moments_fn(torch.cat(allf, dim=0)[:, None, :].contiguous())       # [M, 1, n] -> [1, n]
        off = 0
        for k in small_keys:
            n = members[0][k].numel()
            mean[k], var[k] = sm[0, off:off + n].reshape(members[0][k].shape), sv[0, off:off + n].reshape(members[0][k].shape)
            off += n
    alea = {k: mean[k + "_var"] for k in ("rgb", "depth") if k + "_var" in mean}
    res = _finish(all_keys, mean, var, alea)
    clock.mark("unpack")
    clock.finish()
    if stage_ms is not None:
        stage_ms["bytes_all_to_all_sent_per_rank"] = int(send.numel() * send.element_size() * (world - 1) // max(world, 1))
        stage_ms["bytes_all_gather_received_per_rank"] = int(padded.numel() * padded.element_size() * (world - 1))
        stage_ms["packed_image_bytes_per_member"] = int(P * Ctot * packed.element_size())
    return res


class EnsemblePipeline:
    """The render-side surface of models/ensemble/ensemble_pipeline.py: `models` (the members this process holds)
    and `get_ensemble_outputs_for_camera_ray_bundle(camera, obb_box)` (:144-191), which is what
    scripts/eval_uncertainty.py:1127 calls.  The reference keeps all M members in one process and renders them
    in sequence; here a process holds M / world_size of them (all M without torch.distributed, one per GPU under
    `torchrun`), and the per-pixel moments go through aggregate / aggregate_distributed.  Pass built Models; the
    members' checkpoints (ensemble_pipeline.py:62-108, ensemble_utils.py:36-110) are read by
    `checkpoints.load_ensemble(models, config_paths)`."""

    def __init__(self, models: Sequence, group=None, moments_fn: Optional[MomentsFn] = None):
        self.models = torch.nn.ModuleList(models) if all(isinstance(m, torch.nn.Module) for m in models) else list(models)
        self.group, self.moments_fn = group, moments_fn

    @property
    def model(self):  # VanillaPipeline.model = the first member (ensemble_pipeline.py:51)
        return self.models[0]

    def _distributed(self) -> bool:
        import torch.distributed as dist
        return dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1

    @torch.no_grad()
    def get_ensemble_outputs_for_camera_ray_bundle(self, camera, obb_box=None) -> Dict[str, torch.Tensor]:
        kw = {} if obb_box is None else {"obb_box": obb_box}
        outs = [m.get_outputs_for_camera(camera, **kw) for m in self.models]
        if self._distributed():
            return aggregate_distributed(outs, group=self.group, moments_fn=self.moments_fn)
        assert len(outs) > 1, "Ensemble requires at least two models."
        return aggregate(outs, moments_fn=self.moments_fn)


def views_for_rank(num_views: int, rank: int, world: int) -> List[int]:
    """View-batch data parallelism for splats / single models: replicas of the scene,
    disjoint cameras per rank, no data-path collective."""
    return list(range(rank, num_views, world))
