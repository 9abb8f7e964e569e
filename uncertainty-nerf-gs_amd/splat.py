"""Frame-level active-splatfacto pipeline on the HIP kernels.

Mirrors ActiveSplatfactoModel.get_outputs (models/activesplatfacto/activesplatfacto_model.py:142-367),
eval branch, with the MI355X restructuring: the reference calls gsplat's rasterize_gaussians four
times (rgb, beta, depth, depth-variance), each repeating the bin-and-sort; here the intersections
are binned and sorted ONCE and rgb + beta + depth are blended in one 5-channel pass, followed by
the data-dependent depth-variance pass (it needs the finished depth image, :336).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import lib as _l
from . import ops


def viewmat_from_c2w(c2w: torch.Tensor) -> torch.Tensor:
    """activesplatfacto_model.py:184-195: flip y/z to gsplat's convention, analytic inverse."""
    c2w = c2w.detach().to("cpu", torch.float32)
    R = c2w[:3, :3] @ torch.diag(torch.tensor([1.0, -1.0, -1.0]))
    T = c2w[:3, 3:4]
    Rinv = R.T
    V = torch.eye(4)
    V[:3, :3] = Rinv
    V[:3, 3:4] = -Rinv @ T
    return V


# [UPSTREAM nerfstudio 1.1.0 SplatfactoModel.populate_modules] the stored eval background: config "random" ->
# the Viser grey, otherwise the named colour (nerfstudio.utils.colors.get_color)
BACKGROUND_RANDOM = (0.1490, 0.1647, 0.2157)
NAMED_COLORS = {"white": (1.0, 1.0, 1.0), "black": (0.0, 0.0, 0.0), "red": (1.0, 0.0, 0.0), "green": (0.0, 1.0, 0.0),
                "blue": (0.0, 0.0, 1.0)}


def background_for(config_background_color: str) -> torch.Tensor:
    if config_background_color == "random":
        return torch.tensor(BACKGROUND_RANDOM, dtype=torch.float32)
    if config_background_color not in NAMED_COLORS:
        raise ValueError(f"unknown background_color {config_background_color!r}; known: random, {', '.join(NAMED_COLORS)}")
    return torch.tensor(NAMED_COLORS[config_background_color], dtype=torch.float32)


def empty_outputs(W: int, H: int, background: torch.Tensor) -> Dict[str, torch.Tensor]:
    """[UPSTREAM SplatfactoModel.get_empty_outputs] what the reference returns when the crop box holds no splat or
    every projected radius is zero (activesplatfacto_model.py:176-177, 239-240): background image, depth 10, zero
    accumulation -- and none of the uncertainty keys."""
    rgb = background.repeat(H, W, 1)
    return {"rgb": rgb, "depth": background.new_ones(H, W, 1) * 10, "accumulation": background.new_zeros(H, W, 1),
            "background": background}


def active_splatfacto_outputs(gp: Dict[str, torch.Tensor], c2w: torch.Tensor, fx: float, fy: float, cx: float,
                              cy: float, H: int, W: int, background: torch.Tensor, beta_min: float = 0.01,
                              sh_degree: int = 3, rasterize_mode: str = "classic",
                              block_width: int = 16, crop_ids: Optional[torch.Tensor] = None,
                              config_sh_degree: Optional[int] = None, tight: bool = True) -> Dict[str, Optional[torch.Tensor]]:
    """gp: gauss_params on the device (means, scales, quats, features_dc, features_rest, opacities,
    log_uncertainties).  Returns the reference's output dict (:359-367) as [H,W,C] tensors.
    Without `log_uncertainties` in gp: plain splatfacto [UPSTREAM nerfstudio 1.1.0 SplatfactoModel.get_outputs, the
    parent the reference extends and the member type of its splat ensembles] -- rgb + depth blended in one 4-channel
    pass, outputs rgb / depth / accumulation / background only.
    crop_ids: bool [N] from `crop_box.within(means)` (:174-180, 202-217) -- only those splats are rendered.
    sh_degree: the active degree n = min(step // interval, config.sh_degree) (:244);
    config_sh_degree == 0 selects the sigmoid(features_dc) colours of :247-248.
    tight: bin each splat into the tiles its alpha >= 1/255 ellipse reaches instead of gsplat's whole radius box (the
    left-out pairs are ones the blend loop skips itself: same output bits, about half the sort and staging work);
    False: gsplat's lists."""
    _l.require_gpu()
    background = background.to(gp["means"].device, torch.float32)
    if gp["means"].shape[0] == 0:          # no splats at all: the same picture as "nothing visible" (:239-240)
        return empty_outputs(W, H, background)
    if crop_ids is not None:
        crop_ids = crop_ids.reshape(-1).to(gp["means"].device)
        if int(crop_ids.sum().item()) == 0:
            return empty_outputs(W, H, background)
        gp = {k: v[crop_ids].contiguous() for k, v in gp.items()}
    means = gp["means"].contiguous()
    dev = means.device
    # the pose goes to the host ONCE, here, before anything is queued: a device-resident camera_to_worlds (the normal
    # nerfstudio model path) would otherwise be read back behind the projection and the scan -- a blocking stream sync
    # exactly where the intersection count's read-back is meant to overlap the SH-colour kernel
    c2w = c2w.detach().to("cpu", torch.float32)
    V = viewmat_from_c2w(c2w)
    if rasterize_mode not in ("classic", "antialiased"):
        raise ValueError(f"Unknown rasterize_mode: {rasterize_mode}")
    # exp(scales) and quats / quats.norm() (:221-223) are taken inside the projection kernel; with `tight` also
    # sigmoid(opacities) [* comp] (:252-256), which the tight tile counts depend on
    logits = gp["opacities"].reshape(-1).contiguous()
    proj = ops.splat_project(means, gp["scales"].contiguous(), 1.0, gp["quats"].contiguous(), V[:3], fx, fy, cx, cy, H, W,
                             block_width, raw=True, opacity_logits=logits if tight else None,
                             antialiased=rasterize_mode == "antialiased")
    xys, depths, radii, conics, comp, tiles, _cov = proj[:7]
    opac = proj[7] if tight else None
    # the intersection count (the frame's one host read-back) starts its way to the host now and is awaited inside
    # splat_bin_sort; the SH colours do not depend on it and keep the GPU busy meanwhile
    count = ops.SplatCount(tiles, defer_copy=True)
    if config_sh_degree is not None and config_sh_degree <= 0:
        sh_degree = -1                                     # kernel: colours = sigmoid(features_dc)
    # the reference concatenates features_dc and features_rest first (:242-243); the kernel reads them in place
    plain = "log_uncertainties" not in gp
    # one launch leaves the rasteriser's per-splat rows [rgb, (beta), depth] (and the opacities unless made above)
    cols, opac2 = ops.splat_shade_inputs(sh_degree, means, c2w[:3, 3], gp["features_dc"].contiguous(),
                                         gp["features_rest"].contiguous(),
                                         None if plain else gp["log_uncertainties"].reshape(-1).contiguous(), beta_min,
                                         None if tight else logits,
                                         comp if rasterize_mode == "antialiased" else None, depths)
    count.start_copy()      # (its host-side set-up runs under the SH kernel just queued)
    opac = opac if tight else opac2
    # (self.radii).sum() == 0 -> get_empty_outputs (:239-240).  A splat has a non-zero radius exactly when it hits at
    # least one tile, so "no intersections" is the same test and rides on the one host read-back of the frame
    I, _cum, _keys, gids, bins = ops.splat_bin_sort(xys, depths, radii, tiles, H, W, block_width, want_isect_ids=False,
                                                    count=count, tight=(conics, opac) if tight else None)
    if I == 0 and not (tight and bool((radii > 0).any())):
        return empty_outputs(W, H, background)
    # (tight lists can be empty while splats are "visible" by radius -- every opacity below 1/255: the reference then
    # rasterises a frame in which nothing blends, which the empty lists give as well; the extra read-back is on that path
    # only)
    # frame scratch: the background padded to the row length, and the two channel maxima the rasteriser passes leave for
    # their alpha normalisations
    Cn = cols.shape[1]
    scratch = torch.zeros(Cn + 2, device=dev)
    scratch[:3] = background
    bg, mx1, mx2 = scratch[:Cn], scratch[Cn:Cn + 1], scratch[Cn + 1:Cn + 2]
    if plain:
        img, fT, _ = ops.splat_rasterize(gids, bins, xys, conics, cols, opac, H, W, bg, block_width, chan_max=(3, mx1))
        # depth = where(alpha > 0, d / alpha, max(d)), rgb clamp and accumulation in one pass over the pixels
        rgb, alpha, _, _ = ops.splat_normalize_outputs(img, 3, fT, mx1, rgb=True, acc=True)
        return {"rgb": rgb, "depth": img[..., 3:4], "accumulation": alpha, "background": background}
    img, fT, fidx = ops.splat_rasterize(gids, bins, xys, conics, cols, opac, H, W, bg, block_width, want_final_idx=True,
                                        chan_max=(4, mx1))
    # depth = where(alpha>0, d/alpha, max(d)) (:319) + clamp(rgb, max=1) (:275), 1 - final_T, uncertainty^2 in the same pass
    rgb, alpha, rgb_var, _ = ops.splat_normalize_outputs(img, 4, fT, mx1, rgb=True, acc=True, sq_ch=3)
    sq = ops.splat_depth_sqdiff(xys, depths, img, 4)       # (z_i - depth[floor(xy_i)])^2            (:325-341)
    # same ids / bins / geometry / opacities as the first pass: every pixel stops at the index that pass ended on
    dv, fT2, _ = ops.splat_rasterize(gids, bins, xys, conics, sq[:, None], opac, H, W, None, block_width,
                                     stop_idx=fidx, chan_max=(0, mx2))
    _, _, _, dstd = ops.splat_normalize_outputs(dv, 0, fT2, mx2, sqrt=True)   # (:356) + depth_std = sqrt(depth_var)
    unc = img[..., 3:4]
    return {"rgb": rgb, "depth": img[..., 4:5], "accumulation": alpha,
            "background": background, "uncertainty": unc, "rgb_var": rgb_var, "rgb_std": unc,
            "depth_var": dv, "depth_std": dstd}
