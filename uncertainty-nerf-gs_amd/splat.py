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


def active_splatfacto_outputs(gp: Dict[str, torch.Tensor], c2w: torch.Tensor, fx: float, fy: float, cx: float,
                              cy: float, H: int, W: int, background: torch.Tensor, beta_min: float = 0.01,
                              sh_degree: int = 3, rasterize_mode: str = "classic",
                              block_width: int = 16) -> Dict[str, Optional[torch.Tensor]]:
    """gp: gauss_params on the device (means, scales, quats, features_dc, features_rest, opacities,
    log_uncertainties).  Returns the reference's output dict (:359-367) as [H,W,C] tensors."""
    _l.require_gpu()
    means = gp["means"]
    dev = means.device
    V = viewmat_from_c2w(c2w)
    quats = gp["quats"] / gp["quats"].norm(dim=-1, keepdim=True)
    xys, depths, radii, conics, comp, tiles, _cov = ops.splat_project(
        means, torch.exp(gp["scales"]), 1.0, quats.contiguous(), V[:3], fx, fy, cx, cy, H, W, block_width)
    # the reference concatenates features_dc and features_rest first (:242-243); the kernel reads them in place
    rgbs, beta = ops.splat_sh_colors_split(sh_degree, means, c2w[:3, 3], gp["features_dc"].contiguous(),
                                           gp["features_rest"].contiguous(),
                                           gp["log_uncertainties"].reshape(-1).contiguous(), beta_min)
    opac = torch.sigmoid(gp["opacities"]).reshape(-1)
    if rasterize_mode == "antialiased":
        opac = opac * comp
    elif rasterize_mode != "classic":
        raise ValueError(f"Unknown rasterize_mode: {rasterize_mode}")
    opac = opac.contiguous()
    I, _cum, _keys, gids, bins = ops.splat_bin_sort(xys, depths, radii, tiles, H, W, block_width, want_isect_ids=False)
    cols = torch.cat([rgbs, beta[:, None], depths[:, None]], dim=1).contiguous()
    bg5 = torch.cat([background.to(dev, torch.float32), torch.zeros(2, device=dev)])
    img, fT, _ = ops.splat_rasterize(gids, bins, xys, conics, cols, opac, H, W, bg5, block_width)
    alpha = (1.0 - fT)[..., None]
    ops.splat_alpha_normalize(img, 4, fT)           # depth = where(alpha>0, d/alpha, max(d))   (:319)
    sq = ops.splat_depth_sqdiff(xys, depths, img, 4)  # (z_i - depth[floor(xy_i)])^2            (:325-341)
    dv, fT2, _ = ops.splat_rasterize(gids, bins, xys, conics, sq[:, None].contiguous(), opac, H, W, None, block_width)
    ops.splat_alpha_normalize(dv, 0, fT2)           # (:356)
    unc = img[..., 3:4]
    return {"rgb": torch.clamp(img[..., 0:3], max=1.0), "depth": img[..., 4:5], "accumulation": alpha,
            "background": background, "uncertainty": unc, "rgb_var": unc ** 2, "rgb_std": unc,
            "depth_var": dv, "depth_std": dv.sqrt()}
