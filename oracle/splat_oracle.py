"""CPU (numpy fp32) restatement of the Gaussian-splat half of the hot path.

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.

Restates gsplat==0.1.11 (README.md:30 of the reference pins it; not vendored, not installed ->
"parity unpinned") as called from models/activesplatfacto/activesplatfacto_model.py:142-367,
plus the reference's own glue in that function.

Every arithmetic step is a separate fp32 numpy op in a fixed left-to-right order, which is the
order csrc/unerf_splat.hip uses under -ffp-contract=off, so projection, tile boxes, sort keys
and bin edges compare BIT-EXACT; the rasteriser (exp) compares within tolerance.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import numpy as np

f32 = np.float32


def _f(x):
    return np.asarray(x, dtype=np.float32)


def project_gaussians(means3d, scales, glob_scale, quats, viewmat, fx, fy, cx, cy, H, W, block_width=16,
                      clip_thresh=0.01):
    """gsplat project_gaussians_forward_kernel.  Returns dict of zero-initialised outputs."""
    means3d, scales, quats, V = _f(means3d), _f(scales), _f(quats), _f(viewmat).reshape(-1)[:12].reshape(3, 4)
    N = means3d.shape[0]
    fx, fy, cx, cy, gs, clip = f32(fx), f32(fy), f32(cx), f32(cy), f32(glob_scale), f32(clip_thresh)
    out = dict(xys=np.zeros((N, 2), f32), depths=np.zeros(N, f32), radii=np.zeros(N, np.int32),
               conics=np.zeros((N, 3), f32), compensation=np.zeros(N, f32), num_tiles_hit=np.zeros(N, np.int32),
               cov3d=np.zeros((N, 6), f32))
    p0, p1, p2 = means3d[:, 0], means3d[:, 1], means3d[:, 2]
    with np.errstate(all="ignore"):
        tx = ((V[0, 0] * p0 + V[0, 1] * p1) + V[0, 2] * p2) + V[0, 3]
        ty = ((V[1, 0] * p0 + V[1, 1] * p1) + V[1, 2] * p2) + V[1, 3]
        tz = ((V[2, 0] * p0 + V[2, 1] * p1) + V[2, 2] * p2) + V[2, 3]
        live = ~(tz <= clip)
        qw, qx, qy, qz = quats[:, 0], quats[:, 1], quats[:, 2], quats[:, 3]
        qs = f32(1) / np.sqrt(((qw * qw + qx * qx) + qy * qy) + qz * qz)
        w, x, y, z = qw * qs, qx * qs, qy * qs, qz * qs
        one, two = f32(1), f32(2)
        R = [[one - two * (y * y + z * z), two * (x * y - w * z), two * (x * z + w * y)],
             [two * (x * y + w * z), one - two * (x * x + z * z), two * (y * z - w * x)],
             [two * (x * z - w * y), two * (y * z + w * x), one - two * (x * x + y * y)]]
        s = [gs * scales[:, 0], gs * scales[:, 1], gs * scales[:, 2]]
        M = [[R[r][c] * s[c] for c in range(3)] for r in range(3)]
        Sg = [[(M[r][0] * M[c][0] + M[r][1] * M[c][1]) + M[r][2] * M[c][2] for c in range(3)] for r in range(3)]
        cov3d = np.stack([Sg[0][0], Sg[0][1], Sg[0][2], Sg[1][1], Sg[1][2], Sg[2][2]], axis=-1)
        C3 = [[Sg[0][0], Sg[0][1], Sg[0][2]], [Sg[0][1], Sg[1][1], Sg[1][2]], [Sg[0][2], Sg[1][2], Sg[2][2]]]
        tan_fovx, tan_fovy = f32(0.5) * f32(W) / fx, f32(0.5) * f32(H) / fy
        lim_x, lim_y = f32(1.3) * tan_fovx, f32(1.3) * tan_fovy
        ex = tz * np.minimum(lim_x, np.maximum(-lim_x, tx / tz))
        ey = tz * np.minimum(lim_y, np.maximum(-lim_y, ty / tz))
        rz = f32(1) / tz
        rz2 = rz * rz
        J00, J02, J11, J12 = fx * rz, (-fx * ex) * rz2, fy * rz, (-fy * ey) * rz2
        T0 = [J00 * V[0, c] + J02 * V[2, c] for c in range(3)]
        T1 = [J11 * V[1, c] + J12 * V[2, c] for c in range(3)]
        TV0 = [(T0[0] * C3[0][c] + T0[1] * C3[1][c]) + T0[2] * C3[2][c] for c in range(3)]
        TV1 = [(T1[0] * C3[0][c] + T1[1] * C3[1][c]) + T1[2] * C3[2][c] for c in range(3)]
        c00 = (TV0[0] * T0[0] + TV0[1] * T0[1]) + TV0[2] * T0[2]
        c01 = (TV0[0] * T1[0] + TV0[1] * T1[1]) + TV0[2] * T1[2]
        c11 = (TV1[0] * T1[0] + TV1[1] * T1[1]) + TV1[2] * T1[2]
        det_orig = c00 * c11 - c01 * c01
        ca, cb, cc = c00 + f32(0.3), c01, c11 + f32(0.3)
        det = ca * cc - cb * cb
        comp = np.sqrt(np.maximum(f32(0), det_orig / det))
        ok_det = live & ~(det == 0)
        inv_det = f32(1) / det
        conics = np.stack([cc * inv_det, -cb * inv_det, ca * inv_det], axis=-1)
        bh = f32(0.5) * (ca + cc)
        sq = np.sqrt(np.maximum(f32(0.1), bh * bh - det))
        v1, v2 = bh + sq, bh - sq
        radius = np.ceil(f32(3) * np.sqrt(np.maximum(v1, v2)))
        rw = f32(1) / (tz + f32(1e-6))
        u = (tx * rw) * fx + cx
        v = (ty * rw) * fy + cy
        x0, y0, x1, y1 = tile_bbox(u, v, radius, block_width, H, W)
        area = (x1 - x0) * (y1 - y0)
        ok = ok_det & (area > 0)
    out["cov3d"][live] = cov3d[live]
    out["conics"][ok_det] = conics[ok_det]
    out["num_tiles_hit"][ok] = area[ok]
    out["depths"][ok] = tz[ok]
    out["radii"][ok] = radius[ok].astype(np.int32)
    out["xys"][ok, 0] = u[ok]
    out["xys"][ok, 1] = v[ok]
    out["compensation"][ok] = comp[ok]
    return out


def tile_bbox(cx, cy, radius, bw, H, W):
    tbx, tby = (W + bw - 1) // bw, (H + bw - 1) // bw
    with np.errstate(all="ignore"):
        tcx, tcy, tr = _f(cx) / f32(bw), _f(cy) / f32(bw), _f(radius) / f32(bw)
        # (int) truncation toward zero; NaN / huge values are never live (masked by caller)
        def ti(a):
            a = np.nan_to_num(a, nan=0.0, posinf=2e9, neginf=-2e9)
            return np.clip(np.trunc(a), -2e9, 2e9).astype(np.int64)
        x0 = np.minimum(np.maximum(0, ti(tcx - tr)), tbx)
        x1 = np.minimum(np.maximum(0, ti(tcx + tr + f32(1))), tbx)
        y0 = np.minimum(np.maximum(0, ti(tcy - tr)), tby)
        y1 = np.minimum(np.maximum(0, ti(tcy + tr + f32(1))), tby)
    return x0.astype(np.int32), y0.astype(np.int32), x1.astype(np.int32), y1.astype(np.int32)


SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
SH_C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
         1.445305721320277, -0.5900435899266435]


def spherical_harmonics(degree: int, viewdirs, coeffs):
    """gsplat compute_sh_forward (viewdirs normalised inside); coeffs [N,16,3] -> [N,3]"""
    v, k = _f(viewdirs), _f(coeffs)
    col = f32(SH_C0) * k[:, 0]
    if degree < 1:
        return col
    n = np.sqrt(np.sum(v * v, axis=-1, keepdims=True))
    x, y, z = (v / n)[:, 0:1], (v / n)[:, 1:2], (v / n)[:, 2:3]
    xx, xy, xz, yy, yz, zz = x * x, x * y, x * z, y * y, y * z, z * z
    col = col + f32(SH_C1) * (-y * k[:, 1] + z * k[:, 2] - x * k[:, 3])
    if degree >= 2:
        col = col + (f32(SH_C2[0]) * xy * k[:, 4] + f32(SH_C2[1]) * yz * k[:, 5]
                     + f32(SH_C2[2]) * (f32(2) * zz - xx - yy) * k[:, 6] + f32(SH_C2[3]) * xz * k[:, 7]
                     + f32(SH_C2[4]) * (xx - yy) * k[:, 8])
    if degree >= 3:
        col = col + (f32(SH_C3[0]) * y * (f32(3) * xx - yy) * k[:, 9] + f32(SH_C3[1]) * xy * z * k[:, 10]
                     + f32(SH_C3[2]) * y * (f32(4) * zz - xx - yy) * k[:, 11]
                     + f32(SH_C3[3]) * z * (f32(2) * zz - f32(3) * xx - f32(3) * yy) * k[:, 12]
                     + f32(SH_C3[4]) * x * (f32(4) * zz - xx - yy) * k[:, 13] + f32(SH_C3[5]) * z * (xx - yy) * k[:, 14]
                     + f32(SH_C3[6]) * x * (xx - f32(3) * yy) * k[:, 15])
    return col.astype(f32)


def softplus(x):
    x = _f(x)
    return np.where(x > 20, x, np.log1p(np.exp(np.minimum(x, f32(20))))).astype(f32)


def bin_and_sort(xys, depths, radii, num_tiles_hit, H, W, bw=16):
    """compute_cumulative_intersects + map_gaussian_to_intersects + radix sort + get_tile_bin_edges.
    Stable sort on (tile_id << 32 | depth bits) == radix sort order."""
    tbx, tby = (W + bw - 1) // bw, (H + bw - 1) // bw
    cum = np.cumsum(num_tiles_hit.astype(np.int64)).astype(np.int32)
    I = int(cum[-1]) if len(cum) else 0
    keys = np.zeros(I, np.int64)
    vals = np.zeros(I, np.int32)
    x0, y0, x1, y1 = tile_bbox(xys[:, 0], xys[:, 1], radii.astype(f32), bw, H, W)
    dbits = depths.astype(f32).view(np.int32).astype(np.int64)
    for i in np.nonzero(radii > 0)[0]:
        cur = 0 if i == 0 else int(cum[i - 1])
        for ty in range(y0[i], y1[i]):
            for tx in range(x0[i], x1[i]):
                keys[cur] = ((ty * tbx + tx) << 32) | dbits[i]
                vals[cur] = i
                cur += 1
    order = np.argsort(keys, kind="stable")
    keys, vals = keys[order], vals[order]
    bins = np.zeros((tbx * tby, 2), np.int32)
    if I:
        tiles = (keys >> 32).astype(np.int64)
        starts = np.nonzero(np.diff(tiles, prepend=-1))[0]
        for a, b in zip(starts, list(starts[1:]) + [I]):
            bins[tiles[a]] = (a, b)
    return I, cum, keys, vals, bins


def rasterize(gids, bins, xys, conics, colors, opacities, H, W, background=None, bw=16):
    """rasterize_forward / nd_rasterize_forward for C channels.  colors [N,C], opacities [N].
    -> out [H,W,C], final_T [H,W], final_idx [H,W]"""
    colors, xys, conics, opacities = _f(colors), _f(xys), _f(conics), _f(opacities).reshape(-1)
    C = colors.shape[1]
    bg = np.zeros(C, f32) if background is None else _f(background)
    tbx, tby = (W + bw - 1) // bw, (H + bw - 1) // bw
    out = np.zeros((H, W, C), f32)
    fT = np.ones((H, W), f32)
    fidx = np.zeros((H, W), np.int32)
    for ty in range(tby):
        for tx in range(tbx):
            r0, r1 = bins[ty * tbx + tx]
            ys = np.arange(ty * bw, min((ty + 1) * bw, H))
            xs = np.arange(tx * bw, min((tx + 1) * bw, W))
            py, px = np.meshgrid(ys.astype(f32) + f32(0.5), xs.astype(f32) + f32(0.5), indexing="ij")
            T = np.ones(py.shape, f32)
            pix = np.zeros(py.shape + (C,), f32)
            done = np.zeros(py.shape, bool)
            cur = np.zeros(py.shape, np.int32)
            for idx in range(r0, r1):
                if done.all():
                    break
                g = gids[idx]
                dx, dy = xys[g, 0] - px, xys[g, 1] - py
                ca, cb, cc = conics[g]
                sigma = f32(0.5) * (ca * dx * dx + cc * dy * dy) + cb * dx * dy
                with np.errstate(over="ignore"):
                    alpha = np.minimum(f32(0.999), opacities[g] * np.exp(-sigma))
                act = ~done & ~((sigma < 0) | (alpha < f32(1.0) / f32(255.0)))
                nT = T * (f32(1) - alpha)
                stop = act & (nT <= f32(1e-4))
                done |= stop
                act &= ~stop
                vis = alpha * T
                pix[act] += colors[g][None, :] * vis[act][:, None]
                T[act] = nT[act]
                cur[act] = idx
            out[ys[0]:ys[-1] + 1, xs[0]:xs[-1] + 1] = pix + T[..., None] * bg
            fT[ys[0]:ys[-1] + 1, xs[0]:xs[-1] + 1] = T
            fidx[ys[0]:ys[-1] + 1, xs[0]:xs[-1] + 1] = cur
    return out, fT, fidx


def viewmat_from_c2w(c2w):
    """[REF activesplatfacto_model.py:184-195] y/z flip, analytic inverse."""
    c2w = _f(c2w)
    R = c2w[:3, :3] @ np.diag(_f([1, -1, -1]))
    T = c2w[:3, 3:4]
    Rinv = R.T
    Tinv = -Rinv @ T
    V = np.eye(4, dtype=f32)
    V[:3, :3] = Rinv
    V[:3, 3:4] = Tinv
    return V


def empty_outputs(W, H, background) -> Dict[str, np.ndarray]:
    """[UPSTREAM SplatfactoModel.get_empty_outputs] (activesplatfacto_model.py:176-177, 239-240)"""
    bg = _f(background)
    return {"rgb": np.broadcast_to(bg, (H, W, 3)).copy(), "depth": np.full((H, W, 1), 10, f32),
            "accumulation": np.zeros((H, W, 1), f32), "background": bg}


def active_splatfacto_outputs(gp: Dict[str, np.ndarray], c2w, fx, fy, cx, cy, H, W, background, beta_min=0.01,
                              sh_degree=3, rasterize_mode="classic", config_sh_degree=None,
                              crop_ids=None) -> Dict[str, np.ndarray]:
    """[REF activesplatfacto_model.py:142-367] eval branch.  sh_degree = the active degree n (:244);
    config_sh_degree == 0 -> sigmoid(features_dc) colours (:247-248); crop_ids: bool [N] of `crop_box.within`.
    The four rasterize_gaussians calls share their blending weights, so they are evaluated as one
    5-channel pass (rgb, beta, depth) plus the depth-variance pass -- pinned to the reference's own four-pass code by
    tests/golden/splat_get_outputs.npz (fake-self run of ActiveSplatfactoModel.get_outputs)."""
    if crop_ids is not None:
        crop_ids = np.asarray(crop_ids).reshape(-1).astype(bool)
        if crop_ids.sum() == 0:
            return empty_outputs(W, H, background)
        gp = {k: np.asarray(v)[crop_ids] for k, v in gp.items()}
    import torch   # the elementwise activations are torch's in the reference (torch.exp / sigmoid / Softplus / norm)
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(_f(a)))
    means = _f(gp["means"])
    V = viewmat_from_c2w(c2w)
    quats = tt(gp["quats"])
    quats = (quats / quats.norm(dim=-1, keepdim=True)).numpy()
    pr = project_gaussians(means, torch.exp(tt(gp["scales"])).numpy(), 1.0, quats, V[:3], fx, fy, cx, cy, H, W, 16)
    coeffs = np.concatenate([_f(gp["features_dc"])[:, None, :], _f(gp["features_rest"])], axis=1)
    viewdirs = means - _f(c2w)[:3, 3]
    if pr["radii"].sum() == 0:
        return empty_outputs(W, H, background)
    if config_sh_degree is not None and config_sh_degree <= 0:
        rgbs = torch.sigmoid(tt(coeffs)[:, 0, :]).numpy()   # the same strided view the reference takes (:248)
    else:
        rgbs = np.maximum(spherical_harmonics(sh_degree, viewdirs, coeffs) + f32(0.5), f32(0))
    opac = torch.sigmoid(tt(gp["opacities"])).numpy().reshape(-1)
    if rasterize_mode == "antialiased":
        opac = opac * pr["compensation"]
    elif rasterize_mode != "classic":
        raise ValueError(f"Unknown rasterize_mode: {rasterize_mode}")
    beta = (torch.nn.functional.softplus(tt(gp["log_uncertainties"])) + beta_min).numpy().reshape(-1)
    I, cum, keys, gids, bins = bin_and_sort(pr["xys"], pr["depths"], pr["radii"], pr["num_tiles_hit"], H, W)
    cols = np.concatenate([rgbs, beta[:, None], pr["depths"][:, None]], axis=1)
    bg5 = np.concatenate([_f(background), np.zeros(2, f32)])
    img, fT, _ = rasterize(gids, bins, pr["xys"], pr["conics"], cols, opac, H, W, bg5)
    alpha = (f32(1) - fT)[..., None]
    rgb = np.minimum(img[..., :3], f32(1))
    unc = img[..., 3:4]
    d = img[..., 4:5]
    with np.errstate(all="ignore"):
        depth = np.where(alpha > 0, d / alpha, d.max())
    pix = np.floor(pr["xys"]).astype(np.int64)
    valid = (pix[:, 0] > 0) & (pix[:, 0] < W) & (pix[:, 1] > 0) & (pix[:, 1] < H)
    diff = pr["depths"].copy()
    diff[valid] -= depth[pix[valid, 1], pix[valid, 0], 0]
    dv_img, _, _ = rasterize(gids, bins, pr["xys"], pr["conics"], (diff ** 2)[:, None], opac, H, W, np.zeros(1, f32))
    with np.errstate(all="ignore"):
        depth_var = np.where(alpha > 0, dv_img / alpha, dv_img.max())
    return {"rgb": rgb, "depth": depth, "accumulation": alpha, "background": _f(background), "uncertainty": unc,
            "rgb_var": unc ** 2, "rgb_std": unc, "depth_var": depth_var, "depth_std": torch.sqrt(tt(depth_var)).numpy(),
            "_proj": pr, "_sort": (I, cum, keys, gids, bins), "_sqdiff": diff ** 2}


def splatfacto_outputs(gp: Dict[str, np.ndarray], c2w, fx, fy, cx, cy, H, W, background, **kw) -> Dict[str, np.ndarray]:
    """[UPSTREAM nerfstudio 1.1.0 SplatfactoModel.get_outputs, eval branch] plain splatfacto -- the parent whose
    get_outputs the reference extends (activesplatfacto_model.py:142-319 up to the depth pass) and the member type of
    its splat ensembles (ensemble_utils.py:153-156): rgb (clamped at 1), depth = where(alpha > 0, d / alpha, max d),
    accumulation, background.  Restated as the corresponding outputs of the active model with the uncertainty channel
    unused (the rasteriser's blending weights do not depend on it)."""
    gp = dict(gp)
    gp.setdefault("log_uncertainties", np.zeros((np.asarray(gp["means"]).shape[0], 1), f32))
    out = active_splatfacto_outputs(gp, c2w, fx, fy, cx, cy, H, W, background, **kw)
    return {k: out[k] for k in ("rgb", "depth", "accumulation", "background")}
