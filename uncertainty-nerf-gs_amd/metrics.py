"""Parity metrics: PSNR, AUSE, AUCE, Gaussian NLL.

Host-side mirror of nerfuncertainty/metrics/{ause,auce}.py and of the error definitions in
scripts/eval_uncertainty.py:306-412.  Unlike the reference (100 Python-loop slices + numpy on the
CPU, seconds per 1080p image), AUSE here is one sort + one prefix sum on whatever device the
tensors live on; results agree with the reference implementation to ~1e-7 (golden-vector test).
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import numpy as np
import torch

_RATIOS = np.linspace(0, 1, 100, endpoint=False)


def _trapz(y: np.ndarray, x: np.ndarray) -> float:
    y = np.asarray(y, dtype=np.float64)
    x = np.asarray(x, dtype=np.float64)
    return float(np.sum((y[1:] + y[:-1]) * np.diff(x) / 2.0))


def _sparsification_curve(err_sorted: torch.Tensor, err_type: str) -> np.ndarray:
    n = err_sorted.numel()
    keep = torch.tensor([int((1 - r) * n) for r in _RATIOS], device=err_sorted.device, dtype=torch.long)
    csum = torch.cumsum(err_sorted.to(torch.float64), dim=0)
    means = csum[(keep - 1).clamp(min=0)] / keep.clamp(min=1).to(torch.float64)
    means = torch.where(keep > 0, means, torch.full_like(means, float("nan")))
    if err_type == "rmse":
        means = means.sqrt()
    return means.to(torch.float32).cpu().numpy()


def ause(unc_vec: torch.Tensor, err_vec: torch.Tensor, err_type: str = "rmse"):
    """metrics/ause.py:7-44.  -> (ratio_removed, oracle_curve, by_variance_curve, ause)"""
    assert err_type in ("rmse", "mae", "mse")
    err_sorted, _ = torch.sort(err_vec)
    oracle = _sparsification_curve(err_sorted, err_type)
    _, order = torch.sort(unc_vec)
    by_var = _sparsification_curve(err_vec[order], err_type).astype(np.float64)
    max_val = max(float(oracle.max()), float(by_var.max()))
    oracle = oracle / np.float32(max_val)
    by_var = by_var / max_val
    return _RATIOS, oracle, by_var, _trapz(by_var - oracle, _RATIOS)


def _norm_ppf(p: np.ndarray) -> np.ndarray:
    from scipy.stats import norm  # the reference uses scipy.stats.norm.ppf (auce.py:21-22)
    return norm.ppf(p)


def auce(mean_values: np.ndarray, sigma_values: np.ndarray, target_values: np.ndarray) -> Dict[str, np.ndarray]:
    """metrics/auce.py:10-57 (99 two-sided Gaussian intervals, alpha = 0.01..0.99)."""
    n = float(np.prod(target_values.shape))
    alphas = np.arange(start=0.01, stop=1.0, step=0.01)
    z = _norm_ppf(1.0 - alphas / 2)
    dev = np.abs(target_values - mean_values)
    coverage, length = [], []
    for zi in z:
        lo = mean_values - zi * sigma_values
        hi = mean_values + zi * sigma_values
        coverage.append(np.count_nonzero(np.logical_and(target_values >= lo, target_values <= hi)) / n)
        length.append(np.mean(hi - lo))
    coverage = np.array(coverage)
    length = np.array(length)
    cov_err = coverage - (1.0 - alphas)
    abs_err = np.abs(cov_err)
    neg_err = (np.abs(cov_err) - cov_err) / 2.0
    return {
        "coverage_values": coverage, "avg_length_values": length, "coverage_error_values": cov_err,
        "abs_coverage_error_values": abs_err, "neg_coverage_error_values": neg_err,
        "auc_abs_error_values": _trapz(abs_err, alphas), "auc_length_values": _trapz(length, alphas),
        "auc_neg_error_values": _trapz(neg_err, alphas),
    }


def auce_torch(mean_values: torch.Tensor, sigma_values: torch.Tensor, target_values: torch.Tensor) -> Dict[str, np.ndarray]:
    """`auce` on whatever device the tensors live on, in one sort: an interval [m - z s, m + z s] covers the
    target iff |t - m| / s <= z, so the 99 coverages are 99 binary searches into the sorted standardised residuals
    and the 99 mean interval lengths are 2 z mean(s).  Same dictionary as `auce` (the reference's 99-pass numpy
    loop, metrics/auce.py:10-57, takes ~1 s per 1080p image and would dominate the eval loop's rays/s); float64
    throughout, so the coverage counts agree with it except for residuals within 1e-16 of an interval edge."""
    m, sg, t = (x.detach().reshape(-1).to(torch.float64) for x in (mean_values, sigma_values, target_values))
    n = float(t.numel())
    alphas = np.arange(start=0.01, stop=1.0, step=0.01)
    z = torch.as_tensor(_norm_ppf(1.0 - alphas / 2), dtype=torch.float64, device=t.device)
    r = (t - m).abs()
    ratio = torch.where(sg > 0, r / sg, torch.where(r == 0, torch.zeros_like(r), torch.full_like(r, float("inf"))))
    ratio, _ = torch.sort(ratio)
    coverage = (torch.searchsorted(ratio, z, right=True).to(torch.float64) / n).cpu().numpy()
    length = (2.0 * z * sg.mean()).cpu().numpy()
    cov_err = coverage - (1.0 - alphas)
    abs_err = np.abs(cov_err)
    neg_err = (np.abs(cov_err) - cov_err) / 2.0
    return {
        "coverage_values": coverage, "avg_length_values": length, "coverage_error_values": cov_err,
        "abs_coverage_error_values": abs_err, "neg_coverage_error_values": neg_err,
        "auc_abs_error_values": _trapz(abs_err, alphas), "auc_length_values": _trapz(length, alphas),
        "auc_neg_error_values": _trapz(neg_err, alphas),
    }


def psnr(pred: torch.Tensor, gt: torch.Tensor) -> float:
    """torchmetrics PeakSignalNoiseRatio(data_range=1.0) as used via model.psnr
    (scripts/eval_uncertainty.py:683): 10*log10(1/mse) over all elements."""
    mse = torch.mean((pred.to(torch.float64) - gt.to(torch.float64)) ** 2).item()
    return 10.0 * math.log10(1.0 / mse)


def ssim(pred: torch.Tensor, gt: torch.Tensor, data_range=None) -> float:
    """torchmetrics.functional.structural_similarity_index_measure with its defaults, as reached through
    model.ssim (scripts/eval_uncertainty.py:684) on [1,3,H,W] images.  [UPSTREAM-RECALL, torchmetrics is absent
    here]: 11x11 gaussian window, sigma 1.5, k1 0.01, k2 0.03; inputs reflect-padded by 5, depthwise filtered
    (no further padding), the padded border cropped again, mean over everything; data_range=None means
    max(pred.max() - pred.min(), gt.max() - gt.min()).  Runs on the images' device."""
    p, t = pred.to(torch.float32), gt.to(torch.float32)
    if p.dim() == 3:  # [H,W,C] -> [1,C,H,W]
        p, t = p.permute(2, 0, 1)[None], t.permute(2, 0, 1)[None]
    if data_range is None:
        data_range = max(float(p.max() - p.min()), float(t.max() - t.min()))
    c1, c2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    ks, sigma, pad = 11, 1.5, 5
    d = torch.arange((1 - ks) / 2, (1 + ks) / 2, 1, dtype=p.dtype, device=p.device)
    g = torch.exp(-((d / sigma) ** 2) / 2)
    g = (g / g.sum())[None]
    C = p.shape[1]
    kernel = (g.t() @ g).expand(C, 1, ks, ks)
    p = torch.nn.functional.pad(p, (pad, pad, pad, pad), mode="reflect")
    t = torch.nn.functional.pad(t, (pad, pad, pad, pad), mode="reflect")
    stack = torch.cat((p, t, p * p, t * t, p * t))
    out = torch.nn.functional.conv2d(stack, kernel, groups=C)
    B = pred.shape[0] if pred.dim() == 4 else 1
    mu_p, mu_t, e_pp, e_tt, e_pt = out.split(B)
    s_pp, s_tt, s_pt = e_pp - mu_p ** 2, e_tt - mu_t ** 2, e_pt - mu_p * mu_t
    upper, lower = 2 * s_pt + c2, s_pp + s_tt + c2
    idx = ((2 * mu_p * mu_t + c1) * upper) / ((mu_p ** 2 + mu_t ** 2 + c1) * lower)
    return float(idx[..., pad:-pad, pad:-pad].mean().item())


def negative_gaussian_loglikelihood(preds: torch.Tensor, targets: torch.Tensor, stds: torch.Tensor,
                                    eps: float = 1e-6) -> torch.Tensor:
    """scripts/eval_uncertainty.py:404-412"""
    s = torch.clamp_min(stds.reshape(-1, 1), eps)
    c = preds.shape[-1]
    p, t = preds.reshape(-1, c), targets.reshape(-1, c)
    return ((t - p) ** 2) / (2 * s ** 2) + torch.log(s) + 0.5 * math.log(2 * math.pi)


def rgb_uncertainty_metrics(rgb_pred: torch.Tensor, rgb_std: torch.Tensor, rgb_gt: torch.Tensor,
                            min_rgb_std_for_nll: float = 3e-2) -> Dict[str, float]:
    """scripts/eval_uncertainty.py:306-402 without the plotting: error definitions, the three
    AUSE variants, NLL and AUCE for one image [H,W,3] / [H,W,1]."""
    sq = torch.sum((rgb_pred - rgb_gt) ** 2, dim=-1).flatten()
    ab = torch.sum(torch.abs(rgb_pred - rgb_gt), dim=-1).flatten()
    var = (rgb_std ** 2).flatten()
    out = {"avg_var": var.mean().item(), "psnr": psnr(rgb_pred, rgb_gt)}
    out["ause_mae"] = ause(var, ab, "mae")[3]
    out["ause_mse"] = ause(var, sq, "mse")[3]
    out["ause_rmse"] = ause(var, sq, "rmse")[3]
    out["nll_rgb"] = negative_gaussian_loglikelihood(rgb_pred.reshape(-1, 3), rgb_gt.reshape(-1, 3), rgb_std,
                                                     eps=min_rgb_std_for_nll).mean().item()
    std3 = var.sqrt().unsqueeze(-1).repeat(1, 3)
    a = auce(rgb_pred.reshape(-1, 3).cpu().numpy(), std3.cpu().numpy(), rgb_gt.reshape(-1, 3).cpu().numpy())
    out.update({k: v for k, v in a.items() if k.startswith("auc_")})
    return out
