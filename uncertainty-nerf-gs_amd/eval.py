"""Eval harness: the counterpart of scripts/eval_uncertainty.py for this build.

`get_average_uncertainty_metrics` walks (camera, ground-truth image) pairs, renders each camera with
the method's callable (`model.get_outputs_for_camera`, `get_outputs_for_camera_unc`, an ensemble
closure ...), computes the reference's per-image RGB metrics under the reference's key names
(eval_uncertainty.py:756-771: psnr, rgb_ause_{mse,mae,rmse}, rgb_mse, rgb_rmse, rgb_nll, rgb_avg_var,
rgb_auc_{abs_error,length,neg_error}) plus `num_rays_per_sec` / `fps` (:948-952), averages them over
the images (:1069-1077) and writes the same `metrics.json` envelope (:1156-1169).

Differences, on purpose: SSIM / LPIPS (torchmetrics / torchvision networks, absent here) are not
computed; no plots; and `num_rays_per_sec` is reported twice -- `num_rays_per_sec` covers render +
metrics like the reference's counter (so numbers stay comparable with its metrics.json), while
`render_rays_per_sec` times the render alone (HIP-synchronised).  Depth metrics (`depth_metrics_unc`,
eval_uncertainty.py:415-644) take the dataset's `depth_gt_XX.npy` map and `scale_parameters.txt` factor,
read by `load_depth_gt`; pass `depth_gt_fn` to `get_average_uncertainty_metrics` to include them.
"""
from __future__ import annotations

import json
import os
import time
from dataclasses import dataclass
from functools import partial
from pathlib import Path
from typing import Callable, Dict, Iterable, List, Optional, Tuple, Union

import numpy as np
import torch

from . import metrics as M


def image_metrics_unc(outputs: Dict[str, torch.Tensor], gt_image: torch.Tensor, eval_rgb_unc: bool = True,
                      min_rgb_std_for_nll: float = 3e-2, composite_gt: Optional[Callable] = None):
    """get_image_metrics_and_images_unc (eval_uncertainty.py:647-813), RGB part.
    -> (metrics_dict, curves) where curves carries the per-image sparsification / calibration curves
    that the reference accumulates for its test-set plots."""
    rgb = torch.clip(outputs["rgb"], max=1.0)
    image = gt_image.to(rgb.device)
    if "background" in outputs and composite_gt is not None:  # splatfacto: blend GT alpha with the background
        image = composite_gt(image, outputs["background"])
    # psnr / ssim as at eval_uncertainty.py:683-688; lpips needs the pretrained AlexNet weights (not available
    # offline) and is left out of the dict rather than faked
    md: Dict[str, float] = {"psnr": M.psnr(rgb, image), "ssim": M.ssim(rgb, image[..., :3])}
    curves: Dict[str, np.ndarray] = {}
    if eval_rgb_unc:
        rgb_std = outputs["rgb_std"]
        sq = torch.sum((rgb - image) ** 2, dim=-1).flatten()
        ab = torch.sum(torch.abs(rgb - image), dim=-1).flatten()
        var = (rgb_std ** 2).flatten()
        for et, err in (("mae", ab), ("mse", sq), ("rmse", sq)):
            _, e, ev, a = M.ause(var, err, et)
            md[f"rgb_ause_{et}"] = float(a)
            curves[f"rgb_all_ause_{et}"], curves[f"rgb_all_var_ause_{et}"] = e, ev
        md["rgb_mse"] = float(sq.mean().item())
        md["rgb_rmse"] = float(np.sqrt(sq.mean().item()))
        md["rgb_nll"] = float(M.negative_gaussian_loglikelihood(rgb.reshape(-1, 3), image.reshape(-1, 3), rgb_std,
                                                               eps=min_rgb_std_for_nll).mean().item())
        md["rgb_avg_var"] = float(var.mean().item())
        std3 = var.sqrt().unsqueeze(-1).repeat(1, 3)
        a = M.auce_torch(rgb.reshape(-1, 3), std3, image.reshape(-1, 3))   # = M.auce (reference loop), one sort on device
        md["rgb_auc_abs_error"], md["rgb_auc_length"] = a["auc_abs_error_values"], a["auc_length_values"]
        md["rgb_auc_neg_error"] = a["auc_neg_error_values"]
        for k in ("coverage_values", "avg_length_values", "coverage_error_values", "abs_coverage_error_values",
                  "neg_coverage_error_values"):
            curves[f"rgb_all_auce_{k}"] = a[k]
    return md, curves


def load_depth_gt(dataset_path: str, img_num: int) -> Tuple[np.ndarray, float]:
    """the two files get_unc_metrics_depth reads (eval_uncertainty.py:432-437): -> (depth_gt [H,W], scale a)"""
    a = float(np.loadtxt(os.path.join(str(dataset_path), "scale_parameters.txt"), delimiter=","))
    return np.load(os.path.join(str(dataset_path), "depth_gt_{:02d}.npy".format(img_num))), a


def depth_metrics_unc(outputs: Dict[str, torch.Tensor], depth_gt, scale: float, min_depth_std_for_nll: float = 1.0):
    """get_unc_metrics_depth (eval_uncertainty.py:415-644) without the plots, plus the key renaming of
    get_image_metrics_and_images_unc (:702-733).  depth / depth_std are resized to the GT map if the shapes
    differ (torchvision `resize` on a tensor = bilinear `interpolate`, no antialias), scaled by `scale`;
    NLL uses the prediction clipped to [1e-3, max GT] on the full image and is then masked by `GT > 0`;
    errors, AUSE and AUCE use the masked, clipped prediction.  -> (metrics_dict, curves)"""
    depth = outputs["depth"].squeeze(-1).to(torch.float32)
    depth_std = outputs["depth_std"].squeeze(-1).to(torch.float32)
    gt = torch.as_tensor(depth_gt, device=depth.device)

    def _fit(x):
        if gt.shape[-2:] == x.shape[-2:]:
            return x
        return torch.nn.functional.interpolate(x[None, None], size=tuple(gt.shape[-2:]), mode="bilinear",
                                               align_corners=False, antialias=False)[0, 0]

    depth, depth_std = _fit(depth), _fit(depth_std)
    lo, hi = 1e-3, gt.max().float()
    depth = scale * depth
    depth_std = scale * depth_std
    clipped = torch.minimum(torch.clamp_min(depth, lo), hi)
    nll_img = M.negative_gaussian_loglikelihood(clipped.unsqueeze(-1), gt.unsqueeze(-1), depth_std.unsqueeze(-1),
                                                eps=min_depth_std_for_nll).reshape(clipped.shape)
    mask = gt > 0
    d, g, sd = clipped[mask], gt[mask], depth_std[mask]
    sq, ab, var = (g - d) ** 2, (g - d).abs(), sd ** 2
    md: Dict[str, float] = {}
    curves: Dict[str, np.ndarray] = {}
    for et, err in (("mse", sq), ("mae", ab), ("rmse", sq)):
        _, e, ev, a = M.ause(var, err, et)
        md[f"depth_ause_{et}"] = float(a)
        curves[f"depth_all_ause_{et}"], curves[f"depth_all_var_ause_{et}"] = e, ev
    md["depth_mse"] = float(sq.mean().item())
    md["depth_rmse"] = float(np.sqrt(sq.mean().item()))
    md["depth_nll"] = float(nll_img[mask].mean().item())
    md["depth_avg_var"] = float(var.mean().item())
    a = M.auce_torch(d.flatten(), sd.flatten(), g.flatten())
    md["depth_auc_abs_error"], md["depth_auc_length"] = a["auc_abs_error_values"], a["auc_length_values"]
    md["depth_auc_neg_error"] = a["auc_neg_error_values"]
    for k in ("coverage_values", "avg_length_values", "coverage_error_values", "abs_coverage_error_values",
              "neg_coverage_error_values"):
        curves[f"depth_all_auce_{k}"] = a[k]
    return md, curves


def get_average_uncertainty_metrics(get_outputs_for_camera: Callable, eval_set: Iterable[Tuple[object, torch.Tensor]],
                                    eval_rgb_unc: bool = True, min_rgb_std_for_nll: float = 3e-2,
                                    composite_gt: Optional[Callable] = None, depth_gt_fn: Optional[Callable] = None,
                                    min_depth_std_for_nll: float = 1.0):
    """eval_uncertainty.py:816-1079.  -> (averaged metrics dict, averaged curves dict).
    depth_gt_fn(image_index) -> (depth_gt [H,W], scale) switches the depth metrics on (eval_depth_unc)."""
    rows: List[Dict[str, float]] = []
    sums: Dict[str, np.ndarray] = {}
    for img_num, (camera, gt) in enumerate(eval_set):
        inner_start = time.time()
        outputs = get_outputs_for_camera(camera)
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        render_s = time.time() - inner_start
        H, W = outputs["rgb"].shape[:2]
        md, curves = image_metrics_unc(outputs, gt, eval_rgb_unc, min_rgb_std_for_nll, composite_gt)
        if depth_gt_fn is not None:
            dgt, scale = depth_gt_fn(img_num)
            dmd, dcurves = depth_metrics_unc(outputs, dgt, scale, min_depth_std_for_nll)
            md.update(dmd)
            curves.update(dcurves)
        md["num_rays_per_sec"] = H * W / (time.time() - inner_start)
        md["fps"] = md["num_rays_per_sec"] / (H * W)
        md["render_rays_per_sec"] = H * W / render_s
        rows.append(md)
        for k, v in curves.items():
            sums[k] = sums.get(k, 0) + np.asarray(v, dtype=np.float64)
    avg = {k: float(torch.mean(torch.tensor([r[k] for r in rows], dtype=torch.float64))) for k in rows[0]}
    return avg, {k: v / len(rows) for k, v in sums.items()}


def write_metrics_json(path: str, experiment_name: str, method_name: str, checkpoint: str, results: Dict[str, float]):
    """the envelope of eval_uncertainty.py:1156-1169"""
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "w", encoding="utf8") as f:
        json.dump({"experiment_name": experiment_name, "method_name": method_name, "checkpoint": checkpoint,
                   "results": results}, f, indent=2)


# ---- the eval script's configuration surface (scripts/eval_configs.py) and its per-method dispatch ----------

@dataclass
class EvalUncertainty:
    """scripts/eval_configs.py:7-49 (field names and defaults; pinned by tests/golden/eval_configs.json)"""
    load_config: Union[Path, List[Path], None] = None
    dataset_path: Optional[Path] = None
    output_path: Path = Path("output.json")
    render_output_path: Optional[Path] = None
    save_all_ause: bool = False
    seed: int = 42
    eval_depth: bool = True
    eval_rgb: bool = True
    plot_ause: bool = False
    save_rendered_images: bool = False
    min_rgb_std_for_nll: float = 3e-2
    min_depth_std_for_nll: float = 2.0
    unc_max: float = 1.0
    unc_min: float = 0.0


@dataclass
class LaplaceConfig(EvalUncertainty):
    prior_precision: float = 1.0
    n_samples: int = 100
    n_iters: int = 300
    use_deterministic_density: bool = False


@dataclass
class EnsembleConfig(EvalUncertainty):
    pass


@dataclass
class MCDropoutConfig(EvalUncertainty):
    mc_samples: Optional[int] = None


@dataclass
class ActiveNerfactoConfig(EvalUncertainty):
    eval_depth: bool = True


@dataclass
class ActiveSplatfactoConfig(EvalUncertainty):
    eval_depth: bool = False


EvalConfigs = Union[LaplaceConfig, EnsembleConfig, MCDropoutConfig, ActiveNerfactoConfig, ActiveSplatfactoConfig]


def outputs_fn_for(eval_config: EvalConfigs, model, ggn_batches=None, pipeline=None) -> Callable:
    """The per-method callable scripts/eval_uncertainty.py:1086-1134 selects, for this build's Model mirrors.
    `model` is one Model, or a list of member Models for EnsembleConfig (single process; one member per rank goes
    through ensemble.aggregate_distributed instead).  LaplaceConfig loads `ggn_{n_iters}.pt` next to load_config when
    it exists, else fits the GGN (from `pipeline.datamanager` or `ggn_batches`) and saves it there (:1103-1116)."""
    import torch
    if isinstance(eval_config, EnsembleConfig):
        from . import ensemble
        if hasattr(model, "get_ensemble_outputs_for_camera_ray_bundle"):   # an EnsemblePipeline (eval_uncertainty.py:1127)
            return model.get_ensemble_outputs_for_camera_ray_bundle
        return ensemble.EnsemblePipeline(list(model)).get_ensemble_outputs_for_camera_ray_bundle
    if isinstance(eval_config, MCDropoutConfig):
        model.config.mc_samples = eval_config.mc_samples if eval_config.mc_samples is not None else model.config.mc_samples
        model.invalidate()
        return model.get_outputs_for_camera
    if isinstance(eval_config, LaplaceConfig):
        hessian_path = None
        if eval_config.load_config is not None:
            hessian_path = Path(eval_config.load_config).parent / f"ggn_{eval_config.n_iters}.pt"
        if hessian_path is not None and hessian_path.exists():
            saved = torch.load(hessian_path)
            model.field.mlp_density_ggn, model.field.mlp_rgb_ggn = saved["mlp_density_ggn"], saved["mlp_rgb_ggn"]
        else:
            model.compute_hessian_naive(pipeline=pipeline, n_iters=eval_config.n_iters, ray_batches=ggn_batches)
            if hessian_path is not None:
                hessian_path.parent.mkdir(parents=True, exist_ok=True)
                torch.save({"mlp_density_ggn": model.field.mlp_density_ggn.cpu(),
                            "mlp_rgb_ggn": model.field.mlp_rgb_ggn.cpu()}, hessian_path)
        model.prior_prec = eval_config.prior_precision
        return partial(model.get_outputs_for_camera_unc, is_inference=True,
                       use_deterministic_density=eval_config.use_deterministic_density,
                       prior_prec=eval_config.prior_precision, n_samples=eval_config.n_samples)
    return model.get_outputs_for_camera          # ActiveNerfactoConfig, ActiveSplatfactoConfig


def run_eval(eval_config: EvalConfigs, model, eval_set, experiment_name: str = "", method_name: str = "",
             checkpoint: str = "", depth_gt_fn: Optional[Callable] = None, composite_gt: Optional[Callable] = None,
             **fn_kw) -> Dict[str, float]:
    """main() of scripts/eval_uncertainty.py:1082-1169 without nerfstudio's pipeline loading: pick the method's
    callable, average the per-image metrics, write the metrics.json envelope to eval_config.output_path."""
    fn = outputs_fn_for(eval_config, model, **fn_kw)
    if composite_gt is None and hasattr(model, "composite_gt"):   # splat models: GT alpha over the background
        composite_gt = model.composite_gt
    if eval_config.eval_depth and depth_gt_fn is None and eval_config.dataset_path is not None:
        depth_gt_fn = lambda i: load_depth_gt(str(eval_config.dataset_path), i)
    metrics, _curves = get_average_uncertainty_metrics(
        fn, eval_set, eval_rgb_unc=eval_config.eval_rgb, min_rgb_std_for_nll=eval_config.min_rgb_std_for_nll,
        composite_gt=composite_gt, depth_gt_fn=depth_gt_fn if eval_config.eval_depth else None,
        min_depth_std_for_nll=eval_config.min_depth_std_for_nll)
    write_metrics_json(str(eval_config.output_path), experiment_name, method_name, checkpoint, metrics)
    return metrics
