"""Host logic of the eval harness on CPU tensors (fake renderer): key names and averaging follow
scripts/eval_uncertainty.py; AUSE / AUCE values go through the reference-pinned metrics."""
import json

import numpy as np
import torch

from uncertainty_nerf_gs_amd import eval as E
from uncertainty_nerf_gs_amd import metrics as M


def _fake_eval_set(n=3, H=20, W=24):
    g = torch.Generator().manual_seed(0)
    items = []
    for i in range(n):
        gt = torch.rand(H, W, 3, generator=g)
        std = 0.02 + 0.1 * torch.rand(H, W, 1, generator=g)
        rgb = torch.clamp(gt + std * torch.randn(H, W, 3, generator=g), 0, 1.2)
        items.append(({"rgb": rgb, "rgb_std": std, "accumulation": torch.ones(H, W, 1)}, gt))
    return items


def test_metric_keys_and_averaging(tmp_path):
    items = _fake_eval_set()
    avg, curves = E.get_average_uncertainty_metrics(lambda cam: cam, [(o, gt) for o, gt in items])
    want = {"psnr", "rgb_ause_mse", "rgb_ause_mae", "rgb_ause_rmse", "rgb_mse", "rgb_rmse", "rgb_nll", "rgb_avg_var",
            "rgb_auc_abs_error", "rgb_auc_length", "rgb_auc_neg_error", "num_rays_per_sec", "fps", "render_rays_per_sec"}
    assert set(avg) == want
    per = [E.image_metrics_unc(o, gt)[0] for o, gt in items]
    for k in ("psnr", "rgb_ause_mse", "rgb_nll", "rgb_auc_abs_error"):
        assert abs(avg[k] - np.mean([p[k] for p in per])) < 1e-12
    assert curves["rgb_all_ause_mse"].shape == (100,) and curves["rgb_all_auce_coverage_values"].shape == (99,)
    # rgb is clipped to <= 1 before the metrics (eval_uncertainty.py:681)
    o, gt = items[0]
    assert abs(per[0]["psnr"] - M.psnr(torch.clip(o["rgb"], max=1.0), gt)) < 1e-12
    p = tmp_path / "out" / "metrics.json"
    E.write_metrics_json(str(p), "exp", "active-nerfacto", "step-000029999.ckpt", avg)
    d = json.loads(p.read_text())
    assert list(d) == ["experiment_name", "method_name", "checkpoint", "results"] and d["results"]["psnr"] == avg["psnr"]


def test_calibrated_uncertainty_scores_better_than_shuffled():
    (o, gt), = _fake_eval_set(1, 48, 48)
    good, _ = E.image_metrics_unc(o, gt)
    perm = torch.randperm(48 * 48, generator=torch.Generator().manual_seed(1))
    bad_out = dict(o, rgb_std=o["rgb_std"].reshape(-1, 1)[perm].reshape(48, 48, 1))
    bad, _ = E.image_metrics_unc(bad_out, gt)
    assert good["rgb_ause_mse"] < bad["rgb_ause_mse"] and good["rgb_nll"] < bad["rgb_nll"]


def _depth_case(H=18, W=22, gh=18, gw=22, seed=3):
    g = torch.Generator().manual_seed(seed)
    gt = 1.0 + 4.0 * torch.rand(gh, gw, generator=g)
    gt[0, :5] = 0.0                                  # invalid GT pixels (masked out, eval_uncertainty.py:538)
    base = torch.nn.functional.interpolate(gt[None, None], size=(H, W), mode="bilinear", align_corners=False)[0, 0]
    std = 0.5 + 2.0 * torch.rand(H, W, generator=g)
    depth = (base + std * torch.randn(H, W, generator=g)) / 2.5
    depth[3, 3] = -1.0                               # clipped up to MIN_DEPTH
    depth[4, 4] = 100.0                              # clipped down to max GT
    return {"depth": depth[..., None], "depth_std": (std / 2.5)[..., None]}, gt.numpy(), 2.5


def test_depth_metrics_follow_the_reference_recipe(tmp_path):
    """numpy restatement of get_unc_metrics_depth (eval_uncertainty.py:415-644): scale, clip to
    [1e-3, max GT], NLL on the clipped full image then masked, errors / AUSE / AUCE on GT > 0."""
    out, gt, a = _depth_case()
    md, curves = E.depth_metrics_unc(out, gt, a, min_depth_std_for_nll=1.0)
    d = a * out["depth"][..., 0].double().numpy()
    s = a * out["depth_std"][..., 0].double().numpy()
    dc = np.clip(d, 1e-3, gt.max())
    se = np.maximum(s, 1.0)
    nll = (gt - dc) ** 2 / (2 * se ** 2) + np.log(se) + 0.5 * np.log(2 * np.pi)
    m = gt > 0
    assert abs(md["depth_nll"] - nll[m].mean()) < 1e-5
    assert abs(md["depth_mse"] - ((gt - dc)[m] ** 2).mean()) < 1e-5 * md["depth_mse"]
    assert abs(md["depth_rmse"] - np.sqrt(md["depth_mse"])) < 1e-12
    assert abs(md["depth_avg_var"] - (s[m] ** 2).mean()) < 1e-5 * md["depth_avg_var"]
    _, _, _, ause_mae = M.ause(torch.tensor(s[m] ** 2, dtype=torch.float32), torch.tensor(np.abs(gt - dc)[m], dtype=torch.float32), "mae")
    assert abs(md["depth_ause_mae"] - ause_mae) < 1e-6
    assert set(md) == {"depth_ause_mse", "depth_ause_mae", "depth_ause_rmse", "depth_mse", "depth_rmse", "depth_nll",
                       "depth_avg_var", "depth_auc_abs_error", "depth_auc_length", "depth_auc_neg_error"}
    assert curves["depth_all_var_ause_rmse"].shape == (100,) and curves["depth_all_auce_coverage_values"].shape == (99,)
    # the dataset files the reference reads, and the harness switch
    np.save(tmp_path / "depth_gt_00.npy", gt)
    (tmp_path / "scale_parameters.txt").write_text(f"{a}\n")
    gt2, a2 = E.load_depth_gt(str(tmp_path), 0)
    assert np.array_equal(gt2, gt) and a2 == a
    rgb_items = _fake_eval_set(1, 18, 22)
    outputs = dict(rgb_items[0][0], **out)
    avg, _ = E.get_average_uncertainty_metrics(lambda cam: cam, [(outputs, rgb_items[0][1])],
                                               depth_gt_fn=lambda i: E.load_depth_gt(str(tmp_path), i))
    assert abs(avg["depth_nll"] - md["depth_nll"]) < 1e-12 and "rgb_ause_mse" in avg


def test_depth_metrics_resize_prediction_to_gt_shape():
    """splatfacto renders can be a pixel smaller than the GT map: the prediction is resized (:441-451)"""
    out, gt, a = _depth_case(H=17, W=21, gh=18, gw=22)
    md, _ = E.depth_metrics_unc(out, gt, a)
    assert np.isfinite(list(md.values())).all()
