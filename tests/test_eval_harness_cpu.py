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
