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
    want = {"psnr", "ssim", "rgb_ause_mse", "rgb_ause_mae", "rgb_ause_rmse", "rgb_mse", "rgb_rmse", "rgb_nll", "rgb_avg_var",
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


def test_eval_config_mirror_matches_reference_defaults():
    """field names, order and defaults of the eval-script dataclasses vs tests/golden/eval_configs.json
    (generated from the reference's scripts/eval_configs.py by tests/golden/make_golden.py)"""
    import dataclasses
    import os
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "eval_configs.json")))
    for name, fields in gold.items():
        cls = getattr(E, name)
        ours = dataclasses.fields(cls)
        assert [f.name for f in ours] == list(fields), name
        for f in ours:
            want = fields[f.name]
            if want["required"]:
                continue          # the reference has no default; the mirror uses None so it can be built in tests
            got = f.default
            got = str(got) if got is not None and not isinstance(got, (bool, int, float)) else got
            assert got == want["default"], (name, f.name, got, want["default"])


def test_outputs_fn_dispatch_follows_the_eval_script(tmp_path):
    """eval_uncertainty.py:1086-1134 with fake models: which callable each config selects and what it sets"""
    calls = []

    class FakeField:
        mlp_density_ggn = None
        mlp_rgb_ggn = None

    class FakeModel:
        def __init__(self):
            self.config = type("C", (), {"mc_samples": 10})()
            self.field = FakeField()

        def invalidate(self):
            calls.append("invalidate")

        def get_outputs_for_camera(self, camera):
            return {"who": "plain", "camera": camera}

        def get_outputs_for_camera_unc(self, camera, **kw):
            return {"who": "unc", "kw": kw}

        def compute_hessian_naive(self, pipeline=None, n_iters=1000, ray_batches=None):
            calls.append(("ggn", n_iters))
            self.field.mlp_density_ggn, self.field.mlp_rgb_ggn = torch.ones(65), torch.ones(195)

    m = FakeModel()
    fn = E.outputs_fn_for(E.MCDropoutConfig(mc_samples=8), m)
    assert m.config.mc_samples == 8 and fn("cam")["who"] == "plain" and calls == ["invalidate"]
    m2 = FakeModel()
    E.outputs_fn_for(E.MCDropoutConfig(), m2)
    assert m2.config.mc_samples == 10            # None keeps the model's own setting
    cfg = E.LaplaceConfig(load_config=tmp_path / "cfg" / "config.yml", n_iters=7, prior_precision=2.5, n_samples=50)
    m3 = FakeModel()
    fn = E.outputs_fn_for(cfg, m3)
    out = fn("cam")
    assert ("ggn", 7) in calls and (tmp_path / "cfg" / "ggn_7.pt").exists() and m3.prior_prec == 2.5
    assert out["who"] == "unc" and out["kw"] == dict(is_inference=True, use_deterministic_density=False, prior_prec=2.5,
                                                     n_samples=50)
    calls.clear()
    m4 = FakeModel()
    E.outputs_fn_for(cfg, m4)                      # second time: the saved GGN is loaded instead of refitted
    assert calls == [] and torch.equal(m4.field.mlp_rgb_ggn, torch.ones(195))
    assert E.outputs_fn_for(E.ActiveNerfactoConfig(), m)("c")["who"] == "plain"
    assert E.ActiveSplatfactoConfig().eval_depth is False and E.EvalUncertainty().min_depth_std_for_nll == 2.0


def test_run_eval_writes_the_metrics_envelope(tmp_path):
    items = _fake_eval_set(2, 12, 16)

    class M:
        def get_outputs_for_camera(self, cam):
            return cam
    cfg = E.ActiveNerfactoConfig(output_path=tmp_path / "o" / "metrics.json", eval_depth=False)
    res = E.run_eval(cfg, M(), [(o, gt) for o, gt in items], experiment_name="garden", method_name="active-nerfacto",
                     checkpoint="step-000029999.ckpt")
    d = json.loads((tmp_path / "o" / "metrics.json").read_text())
    assert d["method_name"] == "active-nerfacto" and d["results"]["psnr"] == res["psnr"] and "rgb_ause_mse" in res


def _ssim_scipy(pred, gt):
    """independent restatement of the torchmetrics recipe with scipy filters ('mirror' = torch's reflect padding)"""
    from scipy.ndimage import correlate1d
    p, t = pred.double().numpy(), gt.double().numpy()            # [H,W,C]
    dr = max(p.max() - p.min(), t.max() - t.min())
    c1, c2 = (0.01 * dr) ** 2, (0.03 * dr) ** 2
    d = np.arange(-5, 6, dtype=np.float64)
    g = np.exp(-(d / 1.5) ** 2 / 2)
    g /= g.sum()
    blur = lambda x: correlate1d(correlate1d(x, g, axis=0, mode="mirror"), g, axis=1, mode="mirror")
    mu_p, mu_t = blur(p), blur(t)
    s_pp, s_tt, s_pt = blur(p * p) - mu_p ** 2, blur(t * t) - mu_t ** 2, blur(p * t) - mu_p * mu_t
    m = ((2 * mu_p * mu_t + c1) * (2 * s_pt + c2)) / ((mu_p ** 2 + mu_t ** 2 + c1) * (s_pp + s_tt + c2))
    return m[5:-5, 5:-5].mean()


def test_ssim_follows_the_torchmetrics_recipe():
    g = torch.Generator().manual_seed(4)
    gt = torch.rand(40, 52, 3, generator=g)
    gt = torch.nn.functional.avg_pool2d(gt.permute(2, 0, 1)[None], 5, 1, 2)[0].permute(1, 2, 0)   # some structure
    pred = torch.clamp(gt + 0.05 * torch.randn(gt.shape, generator=g), 0, 1)
    assert abs(M.ssim(gt, gt) - 1.0) < 1e-6
    v = M.ssim(pred, gt)
    assert 0.0 < v < 1.0 and abs(v - _ssim_scipy(pred, gt)) < 2e-5
    assert abs(M.ssim(pred.permute(2, 0, 1)[None], gt.permute(2, 0, 1)[None]) - v) < 1e-7     # [1,C,H,W] form
    assert M.ssim(torch.clamp(gt + 0.2 * torch.randn(gt.shape, generator=g), 0, 1), gt) < v   # more noise, lower score


def test_splat_ground_truth_composition():
    """eval_uncertainty.py:321-322: RGBA ground truth is blended over the render's background colour"""
    from uncertainty_nerf_gs_amd import models as Mo
    m = Mo.ActiveSplatfactoModel(Mo.ActiveSplatfactoModelConfig(), num_points=4)
    rgba = torch.zeros(2, 3, 4, dtype=torch.uint8)
    rgba[..., 0] = 255
    rgba[0, :, 3] = 255          # first row opaque red, second row transparent
    bg = torch.tensor([0.0, 0.0, 1.0])
    out = m.composite_gt(rgba, bg)
    assert out.shape == (2, 3, 3) and out.dtype == torch.float32
    assert torch.equal(out[0], torch.tensor([1.0, 0.0, 0.0]).expand(3, 3)) and torch.equal(out[1], bg.expand(3, 3))
    rgb = torch.rand(2, 3, 3)
    assert torch.equal(m.composite_gt(rgb, bg), rgb)      # no alpha channel: unchanged
    items = [({"rgb": torch.rand(16, 16, 3), "rgb_std": torch.rand(16, 16, 1) + 0.01, "background": bg},
              torch.cat([torch.rand(16, 16, 3), torch.ones(16, 16, 1)], -1))]
    md, _ = E.image_metrics_unc(items[0][0], items[0][1], composite_gt=m.composite_gt)
    assert np.isfinite(md["psnr"]) and np.isfinite(md["ssim"])
