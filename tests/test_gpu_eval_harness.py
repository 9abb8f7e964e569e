"""SURVEY.md 8(f) rank 4 on the GPU: `eval.run_eval` per method on HIP renders of a small synthetic eval set
(device tensors all the way: psnr / ssim / AUSE / NLL / one-sort AUCE run on the render's device), against the CPU
ORACLE's render of the same cameras pushed through the reference-pinned numpy `metrics.ause` / `metrics.auce`
(scripts/eval_uncertainty.py:647-813, 1082-1169; metrics/ause.py, metrics/auce.py)."""
import json

import numpy as np
import pytest
import torch

from oracle import nerf_oracle as O

pytestmark = pytest.mark.gpu

H, W = 36, 48


def _cams(n):
    from uncertainty_nerf_gs_amd import models, synthetic
    return [models.Camera(synthetic.orbit_c2w(0.5 + 1.3 * i), 0.9 * W, 0.9 * W, W / 2, H / 2, H, W) for i in range(n)]


def _gt(ref_rgb, seed):
    g = torch.Generator().manual_seed(seed)
    noise = torch.randn(ref_rgb.shape, generator=g) * 0.05 * (0.3 + torch.rand(ref_rgb.shape[:2] + (1,), generator=g))
    return torch.clamp(ref_rgb + noise, 0, 1)


def _cpu_reference_metrics(out, gt):
    """the reference's metric definitions (eval_uncertainty.py:676-760) with the numpy ause / auce loops"""
    from uncertainty_nerf_gs_amd import metrics as M
    rgb = torch.clip(out["rgb"], max=1.0)
    sq = torch.sum((rgb - gt) ** 2, -1).flatten()
    ab = torch.sum((rgb - gt).abs(), -1).flatten()
    var = (out["rgb_std"] ** 2).flatten()
    md = {"psnr": M.psnr(rgb, gt), "rgb_mse": float(sq.mean()), "rgb_avg_var": float(var.mean())}
    for et, err in (("mae", ab), ("mse", sq), ("rmse", sq)):
        md[f"rgb_ause_{et}"] = float(M.ause(var, err, et)[3])
    std3 = var.sqrt().unsqueeze(-1).repeat(1, 3)
    a = M.auce(rgb.reshape(-1, 3).numpy(), std3.numpy(), gt.reshape(-1, 3).numpy())
    md["rgb_auc_abs_error"], md["rgb_auc_length"] = float(a["auc_abs_error_values"]), float(a["auc_length_values"])
    md["rgb_nll"] = float(M.negative_gaussian_loglikelihood(rgb.reshape(-1, 3), gt.reshape(-1, 3), out["rgb_std"].reshape(-1, 1),
                                                           eps=3e-2).mean())
    return md


def _check(got, refs, tol_ause=1e-3):
    keys = ("psnr", "rgb_mse", "rgb_avg_var", "rgb_ause_mae", "rgb_ause_mse", "rgb_ause_rmse", "rgb_auc_abs_error",
            "rgb_auc_length", "rgb_nll")
    avg = {k: float(np.mean([r[k] for r in refs])) for k in keys}
    assert abs(got["psnr"] - avg["psnr"]) <= 1e-4, (got["psnr"], avg["psnr"])            # the north-star gates
    for k in ("rgb_ause_mae", "rgb_ause_mse", "rgb_ause_rmse"):
        assert abs(got[k] - avg[k]) <= tol_ause, (k, got[k], avg[k])
    for k in ("rgb_mse", "rgb_avg_var", "rgb_nll", "rgb_auc_length"):
        assert abs(got[k] - avg[k]) <= 2e-3 * abs(avg[k]) + 1e-7, (k, got[k], avg[k])
    assert abs(got["rgb_auc_abs_error"] - avg["rgb_auc_abs_error"]) <= 2e-3, (got["rgb_auc_abs_error"], avg["rgb_auc_abs_error"])
    assert got["render_rays_per_sec"] > 0 and got["fps"] > 0


@pytest.mark.parametrize("kind", ["active", "mcdropout"])
def test_run_eval_on_hip_renders_matches_oracle_render_through_reference_metrics(dev, tmp_path, kind):
    from uncertainty_nerf_gs_amd import eval as E
    from uncertainty_nerf_gs_amd import models, synthetic
    import test_gpu_models as TM
    t = synthetic.make_scene_tensors(seed=21, kind=kind, log2T=14, prop_log2T=12)
    sc = O.scene_from_tensors(t)
    K, seed = 8, 0
    if kind == "active":
        cfg, ecfg = TM._small_cfg(models.ActiveNerfactoModelConfig(average_init_density=0.01)), E.ActiveNerfactoConfig(load_config=None, output_path=tmp_path / "m.json", eval_depth=False)
    else:
        cfg = TM._small_cfg(models.NerfactoMCDropoutModelConfig(average_init_density=0.01, mc_samples=3))
        ecfg = E.MCDropoutConfig(load_config=None, output_path=tmp_path / "m.json", eval_depth=False, mc_samples=K)
    model = cfg._target(cfg, num_train_data=4)
    model.load_state_dict(TM._state_dict_from_tensors(t, kind))
    model = model.to(dev)
    cams = _cams(3)
    refs, eval_set = [], []
    for i, cam in enumerate(cams):
        o, d, _ = O.generate_rays(cam.camera_to_worlds, cam.fx, cam.fy, cam.cx, cam.cy, H, W)
        if kind == "active":
            ref = O.render_camera(lambda oo, dd, off: O.active_outputs(sc, oo, dd), o, d)
        else:
            fs = models.frame_seed(seed, i)   # every render draws fresh masks: the i-th camera's stream
            ref = O.render_camera(lambda oo, dd, off: O.mcdropout_outputs(sc, oo, dd, K, fs, 0.2, ray_offset=off), o, d)
        gt = _gt(ref["rgb"], 100 + i)
        refs.append(_cpu_reference_metrics(ref, gt))
        eval_set.append((cam, gt))
    got = E.run_eval(ecfg, model, eval_set, experiment_name="exp", method_name=kind, checkpoint="ckpt")
    if kind == "mcdropout":
        assert model.config.mc_samples == K               # MCDropoutConfig.mc_samples overrides the model's (eval_uncertainty.py:1119)
    _check(got, refs)
    d = json.loads((tmp_path / "m.json").read_text())
    assert list(d) == ["experiment_name", "method_name", "checkpoint", "results"] and abs(d["results"]["psnr"] - got["psnr"]) < 1e-12


def test_run_eval_splat_on_hip_renders(dev, tmp_path):
    """active-splatfacto through the harness: RGBA ground truth is composited over the render's background
    (eval_uncertainty.py:321-322, 676-677) before the metrics."""
    from uncertainty_nerf_gs_amd import eval as E
    from uncertainty_nerf_gs_amd import metrics as M
    from oracle import splat_oracle as SO
    import test_gpu_splat as TS
    m, cam, g = TS._fixture_model(dev)
    gp = {k[3:]: g[k] for k in g.files if k.startswith("gp_")}
    fx, fy, cx, cy, Hs, Ws = g["intr"]
    ref = {k: torch.from_numpy(np.asarray(v)) for k, v in SO.active_splatfacto_outputs(
        gp, g["c2w"], fx, fy, cx, cy, int(Hs), int(Ws), np.array([0.1490, 0.1647, 0.2157], np.float32)).items() if not k.startswith("_")}
    gen = torch.Generator().manual_seed(5)
    rgba = torch.cat([_gt(ref["rgb"], 7), (torch.rand(int(Hs), int(Ws), 1, generator=gen) > 0.2).float()], dim=-1)
    gt = m.composite_gt(rgba, ref["background"])
    refm = _cpu_reference_metrics(ref, gt.cpu())
    ecfg = E.ActiveSplatfactoConfig(load_config=None, output_path=tmp_path / "s.json")
    got = E.run_eval(ecfg, m, [(cam, rgba)], method_name="active-splatfacto")
    _check(got, [refm], tol_ause=2e-3)
