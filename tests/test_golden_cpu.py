"""Pin the oracle and the host-side mirrors against vectors produced by the reference's own
code (tests/golden/make_golden.py imported /root/reference to write them)."""
import json
import os
import warnings

import numpy as np
import pytest
import torch
from torch import nn

from conftest import GOLDEN, golden
from oracle import nerf_oracle as O


# ------------------------------------------------------------------ oracle vs reference
def test_oracle_get_weights_matches_reference_ComputeWeightsModule():
    g = golden("get_weights.npz")
    w = O.get_weights(torch.from_numpy(g["density"])[..., 0], torch.from_numpy(g["deltas"])[..., 0])
    assert np.array_equal(w.numpy(), g["weights"][..., 0])


@pytest.mark.parametrize("tag,out_dim,act", [("density", 1, "exp"), ("rgb", 3, "sigmoid")])
def test_oracle_sample_laplace_matches_reference(tag, out_dim, act):
    g = golden("sample_laplace.npz")
    mu_q, ggn, x = (torch.from_numpy(g[f"{tag}_{k}"]) for k in ("mu_q", "ggn", "x"))
    torch.manual_seed(int(g[f"{tag}_seed"]))
    noise = torch.randn(100, mu_q.numel())  # the draw the reference makes at laplace_field.py:545
    ws = O.laplace_weight_samples(mu_q, ggn, 1.0, 1e-9, noise)
    mean, var = O.sample_laplace(ws, act, x, out_dim)
    np.testing.assert_allclose(mean.numpy(), g[f"{tag}_mean"], rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(var.numpy(), g[f"{tag}_var"], rtol=1e-4, atol=2e-7)


@pytest.mark.parametrize("tag", ["plain", "alea"])
def test_oracle_ensemble_aggregate_matches_reference(tag):
    g = golden("ensemble.npz")
    members = []
    for i in range(5):
        members.append({k[len(f"{tag}_in{i}_"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(f"{tag}_in{i}_")})
    out = O.ensemble_aggregate(members)
    expect = {k[len(f"{tag}_out_"):]: g[k] for k in g.files if k.startswith(f"{tag}_out_")}
    assert set(out) == set(expect)
    for k, v in expect.items():
        assert np.array_equal(out[k].numpy(), v), k


def test_oracle_mc_aggregation_matches_reference():
    g = golden("mc_aggregate.npz")
    passes = [{k[len(f"in{i}_"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(f"in{i}_")} for i in range(8)]
    res = {}
    for key in passes[0]:
        el = torch.stack([p[key] for p in passes], dim=0)
        res[key] = el.mean(dim=0)
        if key in ("rgb", "depth", "expected_depth"):
            res[key + "_std"] = el.std(dim=0).mean(dim=-1)[..., None]
    expect = {k[4:]: g[k] for k in g.files if k.startswith("out_")}
    assert set(res) == set(expect)
    for k, v in expect.items():
        assert np.array_equal(res[k].numpy(), v), k


# ------------------------------------------------------- host mirrors vs reference
def test_create_mlp_topology_matches_reference():
    from uncertainty_nerf_gs_amd.utils import create_mlp
    spec = json.load(open(os.path.join(GOLDEN, "create_mlp.json")))
    names = {"ReLU": nn.ReLU, "Sigmoid": nn.Sigmoid, None: None}
    for case, d in spec.items():
        kw = dict(d["kwargs"])
        kw["activation"] = names[kw["activation"]]
        kw["out_activation"] = names[kw["out_activation"]]
        if kw.get("skip_connections") is not None:
            kw["skip_connections"] = tuple(kw["skip_connections"])
        m = create_mlp(**kw)
        got = []
        for layer in m:
            e = {"type": type(layer).__name__}
            if isinstance(layer, nn.Linear):
                e.update(in_features=layer.in_features, out_features=layer.out_features)
            if isinstance(layer, nn.Dropout):
                e.update(p=layer.p)
            got.append(e)
        assert got == d["modules"], case


def test_create_mlp_state_dict_names():
    from uncertainty_nerf_gs_amd.utils import create_mlp
    trunk = create_mlp(32, 2, 64, 16, activation=nn.ReLU, dropout_layers=[-1], dropout_rate=0.2)
    head = create_mlp(63, 3, 64, 3, activation=nn.ReLU, out_activation=nn.Sigmoid, dropout_layers=[-1], dropout_rate=0.2)
    assert sorted(trunk.state_dict()) == ["0.bias", "0.weight", "3.bias", "3.weight"]
    assert sorted(head.state_dict()) == ["0.bias", "0.weight", "2.bias", "2.weight", "5.bias", "5.weight"]


@pytest.mark.parametrize("tag", ["good", "bad"])
@pytest.mark.parametrize("et", ["rmse", "mae", "mse"])
def test_ause_matches_reference(tag, et):
    from uncertainty_nerf_gs_amd.metrics import ause
    g = golden("metrics.npz")
    ratio, e, ev, a = ause(torch.from_numpy(g[f"unc_{tag}"]), torch.from_numpy(g["err"]), et)
    assert abs(a - float(g[f"ause_{tag}_{et}"])) < 2e-6
    np.testing.assert_allclose(e, g[f"ause_{tag}_{et}_curve"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(ev, g[f"ause_{tag}_{et}_curve_by_var"], rtol=0, atol=2e-6)


def test_auce_matches_reference():
    from uncertainty_nerf_gs_amd.metrics import auce
    g = golden("metrics.npz")
    d = auce(g["auce_mean"], g["auce_sigma"], g["auce_target"])
    for k, v in d.items():
        np.testing.assert_allclose(np.asarray(v), g["auce_" + k], rtol=1e-12, atol=1e-12, err_msg=k)


def test_nll_matches_torch_normal():
    from uncertainty_nerf_gs_amd.metrics import negative_gaussian_loglikelihood
    g = torch.Generator().manual_seed(0)
    p, t, s = torch.rand(50, 3, generator=g), torch.rand(50, 3, generator=g), torch.rand(50, 1, generator=g) * 0.2
    ref = -torch.distributions.Normal(p, torch.clamp_min(s, 3e-2)).log_prob(t)
    torch.testing.assert_close(negative_gaussian_loglikelihood(p, t, s, 3e-2), ref, rtol=1e-5, atol=1e-6)


def test_auce_torch_equals_the_reference_pinned_loop():
    """the one-sort device AUCE against metrics.auce (itself pinned to the reference by the golden vectors)"""
    import torch
    from uncertainty_nerf_gs_amd import metrics as M
    g = torch.Generator().manual_seed(17)
    for n, scale in ((5000, 1.0), (777, 0.2), (4096, 3.0)):
        mean = torch.rand(n, 3, generator=g)
        sigma = 0.02 + scale * 0.1 * torch.rand(n, 3, generator=g)
        target = mean + sigma * torch.randn(n, 3, generator=g) * 1.3
        sigma[:5] = 0.0                      # degenerate intervals: covered only when target == mean
        target[:2] = mean[:2]
        ref = M.auce(mean.numpy(), sigma.numpy(), target.numpy())
        got = M.auce_torch(mean, sigma, target)
        assert set(ref) == set(got)
        np.testing.assert_array_equal(got["coverage_values"], ref["coverage_values"])
        np.testing.assert_allclose(got["avg_length_values"], ref["avg_length_values"], rtol=1e-6)
        for k in ("auc_abs_error_values", "auc_neg_error_values"):
            assert abs(got[k] - ref[k]) < 1e-12
        assert abs(got["auc_length_values"] - ref["auc_length_values"]) < 1e-6 * ref["auc_length_values"]


def test_dropout_mask_generator_statistics():
    """The MC-dropout masks (oracle.mc_keep_mask = twin of the kernel's unerf_mask_word0 / unerf_mask_step): keep
    rate 1 - p in both 16-bit halves at every pass, no correlation between ANY two of K = 10 passes (same unit, and
    low half against high half of the stepped word), between the halves of a word or between neighbouring words,
    and the number of keeps of one unit over 8 passes is Binomial(8, 0.8) -- on 4 M units per pass, tolerances at
    ~5 sigma of the sampling noise."""
    from math import comb
    from oracle import nerf_oracle as O
    n, K, p = 62500, 10, 0.2
    sidx = np.arange(n, dtype=np.int64) * 7 + 11
    keeps = np.stack([O.mc_keep_mask(1234, k, sidx, 0, 64, p) for k in range(K)])      # [K, n, 64]
    N = n * 64
    tol = 5.0 / np.sqrt(N)
    assert np.abs(keeps.reshape(K, -1).mean(axis=1) - (1 - p)).max() < tol * 0.5        # std of a rate = 0.4 / sqrt(N)
    assert abs(keeps[:, :, 0::2].mean() - 0.8) < tol and abs(keeps[:, :, 1::2].mean() - 0.8) < tol

    def corr(a, b):
        a = a.reshape(-1).astype(np.float32)
        b = b.reshape(-1).astype(np.float32)
        a -= a.mean()
        b -= b.mean()
        return abs(float(np.dot(a, b)) / float(np.sqrt(np.dot(a, a) * np.dot(b, b))))
    for k in range(K):
        for m in range(k + 1, K):
            assert corr(keeps[k], keeps[m]) < tol, (k, m)                               # same unit, any two passes
            assert corr(keeps[k][:, 0::2], keeps[m][:, 1::2]) < tol * 1.5, (k, m)       # low half -> high half of a later word
            assert corr(keeps[k][:, 1::2], keeps[m][:, 0::2]) < tol * 1.5, (k, m)
    assert corr(keeps[:, :, 0::2], keeps[:, :, 1::2]) < tol                             # the two halves of a word
    assert corr(keeps[:, :, :-2], keeps[:, :, 2:]) < tol                                # neighbouring words
    # every pair of the 64 units of a sample (pass 0: the words are multilinear hashes of ONE base hash per lane half)
    k0 = keeps[0].astype(np.float64)
    k0 -= k0.mean(axis=0)
    c = (k0.T @ k0) / n / np.outer(k0.std(axis=0), k0.std(axis=0))
    np.fill_diagonal(c, 0.0)
    assert np.abs(c).max() < 5.0 / np.sqrt(n)
    assert np.abs(keeps.mean(axis=(0, 1)) - 0.8).max() < 5.0 * 0.4 / np.sqrt(n * K)        # every unit's own keep rate
    cnt = keeps[:8].sum(axis=0).reshape(-1)
    hist = np.bincount(cnt, minlength=9) / N
    binom = np.array([comb(8, i) * 0.8 ** i * 0.2 ** (8 - i) for i in range(9)])
    assert np.abs(hist - binom).max() < tol
    # p = 0 keeps every unit; the signed-half test is the unsigned test on (half ^ 0x8000)
    assert O.mc_keep_mask(1234, 5, sidx[:1000], 1, 64, 0.0).all()
    s0 = O.mc_keep_mask(1234, 3, sidx[:100], 0, 64, p)
    assert not np.array_equal(s0, O.mc_keep_mask(1234, 3, sidx[:100], 1, 64, p))        # trunk and head streams differ
    assert not np.array_equal(s0, O.mc_keep_mask(1235, 3, sidx[:100], 0, 64, p))        # and so do seeds


@pytest.mark.parametrize("p", [0.5, 0.25, 0.1, 1.0 / 3.0, 0.75, 0.5 + 2.0 ** -9, 0.125, 2.0 ** -6])
def test_dropout_mask_generator_at_other_rates_and_its_joint_structure(p):
    """the statistics above are at p = 0.2 only, and a unit's K keep bits are a function of 16 bits of state (the
    stepped halves are a 16-bit LCG).  Here, for thresholds at, next to and away from powers of two, on 2.56 M units x 8
    passes: the keep rate of every pass, no PAIRWISE dependence between passes (what the mean and the variance over the K
    passes -- the only statistics the renderer forms, mcdropout_models.py:116-126 -- are functions of) beyond sampling
    noise, the variance of a unit's keep count over 8 passes within 0.5 % of Binomial(8, 1 - p)'s; and the third-order
    structure a 16-bit state must have is BOUNDED, not absent: worst triple moment of centred keeps <= 1.2 % of
    (q (1 - q))^1.5 (measured 0.2 - 0.8 %), total-variation distance of the keep-count histogram from the binomial
    <= 0.8 % (measured 0.04 - 0.5 %).  DESIGN.md 4.2 records it."""
    from math import comb
    from oracle import nerf_oracle as O
    n, K = 40000, 8
    sidx = np.arange(n, dtype=np.int64) * 7 + 11
    keeps = np.stack([O.mc_keep_mask(1234, k, sidx, 0, 64, p) for k in range(K)]).astype(np.float64)
    N = n * 64
    q = round((1.0 - p) * 65536.0) / 65536.0
    assert np.abs(keeps.reshape(K, -1).mean(axis=1) - q).max() < 5.0 * np.sqrt(q * (1 - q) / N)
    c = keeps - q
    pair = max(abs((c[a] * c[b]).mean()) for a in range(K) for b in range(a + 1, K)) / (q * (1 - q))
    assert pair < 6.0 / np.sqrt(N), pair
    triple = max(abs((c[a] * c[b] * c[d]).mean()) for a in range(K) for b in range(a + 1, K) for d in range(b + 1, K))
    assert triple / (q * (1 - q)) ** 1.5 < 1.2e-2
    cnt = keeps.sum(axis=0).reshape(-1).astype(int)
    hist = np.bincount(cnt, minlength=K + 1) / N
    binom = np.array([comb(K, i) * q ** i * (1 - q) ** (K - i) for i in range(K + 1)])
    assert 0.5 * np.abs(hist - binom).sum() < 8e-3
    assert abs(cnt.var() / (K * q * (1 - q)) - 1.0) < 5e-3


# ---- [REF] model glue run with a fake `self` (tests/golden/make_golden.py: golden_splat_get_outputs / golden_nerf_model_glue)


def test_depth_draw_generator_statistics():
    """The Laplace depth draws' built-in generator (one xorshift32 stream per (ray, sample), 16-bit uniforms, Box-Muller
    with both outputs; twin of unerf_depth_stream_seed / unerf_xorshift32 / unerf_normal_pair_from_state): standard-normal
    moments and no linear or quadratic dependence between neighbouring samples of a ray, successive draws, the cos / sin
    outputs of a pair, or different seeds -- over 4.8 M normals (100 draws of 48 x 1000 samples)."""
    sidx = np.arange(48 * 1000)
    Z = np.stack([O.normal_noise(7, d, sidx) for d in range(100)])          # [100 draws, N samples]
    assert np.isfinite(Z).all() and np.abs(Z).max() < 4.86                    # sqrt(34 ln 2): the 16-bit radius bound
    assert abs(Z.mean()) < 2e-3 and abs(Z.std() - 1) < 2e-3
    assert abs(((Z - Z.mean()) ** 4).mean() / Z.var() ** 2 - 3.0) < 2e-2       # kurtosis
    assert abs((Z ** 3).mean()) < 1e-2                                         # skewness
    c = lambda a, b: abs(np.corrcoef(a.ravel(), b.ravel())[0, 1])
    assert c(Z[:, :-1], Z[:, 1:]) < 2e-3                                       # neighbouring samples, same draw
    assert c(Z[:-1], Z[1:]) < 2e-3 and c(Z[:-2], Z[2:]) < 2e-3                 # successive draws (cos / sin of a pair; next pair)
    assert c(Z[:-1] ** 2, Z[1:] ** 2) < 2e-3 and c(Z[:, :-1] ** 2, Z[:, 1:] ** 2) < 2e-3
    assert abs(Z.mean(0).std() - 0.1) < 2e-3                                   # per-sample means over the 100 draws
    Z2 = np.stack([O.normal_noise(8, d, sidx) for d in range(4)])
    assert c(Z[:4], Z2) < 5e-3                                                 # another seed: another stream
    # the stream is keyed by the GLOBAL sample index: a launch group starting at ray offset r sees the same draws
    assert np.array_equal(O.normal_noise(7, 5, sidx[4800:9600]), Z[5, 4800:9600])


@pytest.mark.parametrize("tag", ["default", "white", "sh0", "early", "aa"])
def test_oracle_splat_outputs_match_the_references_four_pass_get_outputs(tag):
    """ActiveSplatfactoModel.get_outputs itself (activesplatfacto_model.py:142-367) ran on these splats with gsplat's
    three entry points bound to the oracle's restatements: the oracle's one-sort / 5-channel restructuring, the
    background handling, sh_degree 0, the SH-degree schedule and the antialiased opacity must reproduce its dict."""
    from oracle import splat_oracle as SO
    g = golden("splat_get_outputs.npz")
    gp = {k[3:]: g[k] for k in g.files if k.startswith("gp_")}
    fx, fy, cx, cy, H, W = g["intr"]
    sh_degree, step, aa = (int(v) for v in g[f"{tag}_cfg"])
    n = min(step // 1000, sh_degree) if sh_degree > 0 else 0
    out = SO.active_splatfacto_outputs(gp, g["c2w"], fx, fy, cx, cy, int(H), int(W), g[f"{tag}_out_background"],
                                       beta_min=0.01, sh_degree=n, rasterize_mode="antialiased" if aa else "classic",
                                       config_sh_degree=sh_degree)
    keys = [k[len(tag) + 5:] for k in g.files if k.startswith(f"{tag}_out_")]
    assert set(keys) == {"rgb", "depth", "accumulation", "background", "uncertainty", "rgb_var", "rgb_std", "depth_var", "depth_std"}
    assert set(keys) == {k for k in out if not k.startswith("_")}
    for k in keys:
        ref = g[f"{tag}_out_{k}"]
        # same primitives, same order of operations: the 5-channel pass blends each channel exactly like the
        # reference's separate 3-channel passes, so the images agree to the last bit
        np.testing.assert_array_equal(np.asarray(out[k], dtype=np.float32).reshape(ref.shape), ref, err_msg=f"{tag}:{k}")
    assert float(g[f"{tag}_out_accumulation"].max()) > 0.5 and float(g[f"{tag}_out_accumulation"].min()) < 0.5


@pytest.mark.parametrize("tag,det", [("lap", False), ("lapdet", True)])
def test_oracle_laplace_compose_matches_the_references_get_outputs_unc(tag, det):
    """NerfactoLaplaceModel.get_outputs_unc (laplace_model.py:456-556) ran on these field outputs (renderers = the
    oracle's, get_weights = the reference's own ComputeWeightsModule, the 100 density draws = torch's Normal with a
    recorded seed): the oracle's laplace_compose must reproduce every key."""
    from oracle import nerf_oracle as O
    g = golden("nerf_model_glue.npz")
    t = lambda k: torch.from_numpy(g[k])
    out = O.laplace_compose(t("eb"), t("density")[..., 0], t("lap_density_var")[..., 0], t("rgb"), t("lap_rgb_var")[..., 0],
                            t(f"{tag}_noise"), use_deterministic_density=det)
    for i, (w, eb) in enumerate(((t("w0"), t("eb0")), (t("w1"), t("eb1")))):
        out[f"prop_depth_{i}"] = O.render_depth_median(w[..., 0], (eb[..., :-1] + eb[..., 1:]) / 2)
    keys = [k[len(tag) + 5:] for k in g.files if k.startswith(f"{tag}_out_")]
    assert set(keys) == set(out), set(keys) ^ set(out)
    for k in keys:
        torch.testing.assert_close(out[k].reshape(g[f"{tag}_out_{k}"].shape), t(f"{tag}_out_{k}"), rtol=2e-6, atol=1e-7, msg=f"{tag}:{k}")


def test_oracle_active_compose_matches_the_references_get_outputs():
    """ActiveNerfactoModel.get_outputs (activenerfacto_model.py:83-152) with a fake self: key set, rgb_var = sum w^2 beta,
    depth_var around the median depth + 1e-5, the raw density output, the prop depths."""
    from oracle import nerf_oracle as O
    g = golden("nerf_model_glue.npz")
    t = lambda k: torch.from_numpy(g[k])
    out = O.active_compose(t("eb"), t("density")[..., 0], t("rgb"), t("act_beta")[..., 0])
    for i, (w, eb) in enumerate(((t("w0"), t("eb0")), (t("w1"), t("eb1")))):
        out[f"prop_depth_{i}"] = O.render_depth_median(w[..., 0], (eb[..., :-1] + eb[..., 1:]) / 2)
    keys = [k[len("act_out_"):] for k in g.files if k.startswith("act_out_")]
    assert set(keys) == set(out), set(keys) ^ set(out)
    for k in keys:
        torch.testing.assert_close(out[k].reshape(g[f"act_out_{k}"].shape), t(f"act_out_{k}"), rtol=2e-6, atol=1e-7, msg=k)


# ---- [REF] Field methods run with a fake `self` (tests/golden/make_golden.py: golden_field_glue) -----------------------

def _field_from_fixture(g, prefix, kind):
    from oracle import nerf_oracle as O
    t = lambda k: torch.from_numpy(g[prefix + k])
    from uncertainty_nerf_gs_amd import synthetic
    table = t("table")
    log2T = int(np.log2(table.shape[0] // 16))
    grid = O.GridMLP(table=table, scalings=t("scalings"), log2_T=log2T, weights=[t("w0"), t("w1")], biases=[t("b0"), t("b1")])
    kw = {}
    if kind == "laplace":
        kw = dict(hidden_w=t("w1"), hidden_b=t("b1"), density_w=t("density_w"), density_b=t("density_b"))
    if prefix == "mc_":
        return O.FieldParams(grid=grid, head_w=[], head_b=[], appearance=torch.zeros(32), average_init_density=0.01)
    return O.FieldParams(grid=grid, head_w=[t(f"head_w{i}") for i in range(3)], head_b=[t(f"head_b{i}") for i in range(3)],
                         appearance=t("appearance"), **kw)


def test_oracle_laplace_field_matches_the_references_forward_unc():
    """NerfactoLaplaceField.forward_unc -> get_density -> sample_laplace, get_outputs -> sample_laplace
    (laplace_field.py:279-568) ran on a seeded field: bare-Linear base_mlp, mu_d NOT selector-masked, relu + channel
    mean of the colour variance, and (use_deterministic_density) the masked plain density with a still-sampled colour."""
    g = golden("field_glue.npz")
    fp = _field_from_fixture(g, "lap_", "laplace")
    t = lambda k: torch.from_numpy(g[k])
    o, d, eb = t("o"), t("d"), t("eb")
    mu_q_d = torch.cat([fp.density_w.reshape(-1), fp.density_b.reshape(-1)])
    mu_q_r = torch.cat([fp.head_w[2].reshape(-1), fp.head_b[2].reshape(-1)])
    ws_d = O.laplace_weight_samples(mu_q_d, t("lap_ggn_density"), 1.0, 1e-9, t("lapf_noise_density"))
    ws_r = O.laplace_weight_samples(mu_q_r, t("lap_ggn_rgb"), 1.0, 1e-9, t("lapf_noise_rgb"))
    mu_d, var_d, mu_rgb, var_rgb = O.laplace_field(o, d, eb, fp, ws_d, ws_r)
    torch.testing.assert_close(mu_d, t("lapf_density")[..., 0], rtol=2e-6, atol=1e-7)
    torch.testing.assert_close(var_d, t("lapf_density_var")[..., 0], rtol=1e-4, atol=1e-6 * float(t("lapf_density")[..., 0].max()) ** 2)
    torch.testing.assert_close(mu_rgb, t("lapf_rgb"), rtol=0, atol=2e-6)
    torch.testing.assert_close(var_rgb, t("lapf_rgb_var")[..., 0], rtol=0, atol=2e-7)
    # use_deterministic_density: density = plain masked head (no draw), colour head sampled with the first draw of the seed
    ws_r2 = O.laplace_weight_samples(mu_q_r, t("lap_ggn_rgb"), 1.0, 1e-9, t("lapf_det_noise_rgb"))
    dens_det, _ = O.laplace_field_deterministic(o, d, eb, fp)
    _, _, mu_rgb2, var_rgb2 = O.laplace_field(o, d, eb, fp, ws_d, ws_r2)
    torch.testing.assert_close(dens_det, t("lapf_det_density")[..., 0], rtol=2e-6, atol=1e-7)
    torch.testing.assert_close(mu_rgb2, t("lapf_det_rgb"), rtol=0, atol=2e-6)
    torch.testing.assert_close(var_rgb2, t("lapf_det_rgb_var")[..., 0], rtol=0, atol=2e-7)
    assert "lapf_det_density_var" not in g.files                       # density_var is None on that path (:503)
    # the is_inference=False forward (what compute_hessian_naive differentiates)
    dens_p, rgb_p = O.laplace_field_deterministic(o, d, eb, fp)
    torch.testing.assert_close(dens_p, t("lapf_plain_density")[..., 0], rtol=2e-6, atol=1e-7)
    torch.testing.assert_close(rgb_p, t("lapf_plain_rgb"), rtol=0, atol=2e-6)
    # some samples lie outside the unit box after contraction-normalisation: the selector is exercised
    assert (t("lapf_plain_density") == 0).any() and (t("lapf_density") > 0).all()


def test_oracle_active_and_mcdropout_fields_match_the_references_get_density():
    g = golden("field_glue.npz")
    t = lambda k: torch.from_numpy(g[k])
    o, d, eb = t("o"), t("d"), t("eb")
    fpa = _field_from_fixture(g, "act_", "active")
    dens, rgb, beta = O.active_field(o, d, eb, fpa)                          # activenerfacto_field.py:162-215
    torch.testing.assert_close(dens, t("actf_density")[..., 0], rtol=2e-6, atol=1e-7)
    torch.testing.assert_close(beta, t("actf_rgb_var")[..., 0], rtol=2e-6, atol=1e-7)   # stored under "rgb_var" (:209)
    torch.testing.assert_close(rgb, t("actf_rgb"), rtol=0, atol=2e-6)
    fpm = _field_from_fixture(g, "mc_", "mcdropout")
    fpm.head_w = [torch.zeros(64, 63), torch.zeros(64, 64), torch.zeros(3, 64)]
    fpm.head_b = [torch.zeros(64), torch.zeros(64), torch.zeros(3)]
    dens_m, _ = O.mcdropout_field(o, d, eb, fpm, None, None, 0.2)           # mcdropout_fields.py:146-174, dropout off
    torch.testing.assert_close(dens_m, t("mcf_density")[..., 0], rtol=2e-6, atol=1e-9)
