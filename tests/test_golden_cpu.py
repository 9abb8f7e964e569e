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
    cnt = keeps[:8].sum(axis=0).reshape(-1)
    hist = np.bincount(cnt, minlength=9) / N
    binom = np.array([comb(8, i) * 0.8 ** i * 0.2 ** (8 - i) for i in range(9)])
    assert np.abs(hist - binom).max() < tol
    # p = 0 keeps every unit; the signed-half test is the unsigned test on (half ^ 0x8000)
    assert O.mc_keep_mask(1234, 5, sidx[:1000], 1, 64, 0.0).all()
    s0 = O.mc_keep_mask(1234, 3, sidx[:100], 0, 64, p)
    assert not np.array_equal(s0, O.mc_keep_mask(1234, 3, sidx[:100], 1, 64, p))        # trunk and head streams differ
    assert not np.array_equal(s0, O.mc_keep_mask(1235, 3, sidx[:100], 0, 64, p))        # and so do seeds
