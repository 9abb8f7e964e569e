"""tcnn-layout ("implementation=tcnn") host logic: level records, FullyFusedMLP parameter unpacking and the
state-dict names a reference checkpoint trained with tiny-cuda-nn carries.  [UPSTREAM-RECALL: tiny-cuda-nn is
neither under /root/reference nor installed; the layout follows SURVEY.md A.6 -- parity unpinned.]"""
import math

import pytest
import torch

from oracle import nerf_oracle as O
from uncertainty_nerf_gs_amd import fields as F
from uncertainty_nerf_gs_amd import models, ops, plugin


@pytest.mark.parametrize("L,base,max_res,log2T", [(16, 16, 2048, 19), (5, 16, 128, 17), (5, 16, 256, 17), (16, 16, 2048, 14)])
def test_level_records_product_equals_oracle_and_follow_tcnn_rules(L, base, max_res, log2T):
    growth = math.exp((math.log(max_res) - math.log(base)) / (L - 1))
    got = ops.tcnn_grid_levels(L, base, growth, log2T)
    assert got == O.tcnn_grid_levels(L, base, growth, log2T)
    off = 0
    for l, (scale, res, offset, size, dense) in enumerate(got):
        assert res == math.ceil(scale) + 1 and offset == off and size % 8 == 0 and size <= (1 << log2T)
        assert dense == int(res ** 3 <= size)
        if not dense:
            assert size == 1 << log2T          # hashed levels are power-of-two sized (mod == mask)
        off += size
    assert got[0][0] == base - 1.0 and abs(got[-1][0] - (max_res - 1)) < 1e-2   # scale = res_l - 1
    assert got[0][4] == 1 and got[-1][4] == 0                                   # coarse dense, fine hashed


def test_fully_fused_mlp_unpacking_round_trip():
    g = torch.Generator().manual_seed(0)
    for in_dim, width, layers, out_dim in ((32, 64, 2, 17), (63, 64, 3, 3), (10, 16, 2, 1)):
        pad = lambda n: -(-n // 16) * 16
        mats = [torch.randn(width, pad(in_dim), generator=g)] + [torch.randn(width, width, generator=g) for _ in range(layers - 2)] \
            + [torch.randn(pad(out_dim), width, generator=g)]
        params = torch.cat([m.reshape(-1) for m in mats])
        ws = F.unpack_tcnn_mlp(params, in_dim, width, layers, out_dim)
        ws_o = O.unpack_tcnn_mlp(params, in_dim, width, layers - 1, out_dim)
        assert [tuple(w.shape) for w in ws] == [(width, in_dim)] + [(width, width)] * (layers - 2) + [(out_dim, width)]
        for a, b in zip(ws, ws_o):
            assert torch.equal(a, b)
        assert torch.equal(ws[0], mats[0][:, :in_dim]) and torch.equal(ws[-1], mats[-1][:out_dim])
    with pytest.raises(ValueError):
        F.unpack_tcnn_mlp(torch.zeros(100), 32, 64, 2, 17)


def _tcnn_cfg(method):
    cfg = plugin.MODEL_CONFIGS[method]()
    cfg.implementation = "tcnn"
    cfg.log2_hashmap_size = 14
    cfg.proposal_net_args_list = [dict(a, log2_hashmap_size=12) for a in cfg.proposal_net_args_list]
    return cfg


def test_tcnn_checkpoint_key_names_and_sizes():
    """what `ns-train active-nerfacto` with tiny-cuda-nn installed writes (SURVEY.md 8b state-dict row)"""
    cfg = _tcnn_cfg("active-nerfacto")
    m = cfg._target(cfg, num_train_data=3)
    keys = set(m.state_dict())
    for k in ("field.mlp_base_grid.tcnn_encoding.params", "field.mlp_base.0.tcnn_encoding.params",
              "field.mlp_base_mlp.tcnn_encoding.params", "field.mlp_base.1.tcnn_encoding.params",
              "field.mlp_head.tcnn_encoding.params", "proposal_networks.0.mlp_base.encoder.tcnn_encoding.params",
              "proposal_networks.1.mlp_base.mlp.tcnn_encoding.params",
              "field.embedding_appearance.embedding.weight"):
        assert k in keys, k
    assert not any("hash_table" in k or ".layers." in k for k in keys)
    sd = m.state_dict()
    lv = ops.tcnn_grid_levels(16, 16, math.exp((math.log(2048) - math.log(16)) / 15), 14)
    assert sd["field.mlp_base_grid.tcnn_encoding.params"].numel() == 2 * (lv[-1][2] + lv[-1][3])
    assert sd["field.mlp_base_mlp.tcnn_encoding.params"].numel() == 64 * 32 + 32 * 64      # 17 outputs pad to 32
    assert sd["field.mlp_head.tcnn_encoding.params"].numel() == 64 * 64 + 64 * 64 + 16 * 64  # 63 inputs pad to 64
    assert sd["proposal_networks.0.mlp_base.mlp.tcnn_encoding.params"].numel() == 16 * 16 + 16 * 16
    # mc-dropout / laplace keep their torch create_mlp heads; only the grid is a tcnn module (mcdropout_fields.py:115-135)
    for method, grid_key, torch_key in (("nerfacto-mcdropout", "field.mlp_base_grid.tcnn_encoding.params", "field.mlp_base.3.weight"),
                                        ("nerfacto-laplace", "field.base_grid.tcnn_encoding.params", "field.mlp_rgb_ll.weight")):
        c = _tcnn_cfg(method)
        k2 = set(c._target(c, num_train_data=3).state_dict())
        assert grid_key in k2 and torch_key in k2


def test_tcnn_oracle_lookup_properties():
    """restatement sanity on the CPU: dense levels interpolate their own cells exactly, features are continuous
    across cell borders, and an all-ones table returns exactly 1 (weights sum to one)."""
    lv = O.tcnn_grid_levels(5, 16, 2.0, 12)
    rows = lv[-1][2] + lv[-1][3]
    g = torch.Generator().manual_seed(1)
    x = torch.rand(500, 3, generator=g)
    ones = O.tcnn_hash_encode(x, torch.ones(rows, 2), lv)
    assert torch.allclose(ones, torch.ones_like(ones), atol=2e-7)
    table = torch.randn(rows, 2, generator=g)
    eps = 1e-6
    xb = x.clone()
    xb[:, 0] = (torch.floor(x[:, 0] * 15 + 0.5) + 0.5) / 15      # a cell border of level 0 (scale 15, +0.5 shift)
    a = O.tcnn_hash_encode((xb - torch.tensor([eps, 0, 0])).clamp(0, 1), table, lv)[:, :2]
    b = O.tcnn_hash_encode((xb + torch.tensor([eps, 0, 0])).clamp(0, 1), table, lv)[:, :2]
    assert (a - b).abs().max() < 1e-3
    idx, w = O.tcnn_hash_indices(x, lv)
    assert idx.shape == (500, 5, 8) and w.shape == (500, 5, 3)
    for l, (_, res, off, size, dense) in enumerate(lv):
        assert (idx[:, l] >= off).all() and (idx[:, l] < off + size).all()


def test_half_fma_rounding_of_the_oracle_is_exact():
    """oracle._round_f16_of_sum = round-to-nearest-even-f16(p + c) for p a product of two halves: checked against exact
    rational arithmetic, including sums whose exact value needs more than 53 bits and constructed ties"""
    from fractions import Fraction
    import numpy as np
    from oracle.nerf_oracle import _round_f16_of_sum
    rng = np.random.default_rng(0)

    def halves(n, lo, hi):
        m, e, s = rng.integers(0, 2048, n), rng.integers(lo, hi, n), rng.choice([-1, 1], n)
        return (s * m * np.exp2(e.astype(np.float64) - 10)).astype(np.float16)

    n = 4000
    w, v, c = np.abs(halves(n, -24, 0)), halves(n, -24, 15), halves(n, -24, 15)
    got = _round_f16_of_sum(w.astype(np.float64) * v.astype(np.float64), c.astype(np.float64))
    for i in range(n):
        fr = Fraction(float(w[i])) * Fraction(float(v[i])) + Fraction(float(c[i]))
        if abs(fr) > 65000:
            continue
        h = np.float16(float(fr))
        best = min((x for x in (h, np.nextafter(h, np.float16(np.inf)), np.nextafter(h, np.float16(-np.inf))) if np.isfinite(x)),
                   key=lambda x: (abs(Fraction(float(x)) - fr), int(np.float16(x).view(np.uint16)) & 1))
        assert best == got[i], (w[i], v[i], c[i], best, got[i])
    # a product exactly on a half midpoint, nudged by an addend below the float64 resolution of the sum
    p = np.array([1.0 + 2.0 ** -11] * 2)
    assert list(_round_f16_of_sum(p, np.array([2.0 ** -24, -2.0 ** -24]))) == [np.float16(1.0 + 2.0 ** -10), np.float16(1.0)]


def test_tcnn_half_encode_is_the_fp32_encode_to_half_precision():
    import math
    import torch
    from oracle import nerf_oracle as O
    lv = O.tcnn_grid_levels(8, 16, math.exp((math.log(512) - math.log(16)) / 7), 12)
    g = torch.Generator().manual_seed(1)
    table = torch.rand(lv[-1][2] + lv[-1][3], 2, generator=g) * 2 - 1
    x = torch.rand(500, 3, generator=g)
    h, f = O.tcnn_hash_encode_half(x, table, lv), O.tcnn_hash_encode(x, table, lv)
    assert torch.equal(h, h.half().float()) and (h - f).abs().max() < 4e-3 and (h - f).abs().max() > 1e-5
