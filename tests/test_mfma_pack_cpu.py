"""Host logic: the MFMA operand packer (ops.pack_field_mfma) arranges the MLP weights so that
field_kernel_mfma's dataflow computes the same network.  The kernel's dataflow is emulated here in
numpy with the documented v_mfma_f32_32x32x2_f32 lane maps (cdna_hip_programming.md section 3):
A[i=l&31][k=l>>5], B[k=l>>5][j=l&31], D row = (r&3)+8(r>>2)+4(l>>5), col = l&31."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from uncertainty_nerf_gs_amd import ops

LANE = np.arange(64)
I_, H_ = LANE & 31, LANE >> 5


def unit(r, h):
    return (r & 3) + 8 * (r >> 2) + 4 * h


def mfma(a_frag, b_vals, acc):
    """acc: [16, 64] accumulator registers per lane."""
    A = np.zeros((32, 2))
    B = np.zeros((2, 32))
    A[I_, H_] = a_frag
    B[H_, I_] = b_vals
    D = A @ B
    out = acc.copy()
    for r in range(16):
        out[r] += D[unit(r, H_), I_]
    return out


def emulate_tile(blob, feats, sh, mode_active=True):
    """feats [32 samples, 32], sh [32 samples, 16] -> (trunk_out [32,32 units], rgb_pre [32,3])"""
    blob = blob.numpy().astype(np.float64)
    fr = blob[:ops.MFMA_BIAS_OFF].reshape(ops.MFMA_FRAGS, 64)
    bias = blob[ops.MFMA_BIAS_OFF:ops.MFMA_H2_OFF].reshape(7, 2, 16)
    h2 = blob[ops.MFMA_H2_OFF:ops.MFMA_H2_OFF + 192].reshape(2, 2, 3, 16)
    hb2 = blob[ops.MFMA_H2_OFF + 192:ops.MFMA_H2_OFF + 195]
    j = I_
    # lane (j,h) holds feat[s] = feature 16h+s of sample j  (levels 8h..8h+7)
    feat = np.stack([feats[j, 16 * H_ + s] for s in range(16)])  # [16, 64]
    binit = lambda k: np.stack([bias[k, H_, r] for r in range(16)])
    acc = [binit(0), binit(1)]
    for blk in range(2):
        for s in range(16):
            acc[blk] = mfma(fr[blk * 16 + s], feat[s], acc[blk])
    acc = [np.maximum(a, 0) for a in acc]
    t = binit(2)
    for bi in range(2):
        for r in range(16):
            t = mfma(fr[32 + bi * 16 + r], acc[bi][r], t)
    shv = np.stack([sh[j, 8 * H_ + k] for k in range(8)])
    c = [binit(3), binit(4)]
    for blk in range(2):
        for s in range(8):
            c[blk] = mfma(fr[64 + blk * 16 + s], t[s], c[blk])
        for s in range(8, 16):
            c[blk] = mfma(fr[64 + blk * 16 + s], shv[s - 8], c[blk])
    c = [np.maximum(a, 0) for a in c]
    d = [binit(5), binit(6)]
    for blk in range(2):
        for bi in range(2):
            for r in range(16):
                d[blk] = mfma(fr[96 + blk * 32 + bi * 16 + r], c[bi][r], d[blk])
    d = [np.maximum(a, 0) for a in d]
    rgb = np.zeros((3, 64))
    for cc in range(3):
        for blk in range(2):
            for r in range(16):
                rgb[cc] += d[blk][r] * h2[blk, H_, cc, r]
    rgb = rgb + rgb[:, LANE ^ 32] + hb2[:, None]   # cross-half add (shfl_xor 32)
    trunk = np.zeros((32, 32))
    for r in range(16):
        trunk[j, unit(r, H_)] = t[r]
    return trunk, rgb[:, :32].T


def test_packed_operands_reproduce_the_mlp():
    g = torch.Generator().manual_seed(0)
    rnd = lambda *s: torch.randn(*s, generator=g) * 0.3
    w0, b0, w1, b1 = rnd(64, 32), rnd(64), rnd(17, 64), rnd(17)
    h0, hb0, h1, hb1, h2, hb2 = rnd(64, 31), rnd(64), rnd(64, 64), rnd(64), rnd(3, 64), rnd(3)
    blob = ops.pack_field_mfma(w0, b0, w1, b1, h0, hb0, h1, hb1, h2, hb2)
    assert blob.numel() == ops.MFMA_BLOB_FLOATS
    feats, sh = rnd(32, 32), rnd(32, 16)
    trunk, rgb = emulate_tile(blob, feats.numpy().astype(np.float64), sh.numpy().astype(np.float64))
    hid = F.relu(F.linear(feats.double(), w0.double(), b0.double()))
    t_ref = F.linear(hid, w1.double(), b1.double())
    x = torch.cat([sh.double(), t_ref[:, 1:16]], dim=-1)
    x = F.relu(F.linear(x, h0.double(), hb0.double()))
    x = F.relu(F.linear(x, h1.double(), hb1.double()))
    rgb_ref = F.linear(x, h2.double(), hb2.double())
    np.testing.assert_allclose(trunk[:, :17], t_ref.numpy(), rtol=1e-6, atol=1e-6)
    assert np.all(trunk[:, 17:] == 0)
    np.testing.assert_allclose(rgb, rgb_ref.numpy(), rtol=1e-6, atol=1e-6)


def test_packed_operands_16_wide_trunk():
    g = torch.Generator().manual_seed(1)
    rnd = lambda *s: torch.randn(*s, generator=g) * 0.3
    blob = ops.pack_field_mfma(rnd(64, 32), rnd(64), rnd(16, 64), rnd(16), rnd(64, 31), rnd(64), rnd(64, 64), rnd(64),
                               rnd(3, 64), rnd(3))
    fr = blob[:ops.MFMA_BIAS_OFF].view(ops.MFMA_FRAGS, 64)
    # rows 16..31 of the padded trunk-out block must be zero on every fragment
    assert torch.all(fr[32:64][:, (torch.arange(64) & 31) >= 16] == 0)


def emulate_laplace_tile(blob, lap, feats, sh, n_lap):
    """Laplace dataflow of field_kernel_mfma_laplace: bare Linear base, geo = mlp_hidden, sampled heads."""
    blob = blob.numpy().astype(np.float64)
    lap = lap.numpy().astype(np.float64)
    fr = blob[:ops.MFMA_BIAS_OFF].reshape(ops.MFMA_FRAGS, 64)
    bias = blob[ops.MFMA_BIAS_OFF:ops.MFMA_H2_OFF].reshape(7, 2, 16)
    lfr = lap[:ops.LAP_BIAS_OFF].reshape(4, ops.LAP_BLOCKS, 2, 16, 64)
    lbias = lap[ops.LAP_BIAS_OFF:].reshape(4, ops.LAP_BLOCKS, 2, 16)
    j = I_
    feat = np.stack([feats[j, 16 * H_ + s] for s in range(16)])
    binit = lambda k: np.stack([bias[k, H_, r] for r in range(16)])
    hb = [binit(0), binit(1)]
    for blk in range(2):
        for s in range(16):
            hb[blk] = mfma(fr[blk * 16 + s], feat[s], hb[blk])          # NO ReLU (utils.py:22-23 quirk)
    t = binit(2)
    for bi in range(2):
        for r in range(16):
            t = mfma(fr[32 + bi * 16 + r], hb[bi][r], t)               # geo = mlp_hidden(hb)

    def head(q, src, act):
        s1 = np.zeros(64)
        s2 = np.zeros(64)
        for b in range(ops.LAP_BLOCKS):
            acc = np.stack([lbias[q, b, H_, r] for r in range(16)])
            for bi in range(2):
                for r in range(16):
                    acc = mfma(lfr[q, b, bi, r], src[bi][r], acc)
            p = act(acc)
            s1 += p.sum(0)
            s2 += (p * p).sum(0)
        s1 = s1 + s1[LANE ^ 32]
        s2 = s2 + s2[LANE ^ 32]
        mu, mu2 = s1 / n_lap, s2 / n_lap
        return mu[:32], (mu2 - mu * mu)[:32]

    with np.errstate(over="ignore"):
        mu_d, var_d = head(0, hb, np.exp)
    shv = np.stack([sh[j, 8 * H_ + k] for k in range(8)])
    c = [binit(3), binit(4)]
    for blk in range(2):
        for s in range(8):
            c[blk] = mfma(fr[64 + blk * 16 + s], t[s], c[blk])
        for s in range(8, 16):
            c[blk] = mfma(fr[64 + blk * 16 + s], shv[s - 8], c[blk])
    c = [np.maximum(a, 0) for a in c]
    d = [binit(5), binit(6)]
    for blk in range(2):
        for bi in range(2):
            for r in range(16):
                d[blk] = mfma(fr[96 + blk * 32 + bi * 16 + r], c[bi][r], d[blk])
    d = [np.maximum(a, 0) for a in d]
    sig = lambda x: 1 / (1 + np.exp(-np.maximum(x, -700)))
    mus, vars_ = zip(*[head(1 + ch, d, sig) for ch in range(3)])
    return mu_d, var_d, np.stack(mus, -1), np.stack(vars_, -1)


def test_laplace_packed_heads_reproduce_sample_laplace():
    g = torch.Generator().manual_seed(2)
    rnd = lambda *s: torch.randn(*s, generator=g) * 0.3
    w0, b0, wh, bh = rnd(64, 32), rnd(64), rnd(15, 64), rnd(15)
    h0, hb0, h1, hb1, h2, hb2 = rnd(64, 31), rnd(64), rnd(64, 64), rnd(64), rnd(3, 64), rnd(3)
    n = 100
    ws_d, ws_r = rnd(n, 65), rnd(n, 195)
    blob = ops.pack_field_mfma(w0, b0, wh, bh, h0, hb0, h1, hb1, h2, hb2, geo_first_unit=0)
    lap = ops.pack_laplace_heads(ws_d, ws_r)
    assert lap.numel() == ops.LAP_BLOB_FLOATS
    feats, sh = rnd(32, 32), rnd(32, 16)
    mu_d, var_d, mu_c, var_c = emulate_laplace_tile(blob, lap, feats.numpy().astype(np.float64),
                                                    sh.numpy().astype(np.float64), n)
    hb = F.linear(feats.double(), w0.double(), b0.double())
    geo = F.linear(hb, wh.double(), bh.double())
    pd = torch.exp(F.linear(hb, ws_d[:, :64].double(), ws_d[:, 64].double()))          # [32, n]
    x = torch.cat([sh.double(), geo], dim=-1)
    x = F.relu(F.linear(x, h0.double(), hb0.double()))
    x = F.relu(F.linear(x, h1.double(), hb1.double()))
    np.testing.assert_allclose(mu_d, pd.mean(1).numpy(), rtol=1e-9)
    np.testing.assert_allclose(var_d, ((pd ** 2).mean(1) - pd.mean(1) ** 2).numpy(), rtol=1e-6, atol=1e-12)
    for ch in range(3):
        pc = torch.sigmoid(F.linear(x, ws_r[:, ch * 64:(ch + 1) * 64].double(), ws_r[:, 192 + ch].double()))
        np.testing.assert_allclose(mu_c[:, ch], pc.mean(1).numpy(), rtol=1e-9)
        np.testing.assert_allclose(var_c[:, ch], ((pc ** 2).mean(1) - pc.mean(1) ** 2).numpy(), rtol=1e-6, atol=1e-12)


# ---- split-f16 slabs (ops.pack_field_mfma16 / pack_laplace_heads16, field_kernel_mfma16*) --------------------
# v_mfma_f32_32x32x16_f16 lane maps: A[row l&31][k = 8(l>>5) + e], B[k = 8(l>>5) + e][col l&31], e = 0..7;
# D as for the other 32x32 forms.  The emulation recombines hi + lo in float64, i.e. it checks the ORDER the
# operands are packed in and that hi/lo carry the weights to ~2^-22; the three-product arithmetic itself is
# checked on the GPU against the exact kernels.

def _slabs16(blob, n_slabs):
    raw = blob[:n_slabs * ops.MF16_SLAB_FLOATS].contiguous().view(torch.int16).view(torch.float16)
    fr = raw.view(n_slabs, 2, 64, 8).to(torch.float64).numpy()
    return fr[:, 0] + fr[:, 1]                      # [slab][lane][8]: hi + lo


def mfma16(a_slab, b_vals, acc):
    """a_slab [64 lanes, 8], b_vals [64 lanes, 8] (this lane's 8 k-values of its column) -> acc [16, 64]"""
    A = np.zeros((32, 16))
    B = np.zeros((16, 32))
    for e in range(8):
        A[I_, 8 * H_ + e] = a_slab[:, e]
        B[8 * H_ + e, I_] = b_vals[:, e]
    D = A @ B
    out = acc.copy()
    for r in range(16):
        out[r] += D[unit(r, H_), I_]
    return out


def _regs(acc, s):
    """the B operand a lane builds from accumulator registers 8s..8s+7"""
    return np.stack([acc[8 * s + e] for e in range(8)], axis=1)


def test_split_f16_slabs_reproduce_the_mlp():
    g = torch.Generator().manual_seed(3)
    rnd = lambda *s: torch.randn(*s, generator=g) * 0.3
    w0, b0, w1, b1 = rnd(64, 32), rnd(64), rnd(17, 64), rnd(17)
    h0, hb0, h1, hb1, h2, hb2 = rnd(64, 31), rnd(64), rnd(64, 64), rnd(64), rnd(3, 64), rnd(3)
    blob = ops.pack_field_mfma16(w0, b0, w1, b1, h0, hb0, h1, hb1, h2, hb2)
    assert blob.numel() == ops.MFMA16_BLOB_FLOATS == 11684          # include/unerf.h: UNERF_MFMA16_BLOB_FLOATS
    # the tail (bias rows, rgb layer) is the fp32 blob's tail
    assert torch.equal(blob[ops.MFMA_BIAS_OFF:ops.MFMA_BLOB_FLOATS],
                       ops.pack_field_mfma(w0, b0, w1, b1, h0, hb0, h1, hb1, h2, hb2)[ops.MFMA_BIAS_OFF:])
    sl = _slabs16(blob, ops.MF16_SLABS)
    bias = blob[ops.MFMA_BIAS_OFF:ops.MFMA_H2_OFF].numpy().astype(np.float64).reshape(7, 2, 16)
    binit = lambda k: np.stack([bias[k, H_, r] for r in range(16)])
    feats, sh = rnd(32, 32).double().numpy(), rnd(32, 16).double().numpy()
    j = I_
    feat = np.stack([feats[j, 16 * H_ + s] for s in range(16)])                  # lane (j,h): features 16h + 0..15
    hid = [binit(0), binit(1)]
    for st in range(2):
        for b in range(2):
            hid[b] = mfma16(sl[2 * st + b], _regs(feat, st), hid[b])
    hid = [np.maximum(a, 0) for a in hid]
    t = binit(2)
    for st in range(4):
        t = mfma16(sl[4 + st], _regs(hid[st >> 1], st & 1), t)
    shv = np.stack([sh[j, 8 * H_ + e] for e in range(8)], axis=1)
    c = [binit(3), binit(4)]
    for b in range(2):
        c[b] = mfma16(sl[8 + b], _regs(t, 0), c[b])
        c[b] = mfma16(sl[10 + b], shv, c[b])
    c = [np.maximum(a, 0) for a in c]
    d = [binit(5), binit(6)]
    for st in range(4):
        for b in range(2):
            d[b] = mfma16(sl[12 + 2 * st + b], _regs(c[st >> 1], st & 1), d[b])
    trunk = np.zeros((32, 32))
    hidden2 = np.zeros((32, 64))
    for r in range(16):
        trunk[j, unit(r, H_)] = t[r]
        for b in range(2):
            hidden2[j, 32 * b + unit(r, H_)] = np.maximum(d[b][r], 0)
    hid_ref = F.relu(F.linear(torch.from_numpy(feats), w0.double(), b0.double()))
    t_ref = F.linear(hid_ref, w1.double(), b1.double())
    x = torch.cat([torch.from_numpy(sh), t_ref[:, 1:16]], dim=-1)
    x = F.relu(F.linear(x, h0.double(), hb0.double()))
    x = F.relu(F.linear(x, h1.double(), hb1.double()))
    # hi + lo carries each weight to ~2^-22: sums of 64 products agree to ~1e-6
    np.testing.assert_allclose(trunk[:, :17], t_ref.numpy(), rtol=0, atol=3e-6)
    assert np.all(trunk[:, 17:] == 0)
    np.testing.assert_allclose(hidden2, x.numpy(), rtol=0, atol=5e-6)
    # the 64 -> 3 colour layer of the "f16" form: four single-operand (hi only) slabs behind the fp32 tail, rows 0..2
    c2 = blob[ops.MFMA_BLOB_FLOATS:].contiguous().view(torch.int16).view(torch.float16).view(4, 64, 8).to(torch.float64).numpy()
    o4 = np.zeros((16, 64))
    dr = [np.maximum(a, 0) for a in d]
    for st in range(4):
        o4 = mfma16(c2[st], _regs(dr[st >> 1], st & 1), o4)
    rgb_pre = np.zeros((32, 32))
    for r in range(16):
        rgb_pre[j, unit(r, H_)] = o4[r]
    ref = F.linear(torch.from_numpy(hidden2), h2.to(torch.float16).double()).numpy()     # f16-rounded weights, no bias
    np.testing.assert_allclose(rgb_pre[:, :3], ref, rtol=0, atol=1e-9)
    assert np.all(rgb_pre[:, 3:] == 0)


def test_folded_trunk_slabs_give_both_weight_halves_from_one_mfma():
    """fold_trunk (the MCDROPOUT kernels' 16-row trunk-out layer): the slab's second operand = rows 0..15 W_hi, rows
    16..31 W_lo, so MFMA(second, a_hi) + MFMA(first, a_lo), registers r + 8 added onto r, is W a to ~2^-22"""
    g = torch.Generator().manual_seed(5)
    rnd = lambda *s: torch.randn(*s, generator=g) * 0.3
    w0, b0, w1, b1 = rnd(64, 32), rnd(64), rnd(16, 64), rnd(16)
    h0, hb0, h1, hb1, h2, hb2 = rnd(64, 31), rnd(64), rnd(64, 64), rnd(64), rnd(3, 64), rnd(3)
    plain = ops.pack_field_mfma16(w0, b0, w1, b1, h0, hb0, h1, hb1, h2, hb2)
    fold = ops.pack_field_mfma16(w0, b0, w1, b1, h0, hb0, h1, hb1, h2, hb2, fold_trunk=True)
    raw = lambda blob: blob[:ops.MF16_SLABS * ops.MF16_SLAB_FLOATS].contiguous().view(torch.int16).view(torch.float16).view(
        ops.MF16_SLABS, 2, 64, 8)
    rp, rf = raw(plain), raw(fold)
    keep = [s for s in range(ops.MF16_SLABS) if not 4 <= s < 8]
    assert torch.equal(rp[keep], rf[keep]) and torch.equal(rp[4:8, 0], rf[4:8, 0]) and torch.equal(plain[ops.MFMA_BIAS_OFF:], fold[ops.MFMA_BIAS_OFF:])
    lane = torch.arange(64)
    lower, upper = (lane & 31) < 16, (lane & 31) >= 16
    assert torch.equal(rf[4:8, 1][:, lower], rp[4:8, 0][:, lower])                       # rows 0..15: W_hi again
    assert torch.equal(rf[4:8, 1][:, upper], rp[4:8, 1][:, lower])                       # rows 16..31: W_lo of row - 16
    assert (rp[4:8, :, upper] == 0).all()                                                # 16 output rows only
    # emulate the two MFMAs per k-step on hi / lo activation halves
    hid = (torch.randn(32, 64, generator=g, dtype=torch.float64) * 0.7).clamp_min(0)
    hid_hi = hid.to(torch.float16).double()
    hid_lo = (hid - hid_hi).to(torch.float16).double()
    j = I_
    acc = np.zeros((16, 64))
    first, mixed = rf[:, 0].double().numpy(), rf[:, 1].double().numpy()
    for st in range(4):
        regs = lambda v: np.stack([v[j, 32 * (st >> 1) + unit(8 * (st & 1) + e, H_)] for e in range(8)], axis=1)
        acc = mfma16(first[4 + st], regs(hid_lo.numpy()), acc)
        acc = mfma16(mixed[4 + st], regs(hid_hi.numpy()), acc)
    acc[:8] += acc[8:]
    t = np.zeros((32, 16))
    for r in range(8):
        t[j, unit(r, H_)] = acc[r]
    ref = hid @ w1.double().T
    np.testing.assert_allclose(t, ref.numpy(), rtol=0, atol=3e-6)
    with pytest.raises(AssertionError):                                                  # 17 rows (ACTIVE) cannot fold
        ops.pack_field_mfma16(w0, b0, rnd(17, 64), rnd(17), h0, hb0, h1, hb1, h2, hb2, fold_trunk=True)


def test_laplace_heads_with_the_base_change_folded_into_the_rows():
    """exp2_rows (UNERF_BUILD_LAP_EXP2): density rows x log2 e, colour rows x -log2 e (weights and bias), padded rows
    keep a bias that makes exp2 / (1 / (1 + exp2)) vanish; softplus density rows stay unscaled"""
    g = torch.Generator().manual_seed(6)
    rnd = lambda *s: torch.randn(*s, generator=g) * 0.3
    n = 100
    ws_d, ws_r = rnd(n, 65), rnd(n, 195)
    plain = ops.pack_laplace_heads16(ws_d, ws_r)
    sd, sr = (ws_d.double() * ops.LOG2E).float(), (ws_r.double() * -ops.LOG2E).float()
    scaled = ops.pack_laplace_heads16(ws_d, ws_r, exp2_rows=True)
    want = ops.pack_laplace_heads16(sd, sr)
    assert torch.equal(scaled[:ops.LAP_BIAS_OFF], want[:ops.LAP_BIAS_OFF])
    tail = want[ops.LAP_BIAS_OFF:].view(4, ops.LAP_BLOCKS, 2, 16).clone()
    assert (tail == ops.LAP_PAD_BIAS).sum() == 4 * (128 - n)
    tail[1:][tail[1:] == ops.LAP_PAD_BIAS] = -ops.LAP_PAD_BIAS
    assert torch.equal(scaled[ops.LAP_BIAS_OFF:], tail.reshape(-1))
    soft = ops.pack_laplace_heads16(ws_d, ws_r, exp2_rows=True, softplus=True)
    q = ops.LAP_BIAS_OFF // 4                                   # fragments of head 0 (density)
    assert torch.equal(soft[:q], plain[:q]) and torch.equal(soft[q:ops.LAP_BIAS_OFF], scaled[q:ops.LAP_BIAS_OFF])


def test_split_f16_halves_are_a_22_bit_representation():
    g = torch.Generator().manual_seed(4)
    w = torch.randn(4096, generator=g) * torch.logspace(-3, 2, 4096)
    hi, lo = ops._split_f16(w)
    err = (w.double() - (hi.double() + lo.double())).abs()
    assert (err <= w.abs().double() * 2.0 ** -21 + 3.1e-8).all()
    # weights beyond the f16 range make the packer decline (the caller stays on the exact kernels)
    big = torch.randn(64, 32, generator=g)
    big[3, 5] = 7.0e4
    rnd = lambda *s: torch.randn(*s, generator=g)
    assert ops.pack_field_mfma16(big, rnd(64), rnd(16, 64), rnd(16), rnd(64, 31), rnd(64), rnd(64, 64), rnd(64), rnd(3, 64), rnd(3)) is None


def test_split_f16_laplace_heads_layout():
    g = torch.Generator().manual_seed(5)
    rnd = lambda *s: torch.randn(*s, generator=g) * 0.3
    n = 100
    ws_d, ws_r = rnd(n, 65), rnd(n, 195)
    lap = ops.pack_laplace_heads16(ws_d, ws_r)
    assert lap.numel() == ops.LAP_BLOB_FLOATS
    assert torch.equal(lap[ops.LAP_BIAS_OFF:], ops.pack_laplace_heads(ws_d, ws_r)[ops.LAP_BIAS_OFF:])
    sl = _slabs16(lap, 4 * ops.LAP_BLOCKS * 4).reshape(4, ops.LAP_BLOCKS, 4, 64, 8)
    x = rnd(32, 64).double().numpy()                      # 64 hidden units of 32 samples
    xin = [np.zeros((16, 64)), np.zeros((16, 64))]        # as two accumulator blocks
    for b in range(2):
        for r in range(16):
            xin[b][r] = x[I_, 32 * b + unit(r, H_)]
    for q, W in enumerate([ws_d[:, :64]] + [ws_r[:, c * 64:(c + 1) * 64] for c in range(3)]):
        for blk in range(ops.LAP_BLOCKS):
            acc = np.zeros((16, 64))
            for st in range(4):
                acc = mfma16(sl[q, blk, st], _regs(xin[st >> 1], st & 1), acc)
            rows = np.zeros((32, 32))                     # [sample, row-in-block]
            for r in range(16):
                rows[I_, unit(r, H_)] = acc[r]
            Wp = np.zeros((128, 64))
            Wp[:n] = W.double().numpy()
            np.testing.assert_allclose(rows, x @ Wp[32 * blk:32 * blk + 32].T, rtol=0, atol=3e-6)


@pytest.mark.parametrize("n,exp2_rows,softplus", [(100, True, False), (100, False, False), (37, True, True), (128, True, False)])
def test_batched_laplace_packer_equals_the_per_set_packers(n, exp2_rows, softplus):
    """ops.pack_laplace_sets (one gather per blob, on the device; the per-chunk draws make 64 sets per 1080p frame) is the
    loops of pack_laplace_heads / pack_laplace_heads16 as index maps: every set bit for bit"""
    g = torch.Generator().manual_seed(9)
    T = 3
    ws_d, ws_r = torch.randn(T, n, 65, generator=g) * 0.3, torch.randn(T, n, 195, generator=g) * 0.3
    blob, blob16 = ops.pack_laplace_sets(ws_d, ws_r, "cpu", exp2_rows=exp2_rows, softplus=softplus)
    assert blob.shape == blob16.shape == (T, ops.LAP_BLOB_FLOATS) and blob16.is_contiguous()
    for t in range(T):
        assert torch.equal(blob[t], ops.pack_laplace_heads(ws_d[t], ws_r[t]))
        want = ops.pack_laplace_heads16(ws_d[t], ws_r[t], exp2_rows=exp2_rows, softplus=softplus)
        assert torch.equal(blob16[t].view(torch.int32), want.view(torch.int32))      # bit patterns (f16 pairs read as fp32)
    ws_r[1, 5, 7] = 7.0e4                                                            # beyond the f16 range: no f16 blob
    assert ops.pack_laplace_sets(ws_d, ws_r, "cpu")[1] is None
