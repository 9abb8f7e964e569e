"""Host logic: the MFMA operand packer (ops.pack_field_mfma) arranges the MLP weights so that
field_kernel_mfma's dataflow computes the same network.  The kernel's dataflow is emulated here in
numpy with the documented v_mfma_f32_32x32x2_f32 lane maps (cdna_hip_programming.md section 3):
A[i=l&31][k=l>>5], B[k=l>>5][j=l&31], D row = (r&3)+8(r>>2)+4(l>>5), col = l&31."""
import numpy as np
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
