#!/usr/bin/env python
"""Search over two-instruction MC-dropout mask steps x -> rotr(x, r) * (2^s + 1) (experiment; the step the kernels
use, unerf_mask_step, is (r, s) = (22, 6)).  Scores every (r, s) by the statistics tests/test_golden_cpu.py applies
(keep rate, correlation between all pairs of K = 10 passes in either half, between halves, between neighbouring
words, Binomial count of keeps), in units of the 5-sigma tolerance: everything below 1 passes.

    python benchmarks/mask_step_search.py [n_samples]     (8000 samples: ~8 minutes on 8 cores)
"""
import os
import sys
from math import comb

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import nerf_oracle as O   # noqa: E402

U = np.uint32


def rotr(x, r):
    return (x >> U(r)) | (x << U(32 - r))


def step(r, s):
    def f(x):
        with np.errstate(over="ignore"):
            y = rotr(x, r)
            return y + (y << U(s))
    return f


def words0(n):
    sidx = np.arange(n, dtype=np.int64) * 7 + 11
    base = O.mc_base(1234, 0, sidx)[:, None]
    j = np.arange(32, dtype=U)[None, :]
    with np.errstate(over="ignore"):
        r = O._hash32(base + (j + U(1)) * O.GOLDEN)
    return np.where(r == 0, O.GOLDEN, r).astype(U)


def evaluate(fn, K=10, n=62500, p=0.2):
    thr = int(round((1 - p) * 65536))
    r, keeps = words0(n), []
    for _ in range(K):
        lo, hi = (r & U(0xFFFF)) ^ U(0x8000), (r >> U(16)) ^ U(0x8000)
        keeps.append(np.stack([lo < thr, hi < thr], -1).reshape(n, 64))
        r = fn(r)
    keeps = np.stack(keeps)
    N = n * 64
    tol = 5 / np.sqrt(N)

    def corr(a, b):
        return abs(np.corrcoef(a.reshape(-1).astype(np.float64), b.reshape(-1).astype(np.float64))[0, 1])
    res = {"rate": np.abs(keeps.reshape(K, -1).mean(1) - (1 - p)).max() / (tol * 0.5)}
    m = 0
    for a in range(K):
        for b in range(a + 1, K):
            m = max(m, corr(keeps[a], keeps[b]), corr(keeps[a][:, 0::2], keeps[b][:, 1::2]) / 1.5,
                    corr(keeps[a][:, 1::2], keeps[b][:, 0::2]) / 1.5)
    res["lag"] = m / tol
    res["halves"] = corr(keeps[:, :, 0::2], keeps[:, :, 1::2]) / tol
    res["neigh"] = corr(keeps[:, :, :-2], keeps[:, :, 2:]) / tol
    cnt = keeps[:8].sum(0).reshape(-1)
    hist = np.bincount(cnt, minlength=9) / N
    binom = np.array([comb(8, i) * 0.8 ** i * 0.2 ** (8 - i) for i in range(9)])
    res["binom"] = np.abs(hist - binom).max() / tol
    return res


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
    best = []
    for r in range(1, 32):
        for s in range(1, 16):
            e = evaluate(step(r, s), n=n)
            best.append((max(e.values()), r, s, e))
    best.sort(key=lambda t: t[0])
    for b in best[:15]:
        print(b[1:3], {k: round(float(v), 2) for k, v in b[3].items()})
