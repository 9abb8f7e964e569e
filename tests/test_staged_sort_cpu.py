"""The staged LSD radix pass of unerf_splat.hip (rs_hist / rs_rowscan / rs_scatter_kernel), emulated lane by lane in numpy.

No GPU: this pins the ALGORITHM the kernels implement -- per-chunk digit histogram, exclusive prefix over chunks per digit,
and a scatter in which one 64-lane wave ranks a chunk vector by vector through per-digit "peer words" (every lane ORs its
lane bit into its digit's 64-bit word, reads it back, the first peer advances the digit's running slot by their number),
stages the pairs in digit order and writes each digit's pairs as one run at delta[digit] + slot -- against numpy's stable
argsort, for the tile sort (two passes: low digit, high digit) and the depth sort (four 8-bit passes).  The GPU tests
(tests/test_gpu_splat.py) compare the kernels themselves with rocprim's sorts."""
import numpy as np
import pytest


def staged_pass(keys, vals, shift, B, M, kmax):
    """one stable pass by digit (min(key, kmax) >> shift) & (B - 1); chunks of M pairs, a wave per chunk"""
    n = len(keys)
    nchunk = max(1, -(-n // M))
    digit = lambda k: (np.minimum(k, kmax) >> shift) & (B - 1)
    # rs_hist_kernel: table[digit][chunk]
    table = np.zeros((B, nchunk), np.int64)
    for c in range(nchunk):
        np.add.at(table[:, c], digit(keys[c * M:(c + 1) * M]), 1)
    # rs_rowscan_kernel: exclusive prefix over the chunks per digit (in place) + the digit totals
    dtotal = table.sum(1)
    prefix = np.cumsum(table, 1) - table
    dbase = np.cumsum(dtotal) - dtotal
    out_k, out_v = np.empty_like(keys), np.empty_like(vals)
    for c in range(nchunk):                       # rs_scatter_kernel, one wave
        k, v = keys[c * M:(c + 1) * M], vals[c * M:(c + 1) * M]
        m = len(k)
        cnt = (prefix[:, c + 1] if c + 1 < nchunk else dtotal) - prefix[:, c]      # neighbours of the prefix row, as the kernel does
        lstart = np.cumsum(cnt) - cnt
        cur = lstart.copy()
        delta = dbase + prefix[:, c] - lstart
        pw = np.zeros(B, np.uint64)
        sval, skey = np.empty(m, vals.dtype), np.empty(m, keys.dtype)
        for v0 in range(0, m, 64):                # a vector of 64 pairs
            lanes = np.arange(min(64, m - v0))
            d = digit(k[v0 + lanes])
            for l in lanes:                        # ds_or: every lane ORs its bit into its digit's word
                pw[d[l]] |= np.uint64(1) << np.uint64(l)
            peers = pw[d].copy()                   # read back
            pw[d] = 0                              # cleared
            rank = np.array([bin(int(peers[l]) & ((1 << int(l)) - 1)).count("1") for l in lanes])
            base = cur[d].copy()                   # the digit's running slot, read by all ...
            for l in lanes:                        # ... and advanced by the first of the peers for all of them
                if rank[l] == 0:
                    cur[d[l]] += bin(int(peers[l])).count("1")
            sval[base + rank] = v[v0 + lanes]
            skey[base + rank] = np.minimum(k[v0 + lanes], kmax)
        slot = np.arange(m)                        # write-out: consecutive slots of a digit = consecutive addresses
        dst = delta[digit(skey)] + slot
        out_k[dst], out_v[dst] = skey, sval
    return out_k, out_v


@pytest.mark.parametrize("n,tiles", [(1, 50), (63, 50), (5000, 8160), (4097, 121), (10000, 8160)])
def test_two_pass_tile_sort_is_a_stable_sort(n, tiles):
    rng = np.random.default_rng(n)
    T1 = tiles + 1
    keys = rng.integers(0, T1, n).astype(np.int64)
    keys[rng.random(n) < 0.01] = tiles             # sentinel pairs
    vals = rng.integers(0, 1 << 20, n).astype(np.int64)
    bits = max(1, int(np.ceil(np.log2(T1))))
    b0 = 0 if bits <= 7 else bits // 2
    k, v = keys, vals
    if b0:
        k, v = staged_pass(k, v, 0, 1 << b0, 2048, tiles)
    k, v = staged_pass(k, v, b0, 1 << (bits - b0), 2048, tiles)
    order = np.argsort(keys, kind="stable")
    assert np.array_equal(k, keys[order]) and np.array_equal(v, vals[order])


@pytest.mark.parametrize("n", [3, 1025, 7000])
def test_four_pass_depth_sort_is_a_stable_sort(n):
    rng = np.random.default_rng(n)
    depth = rng.uniform(0.1, 30.0, n).astype(np.float32)
    depth[: n // 3] = depth[n // 3: 2 * (n // 3)]    # equal depths: stability decides
    keys = depth.view(np.uint32).astype(np.int64)
    keys[rng.random(n) < 0.1] = 0xFFFFFFFF           # culled splats last
    vals = np.arange(n, dtype=np.int64)
    k, v = keys, vals
    for p in range(4):
        k, v = staged_pass(k, v, 8 * p, 256, 1024, 0xFFFFFFFF)
    order = np.argsort(keys, kind="stable")
    assert np.array_equal(k, keys[order]) and np.array_equal(v, vals[order])
