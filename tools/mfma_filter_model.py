#!/usr/bin/env python3
"""CPU model (numpy) of a matrix-pipe first stage for the sampled scan -- NOT product code.

The sampled scan (search.hip) tests, for every aligned window of W rows and every diagonal, H head cells on the vector
ALU (~2.7 instructions per cell) and is on its integer-VALU roof at library scale.  The Hamming distances of a window's
head rows against every destination position are an integer matrix product: with a hash as 32 values of +-1,
dot(a, b) = 32 - 2 d(a, b); the H head rows side by side give K = 32 H and one product per (window, destination position)
that holds 32 H - 2 * (SUM of the H distances) -- the shape of v_mfma_i32_32x32x32_i8 with K = 128 at H = 4.  The sum is a
necessary condition only (every row <= t implies sum <= H t), so what passes has to be verified cell by cell.

This model runs exactly that on the oracle's hashes of a small synthetic library:
  1. S[k, j] = sum of the head-row distances through an int8 matrix product (what the matrix pipe would deliver),
  2. survivors S <= H t inside the table -> all W cells tested exactly -> maximal run resolved as search.hip does,
  3. the run list compared with the oracle's table-free scan (ora_diagonal_runs_all_pairs): must be identical,
and prints how many (window, diagonal) outputs pass the sum filter, the exact head test and the whole window -- the numbers
that size the survivor path of such a kernel.

usage: python tools/mfma_filter_model.py [episodes=8] [minutes=24] [H=4]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from needle_amd import synth  # noqa: E402
from oracle import oracle  # noqa: E402

W = 8
POPC = np.array([bin(i).count("1") for i in range(256)], dtype=np.uint8)


def popcount32(x):
    x = x.astype(np.uint32)
    return (POPC[x & 0xFF].astype(np.int32) + POPC[(x >> 8) & 0xFF] + POPC[(x >> 16) & 0xFF] + POPC[(x >> 24) & 0xFF])


def pm1(h):
    """uint32 hashes [n] -> int8 [n, 32] of +-1 (bit set: +1)."""
    bits = ((h[:, None] >> np.arange(32, dtype=np.uint32)[None, :]) & 1).astype(np.int8)
    return (2 * bits - 1).astype(np.int8)


def head_rows(h):
    return [0, W // 2, W - 1] if h == 3 else [(k * (W - 1)) // (h - 1) for k in range(h)]


def scan_pair(src, dst, thr, min_len, H, stats):
    n, m = len(src), len(dst)
    P = min_len - W + 1
    hr = head_rows(H)
    a_pm, b_pm = pm1(src), pm1(dst)
    runs = set()
    w0s = [w0 for w0 in range(1, n, P) if w0 + W - 1 <= n - 1]
    if not w0s:
        return runs
    # A: one row per window = its H head hashes side by side (K = 32 H); B: one column per destination position j = the
    # hashes dst[j + s] for the same head rows s (zero beyond the end: those outputs are outside the table anyway)
    A = np.concatenate([a_pm[[w0 + s for w0 in w0s]] for s in hr], axis=1).astype(np.int32)
    cols = []
    for s in hr:
        shifted = np.zeros((m, 32), dtype=np.int8)
        shifted[: m - s] = b_pm[s:]
        cols.append(shifted)
    B = np.concatenate(cols, axis=1).astype(np.int32).T          # [32 H, m]
    dots = A @ B                                                    # the matrix pipe's output: 32 H - 2 sum
    S = (32 * H - dots) // 2
    for row, w0 in enumerate(w0s):
        # destination position j of the window's first row <-> diagonal d = j - w0; inside the table: j >= 1, w0 + W - 1 + d <= m - 1
        j = np.arange(m)
        inside = (j >= 1) & (j + W - 1 <= m - 1)
        stats["outputs"] += int(inside.sum())
        passed = inside & (S[row] <= H * thr)
        stats["sum_pass"] += int(passed.sum())
        for jj in np.nonzero(passed)[0]:
            d = int(jj) - w0
            cells = popcount32(src[w0:w0 + W] ^ dst[w0 + d:w0 + d + W])
            if (cells[hr] <= thr).all():
                stats["head_pass"] += 1
            if not (cells <= thr).all():
                continue
            stats["window_pass"] += 1
            ilo, ihi = (1 - d if d < 0 else 1), min(n - 1, m - 1 - d)
            a = w0
            while a - 1 >= ilo and popcount32(src[a - 1:a] ^ dst[a - 1 + d:a + d])[0] <= thr:
                a -= 1
            b = w0 + W - 1
            while b + 1 <= ihi and popcount32(src[b + 1:b + 2] ^ dst[b + 1 + d:b + 2 + d])[0] <= thr:
                b += 1
            if b - a + 1 >= min_len:
                runs.add((b, b + d, b - a + 1))
    return runs


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    minutes = float(sys.argv[2]) if len(sys.argv) > 2 else 24.0
    H = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    thr, hash_s = 10, 0.3
    eps = synth.make_library(n, minutes * 60.0, 90.0)
    half = [e.pcm[: len(e.pcm) // 2] for e in eps]
    fhs = oracle.analyze_batch(half, 1, int(hash_s * 1e9), threads=min(8, n))
    seqs = [np.array([h for h, _ in fh.opening], dtype=np.uint32) for fh in fhs]
    min_len = 82                           # what the default 20 s minimum comes to at 0.3 s hashes (search.hip); model and oracle get the same value
    t0 = time.perf_counter()
    stats = {"outputs": 0, "sum_pass": 0, "head_pass": 0, "window_pass": 0}
    mine = []
    pair = 0
    for i in range(n):
        for k in range(i + 1, n):
            for (se, de, ln) in scan_pair(seqs[i], seqs[k], thr, min_len, H, stats):
                mine.append((pair, se, de, ln))
            pair += 1
    model_s = time.perf_counter() - t0
    total, want = oracle.diagonal_runs_all_pairs(seqs, thr, min_len, threads=8, capacity=1 << 20)
    want = sorted(map(tuple, want.tolist()))
    mine = sorted(mine)
    same = mine == want
    out = {"episodes": n, "minutes": minutes, "hashes_per_episode": int(len(seqs[0])), "H": H, "head_rows": head_rows(H), "threshold": thr,
           "min_len": min_len, "runs_model": len(mine), "runs_oracle": int(total), "identical_run_lists": same,
           "window_diagonal_outputs": stats["outputs"],
           "pass_sum_filter": round(stats["sum_pass"] / max(stats["outputs"], 1), 6),
           "pass_exact_head_rows": round(stats["head_pass"] / max(stats["outputs"], 1), 6),
           "pass_whole_window": round(stats["window_pass"] / max(stats["outputs"], 1), 6), "model_seconds": round(model_s, 1)}
    import json
    print(json.dumps(out))
    return 0 if same else 1


if __name__ == "__main__":
    sys.exit(main())
