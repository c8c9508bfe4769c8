#!/usr/bin/env python3
"""BASELINE.json configs[2]-style measurement: search-only over precomputed hashes (N episodes, 24-min or 45-min
sized hash columns with a planted shared intro), all pairs, on one GPU, through needle_hip_hamming_runs_device +
the host epilogue.  Not the headline metric; numbers go to DESIGN.md."""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from needle_amd import capi  # noqa: E402

capi.set_kernel_timing("all")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--episodes", type=int, default=280)
    ap.add_argument("--hashes", type=int, default=2897)
    ap.add_argument("--intro", type=int, default=360)
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    n, h = args.episodes, args.hashes
    rng = np.random.default_rng(7)
    intro = rng.integers(0, 2 ** 32, args.intro, dtype=np.uint64).astype(np.uint32)
    arena = rng.integers(0, 2 ** 32, (n, h), dtype=np.uint64).astype(np.uint32)
    for v in range(n):
        a = 40 + (37 * v) % 900
        flips = (np.uint32(1) << rng.integers(0, 32, args.intro).astype(np.uint32)) * (rng.random(args.intro) < 0.7)
        arena[v, a:a + args.intro] = intro ^ flips
    L = capi.lib()
    seqs = (capi.Seq * n)(*[capi.Seq(v * h, h) for v in range(n)])
    pairs = [(i, j) for i in range(n) for j in range(i + 1, n)]
    probs = (capi.Problem * len(pairs))(*[capi.Problem(i, j, 82, p) for p, (i, j) in enumerate(pairs)])
    d_arena = capi.DeviceBuffer(arena.nbytes)
    capi.check(L.needle_hip_memcpy_h2d(d_arena.ptr, arena.ctypes.data, arena.nbytes))
    cap = 4 * len(pairs) + 1024
    d_runs, d_count = capi.DeviceBuffer(cap * capi.RUN_DTYPE.itemsize), capi.DeviceBuffer(4)
    times, kms = [], []
    for rep in range(args.reps + 1):
        t0 = time.perf_counter()
        capi.check(L.needle_hip_hamming_runs_device(d_arena.ptr, seqs, n, probs, len(pairs), 10, d_runs.ptr, cap,
                                                    d_count.ptr, True))
        found = int(d_count.to_host(np.uint32, 1)[0])
        runs = d_runs.to_host(capi.RUN_DTYPE, min(found, cap))
        dt = time.perf_counter() - t0
        if rep:
            times.append(dt)
            kms.append(capi.last_kernel_ms("hamming_runs"))
    cells = len(pairs) * float(h) * h
    print(f"{n} episodes x {h} hashes: {len(pairs)} pairs, {cells:.3e} table cells, {found} runs")
    print(f"  scan kernel {np.mean(kms):.3f} ms  ({len(pairs) / np.mean(kms) * 1e3:.3e} pairs/s, "
          f"{cells / np.mean(kms) * 1e3:.3e} cell-equivalents/s)")
    print(f"  launch + scan + simhash + D2H of runs: {1e3 * np.mean(times):.3f} ms ({len(pairs) / np.mean(times):.3e} pairs/s)")
    assert found >= len(pairs), "every pair shares the planted intro"


if __name__ == "__main__":
    main()
