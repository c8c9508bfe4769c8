#!/usr/bin/env python3
"""One-off fuzz of the resampler kernels against oracle/ora_resample.c (test infrastructure; run on the GPU box):
random rates among those every kernel family takes, mono / stereo, random stream counts and lengths (empty, shorter than a
window, around tile boundaries), random NEEDLE_HIP_RESAMPLE_GRID (few persistent workgroups).  usage: fuzz_resample.py [cases=150] [seed=1]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from needle_amd import capi  # noqa: E402
from oracle import oracle as O  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
rates = [48000, 48000, 48000, 32000, 24000, 16000, 8000, 44100, 22050, 96000, 12000, 11025]
bad = 0
for case in range(cases):
    rate = int(rng.choice(rates))
    ch = int(rng.integers(1, 3))
    g = np.gcd(11025, rate)
    M = rate // g
    nstreams = int(rng.integers(1, 6))
    pcms = []
    for _ in range(nstreams):
        kind = int(rng.integers(0, 6))
        n = [0, int(rng.integers(1, 300)), 16 * M * int(rng.integers(1, 4)) + int(rng.integers(-3, 4)),
             int(rng.integers(1000, rate * 2)), 16 * M * int(rng.integers(1, 9)), int(rng.integers(rate // 2, rate * 3))][kind]
        n = max(n, 0)
        x = rng.integers(-32768, 32768, n * ch, dtype=np.int32).astype(np.int16) if rng.random() < 0.3 else \
            np.clip(8000 * np.sin(np.arange(n * ch) * rng.uniform(0.001, 0.5)) + rng.normal(0, 3000, n * ch), -32768, 32767).astype(np.int16)
        pcms.append(x)
    grid = rng.choice(["", "1", "2", "5"])
    if grid:
        os.environ["NEEDLE_HIP_RESAMPLE_GRID"] = str(grid)
    else:
        os.environ.pop("NEEDLE_HIP_RESAMPLE_GRID", None)
    got = capi.resample(pcms, ch, rate)
    for k, (a, p) in enumerate(zip(got, pcms)):
        want = O.resample(p, ch, rate)
        if a.tolist() != want.tolist():
            bad += 1
            diff = np.nonzero(np.asarray(a) != np.asarray(want))[0] if len(a) == len(want) else []
            print(f"MISMATCH case {case}: rate {rate} ch {ch} grid {grid!r} stream {k} len {len(p) // ch}: {len(diff)} of {len(want)} differ, first at {diff[:5]}")
    if case % 25 == 0:
        print(f"case {case}: rate {rate} x{ch}, {nstreams} streams, grid {grid!r}: ok so far ({bad} bad)", flush=True)
print(f"{cases} cases, {bad} mismatching streams")
sys.exit(1 if bad else 0)
