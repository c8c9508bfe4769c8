#!/usr/bin/env python3
"""One-off fuzz of the fingerprinter (f32 first pass + certification + f64 recomputation) against oracle/ora_chromaprint.c
(test infrastructure; run on the GPU box): random signal families -- noise at random levels, tone stacks, chirps, gated
bursts, clipped mixtures, near-silence, a strong tone outside chromaprint's band over a weak one inside --, mono / stereo,
ragged batches, step 1..3.  usage: fuzz_fingerprint.py [cases=200] [seed=1]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from needle_amd import capi  # noqa: E402
from oracle import oracle as O  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)


def signal(n):
    if n == 0:
        return np.zeros(0, dtype=np.int16)
    t = np.arange(n) / 11025.0
    kind = int(rng.integers(0, 8))
    if kind == 0:
        x = rng.normal(0, rng.choice([0.5, 3, 50, 2000, 12000]), n)
    elif kind == 1:
        x = sum(rng.uniform(100, 9000) * np.sin(2 * np.pi * rng.uniform(30, 5400) * t + rng.uniform(0, 6)) for _ in range(int(rng.integers(1, 7))))
    elif kind == 2:
        f0, f1 = rng.uniform(20, 3000, 2)
        x = rng.uniform(500, 30000) * np.sin(2 * np.pi * (f0 * t + 0.5 * (f1 - f0) / max(t[-1], 1e-3) * t * t))
    elif kind == 3:
        x = rng.normal(0, 4000, n) * (np.sin(2 * np.pi * rng.uniform(0.2, 3) * t) > rng.uniform(-0.5, 0.8))
    elif kind == 4:
        x = 5 * (rng.uniform(2000, 9000) * np.sin(2 * np.pi * rng.uniform(100, 2000) * t) + rng.normal(0, 5000, n))
    elif kind == 5:
        x = rng.normal(0, rng.uniform(0.2, 2.0), n)
    elif kind == 6:
        x = rng.uniform(3, 300) * np.sin(2 * np.pi * rng.uniform(100, 3000) * t) + 30000 * np.sin(2 * np.pi * rng.uniform(4000, 5400) * t)
    else:
        x = np.zeros(n)
        a = int(rng.integers(0, max(n // 2, 1)))
        x[a:] = rng.uniform(1000, 20000) * np.sin(2 * np.pi * rng.uniform(50, 3000) * t[a:])
    return np.clip(np.rint(x), -32768, 32767).astype(np.int16)


bad = items = 0
capi.cert_stats(reset=True)
for case in range(cases):
    ch = int(rng.integers(1, 3))
    step = int(rng.integers(1, 4))
    pcms, want = [], []
    for _ in range(int(rng.integers(1, 6))):
        n = int(rng.choice([0, 4095, 4096 + 1365 * 19, int(rng.integers(30000, 11025 * 40))]))
        mono = signal(n)
        if ch == 2:
            other = np.clip(mono.astype(np.int32) + rng.integers(-40, 41, n), -32768, 32767).astype(np.int16)
            p = np.stack([mono, other], axis=1).reshape(-1)
            total = mono.astype(np.int32) + other.astype(np.int32)
            mono = np.where(total < 0, -((-total) // 2), total // 2).astype(np.int16)   # (L + R) / 2 with C truncation
        else:
            p = mono
        pcms.append(p)
        want.append(O.fingerprint(mono)[::step])
    got = capi.fingerprint(pcms, step=step, channels=ch)
    for k, (a, w) in enumerate(zip(got, want)):
        items += len(w)
        if a.tolist() != w.tolist():
            bad += 1
            print(f"MISMATCH case {case} stream {k}: ch {ch} step {step} len {len(pcms[k]) // ch}: "
                  f"{int((np.asarray(a) != np.asarray(w)).sum()) if len(a) == len(w) else 'length'} differ")
    if case % 20 == 0:
        print(f"case {case}: {items} items so far, {bad} bad streams", flush=True)
st = capi.cert_stats(reset=True)
print(f"{cases} cases, {items} items, {bad} mismatching streams; recomputed in f64: {st['items_recomputed']} of {st['items']} raw items")
sys.exit(1 if bad else 0)
