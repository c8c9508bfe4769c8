#!/usr/bin/env python3
"""One-off fuzz of the whole path (analyze + search + epilogue through the needle-capi objects) against the oracle (test
infrastructure; run on the GPU box): random libraries -- 2 .. 12 episodes of 60 .. 400 s, shared intros of 0 .. 60 s at the
synthesizer's varying offsets, extra spliced-in common segments between random episode pairs (competing candidates for
find_best_match and the heap order), random hash-match thresholds, minimum opening durations and time padding.
usage: fuzz_pipeline.py [cases=60] [seed=1]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from needle_amd import capi, synth  # noqa: E402
from oracle import oracle as O  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = found = 0
for case in range(cases):
    n = int(rng.integers(2, 13))
    seconds = float(rng.uniform(60, 400))
    intro = float(rng.choice([0.0, 8.0, 25.0, 45.0, 60.0]))
    eps = synth.make_library(n, seconds, min(intro, seconds / 4), seed_base=int(rng.integers(1, 2 ** 31)))
    pcms = [e.pcm.copy() for e in eps]
    for _ in range(int(rng.integers(0, 4))):                       # a second common segment between two episodes
        a, b = rng.choice(n, 2, replace=False)
        L = int(rng.uniform(5, 50) * 11025)
        half = len(pcms[0]) // 2
        if L >= half - 10:
            continue
        ia, ib = int(rng.integers(0, half - L)), int(rng.integers(0, half - L))
        pcms[b][ib:ib + L] = pcms[a][ia:ia + L]
    thr = int(rng.integers(5, 17))
    min_open = int(rng.choice([5, 10, 20, 30, 40]))
    pad = float(rng.choice([0.0, 0.0, 0.5, 2.0]))
    paths = [f"/tmp/needle_fuzz_{case}_{k}.wav" for k in range(n)]
    fhs = capi.Analyzer.from_files(paths).run_pcm(pcms, channels=1)
    hd = O.duration_from_secs_f32(0.3)
    ref = O.analyze_batch([p[: len(p) // 2] for p in pcms], 1, hd, threads=8)
    ok = True
    for got, want in zip(fhs, ref):
        h, ts = got.opening_data()
        if h.tolist() != [x for x, _ in want.opening] or ts.tolist() != [t for _, t in want.opening]:
            ok = False
    cmp_gpu = capi.Comparator.from_files(paths).with_hash_match_threshold(thr).with_min_opening_duration(min_open).with_time_padding(pad)
    res = cmp_gpu.run_with_frame_hashes(fhs)
    want = O.run_with_frame_hashes(O.Comparator(hash_match_threshold=thr, min_opening_duration=min_open * O.NS,
                                                time_padding=O.duration_from_secs_f32(pad)), ref)
    got = [None if r is None else (r.opening, r.ending) for r in res]
    exp = [None if r is None else (r.opening, r.ending) for r in want]
    found += sum(1 for r in exp if r is not None and r[0] is not None)
    if not ok or got != exp:
        bad += 1
        print(f"MISMATCH case {case}: n {n} seconds {seconds:.0f} intro {intro} thr {thr} min_open {min_open} pad {pad}: hashes ok {ok}\n  got {got}\n  exp {exp}")
    if case % 10 == 0:
        print(f"case {case}: n {n}, {found} openings found so far, {bad} bad", flush=True)
print(f"{cases} cases, {found} openings found by the oracle, {bad} mismatching cases")
sys.exit(1 if bad else 0)
