#!/usr/bin/env python3
"""One-off fuzz of the DEVICE epilogue (needle_amd/csrc/epilogue.hip) against the host form and the oracle (test
infrastructure; run on the GPU box): random libraries of 2 - 36 videos with ragged lengths, hashes written straight into
the library's arena -- random rows with shared segments planted bit-identically or with a few flipped bits, so that pairs
have several runs and candidates tie --, endings on or off, thresholds 4 - 14, minimum durations 3 - 30 s, padding 0 - 2 s.
Every case: job through the device epilogue == job through the host epilogue; every fourth case also == the oracle's
run_with_frame_hashes.  usage: fuzz_epilogue.py [cases=200] [seed=1]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from needle_amd import capi, synth  # noqa: E402
from oracle import oracle as O  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
NS = O.NS


def results(lib, cmp, device):
    os.environ["NEEDLE_HIP_DEVICE_EPILOGUE"] = "1" if device else "0"
    lib.job_begin(cmp, 0)
    res, runs = lib.job_end(cmp, 0)
    return [None if r is None else (r.opening, r.ending) for r in res], runs


bad = compared = oracle_checked = 0
for case in range(cases):
    n = int(rng.integers(2, 37))
    endings = bool(rng.random() < 0.4)
    seconds = [float(rng.uniform(60, 260)) for _ in range(n)]
    lens = [int(round(s * synth.RATE)) for s in seconds]
    lib = capi.Library(n)
    if endings:
        lib.include_endings()
    lib.stream_pcm([np.zeros(v, dtype=np.int16) for v in lens], lens)
    R = lib.rows_per_video()
    d_arena, stride = lib.hash_arena()
    fh0 = [lib.frame_hashes(v) for v in range(n)]
    kept = [[len(f.opening_data()[0]), len(f.ending_data()[0]) if endings else 0] for f in fh0]
    segs = [rng.integers(0, 2 ** 32, int(rng.integers(20, 160)), dtype=np.uint64).astype(np.uint32) for _ in range(int(rng.integers(1, 5)))]
    exact = rng.random() < 0.6
    rows = {}
    for v in range(n):
        for r in range(R):
            k = kept[v][r]
            h = rng.integers(0, 2 ** 32, max(k, 1), dtype=np.uint64).astype(np.uint32)[:k]
            for seg in segs:
                if rng.random() < 0.7 and len(seg) + 2 < k:
                    a = int(rng.integers(1, k - len(seg)))
                    flips = np.zeros(len(seg), dtype=np.uint32) if exact else \
                        ((np.uint32(1) << rng.integers(0, 32, len(seg)).astype(np.uint32)) * (rng.random(len(seg)) < 0.5)).astype(np.uint32)
                    h[a:a + len(seg)] = seg ^ flips
            rows[(v, r)] = np.ascontiguousarray(h)
            if k:
                capi.check(capi.lib().needle_hip_memcpy_h2d(d_arena + 4 * (v * R + r) * stride, h.ctypes.data, h.nbytes))
    thr = int(rng.integers(4, 15))
    min_o, min_e = int(rng.integers(3, 31)), int(rng.integers(3, 31))
    pad = float(rng.choice([0.0, 0.0, 0.25, 1.0, 2.0]))
    cmp = capi.Comparator([f"v{v}.wav" for v in range(n)], include_endings=endings, hash_match_threshold=thr,
                          min_opening_duration=min_o, min_ending_duration=min_e, time_padding=pad)
    try:
        host, runs_h = results(lib, cmp, False)
    except capi.NeedleError as e:
        try:
            results(lib, cmp, True)
            print(f"MISMATCH case {case}: the host form failed ({e}) and the device form did not")
            bad += 1
        except capi.NeedleError:
            pass
        continue
    dev, runs_d = results(lib, cmp, True)
    compared += 1
    if dev != host or runs_h != runs_d:
        bad += 1
        print(f"MISMATCH case {case}: n {n} endings {endings} thr {thr} min {min_o}/{min_e} pad {pad} exact {exact}: "
              f"{sum(a != b for a, b in zip(dev, host))} videos differ")
    if case % 4 == 0:
        hd = O.duration_from_secs_f32(0.3)
        ofh = []
        for v in range(n):
            f = lib.frame_hashes(v)
            op = list(zip(f.opening_data()[0].tolist(), f.opening_data()[1].tolist()))
            en = list(zip(f.ending_data()[0].tolist(), f.ending_data()[1].tolist())) if endings else []
            ofh.append(O.FrameHashes(op, en, hd, ""))
        try:
            want = O.run_with_frame_hashes(O.Comparator(include_endings=endings, hash_match_threshold=thr, min_opening_duration=min_o * NS,
                                                        min_ending_duration=min_e * NS, time_padding=O.duration_from_secs_f32(pad)),
                                           ofh, threads=8)
            oracle_checked += 1
            if dev != [None if r is None else (r.opening, r.ending) for r in want]:
                bad += 1
                print(f"MISMATCH vs oracle, case {case}")
        except Exception as e:                                      # noqa: BLE001 -- the oracle refuses what the reference panics on
            print(f"case {case}: oracle raised {type(e).__name__}: skipped")
    if case % 25 == 0:
        print(f"case {case}: {compared} compared, {oracle_checked} also against the oracle, {bad} bad", flush=True)
print(f"{cases} cases: {compared} device-vs-host comparisons, {oracle_checked} against the oracle, {bad} mismatches")
sys.exit(1 if bad else 0)
