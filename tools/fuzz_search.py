#!/usr/bin/env python3
"""One-off fuzz of the search kernels (generic / band / sampled scan + simhash) against the oracle's table walk (test
infrastructure; run on the GPU box): random sequence lengths (2 .. 6000), planted near-duplicate runs incl. runs on the
table's edges and whole diagonals, stretches of one repeated hash (silence), thresholds 0 .. 16, minimum lengths 1 .. 150
(which decides the kernel), several pairs per launch over shared sequences.  Complete run lists incl. both simhashes.
usage: fuzz_search.py [cases=150] [seed=1]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from needle_amd import capi  # noqa: E402
from oracle import oracle as O  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)


def oracle_runs(src, dst, thr, min_len):
    cmp = O.Comparator(hash_match_threshold=thr, min_opening_duration=0)
    ents = O.longest_common_hash_match(cmp, [(int(h), i) for i, h in enumerate(src)], [(int(h), i) for i, h in enumerate(dst)], 0, 0)
    return sorted((e["src_end_idx"], e["dst_end_idx"], e["score"], e["src_match_hash"], e["dst_match_hash"]) for e in ents if e["score"] >= min_len)


bad = runs_total = 0
for case in range(cases):
    nseq = int(rng.integers(2, 5))
    seqs = []
    for _ in range(nseq):
        n = int(rng.choice([2, 3, int(rng.integers(4, 200)), int(rng.integers(200, 3000)), int(rng.integers(3000, 6000))]))
        seqs.append(rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32))
    for _ in range(int(rng.integers(0, 8))):                       # planted runs between random pairs of sequences
        a, b = rng.integers(0, nseq, 2)
        sa, sb = seqs[a], seqs[b]
        L = int(rng.integers(1, min(len(sa), len(sb)) + 1))
        ia = int(rng.choice([0, len(sa) - L, rng.integers(0, len(sa) - L + 1)]))
        ib = int(rng.choice([0, len(sb) - L, rng.integers(0, len(sb) - L + 1)]))
        flips = (np.uint32(1) << rng.integers(0, 32, L).astype(np.uint32)) * (rng.random(L) < rng.uniform(0, 0.9))
        sb[ib:ib + L] = sa[ia:ia + L] ^ flips.astype(np.uint32)
    if rng.random() < 0.3:                                          # silence: one hash repeated
        k = int(rng.integers(0, nseq))
        L = int(rng.integers(1, len(seqs[k]) + 1))
        i0 = int(rng.integers(0, len(seqs[k]) - L + 1))
        seqs[k][i0:i0 + L] = seqs[k][i0]
        if rng.random() < 0.5:
            k2 = int(rng.integers(0, nseq))
            L2 = int(rng.integers(1, len(seqs[k2]) + 1))
            seqs[k2][:L2] = seqs[k][i0]
    thr = int(rng.integers(0, 17))
    big = max(len(s) for s in seqs) > 1500
    problems = []
    for _ in range(int(rng.integers(1, 4))):
        a, b = (int(x) for x in rng.integers(0, nseq, 2))
        min_len = int(rng.choice([1, 2, 20, 21, 22, 23, 24, 40, 82, 150])) if not big else int(rng.choice([21, 23, 30, 82, 150]))
        problems.append((a, b, min_len))
    r = capi.hamming_runs(seqs, problems, thr)
    got = {}
    for x in r:
        got.setdefault(int(x["problem"]), []).append((int(x["src_end"]), int(x["dst_end"]), int(x["len"]), int(x["src_match_hash"]), int(x["dst_match_hash"])))
    for pi, (a, b, min_len) in enumerate(problems):
        want = oracle_runs(seqs[a], seqs[b], thr, min_len)
        runs_total += len(want)
        if sorted(got.get(pi, [])) != want:
            bad += 1
            print(f"MISMATCH case {case} problem {pi}: n {len(seqs[a])} m {len(seqs[b])} thr {thr} min_len {min_len}: got {len(got.get(pi, []))} runs, want {len(want)}")
    if case % 20 == 0:
        print(f"case {case}: {runs_total} runs compared, {bad} bad problems", flush=True)
print(f"{cases} cases, {runs_total} runs compared, {bad} mismatching problems")
sys.exit(1 if bad else 0)
