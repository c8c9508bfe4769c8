"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI, against the
CPU oracle on the same seeded inputs.  Integer stages must be bit-exact; the f64 fingerprint stages must
yield bit-identical u32 hashes (north_star) with intermediate features within 1e-11 relative."""
import os

import numpy as np
import pytest

from needle_amd import capi, synth
from oracle import oracle as O

pytestmark = pytest.mark.gpu
NS = O.NS


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert capi.device_count() > 0, "GPU tests need a HIP device (the product has no CPU fallback)"


@pytest.fixture(scope="module")
def lib3():
    """BASELINE.json configs[0]: 3 synthetic 90 s episodes with a shared 20 s intro."""
    return synth.make_library(3, 90.0, 20.0)


def _rand_hashes(rng, n):
    return rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)


# ---- fingerprint (stft_chroma + fir_norm + classify) ---------------------------------------------------------
def test_fingerprint_hashes_bit_exact_vs_oracle(lib3):
    pcms = [e.pcm[: len(e.pcm) // 2] for e in lib3]
    got = capi.fingerprint(pcms, channels=1, step=1)
    for g, p in zip(got, pcms):
        want = O.fingerprint(p)
        assert len(g) == len(want) == 342
        assert g.tolist() == want.tolist()


def test_fingerprint_intermediates_close_and_margin(lib3):
    pcm = lib3[0].pcm[: 30 * 11025]
    chroma, feats = capi.fingerprint_debug(pcm)
    _, o_chroma, o_feats, margin = O.fingerprint(pcm, debug=True)
    assert chroma.shape == o_chroma.shape and feats.shape == o_feats.shape
    rel = np.max(np.abs(chroma - o_chroma) / np.maximum(np.abs(o_chroma), 1e-300))
    assert rel < 1e-11, rel
    assert np.max(np.abs(feats - o_feats)) < 1e-11
    # every quantiser decision of the oracle sits farther from its threshold than the stages differ
    assert margin > 1e-9


def test_fingerprint_step_and_stereo(lib3):
    pcm = lib3[1].pcm[: 40 * 11025]
    full = O.fingerprint(pcm)
    kept = capi.fingerprint([pcm], step=2)[0]
    assert kept.tolist() == full[::2].tolist()
    kept3 = capi.fingerprint([pcm], step=3)[0]
    assert kept3.tolist() == full[::3].tolist()
    stereo = capi.fingerprint([np.repeat(pcm, 2)], channels=2, step=1)[0]   # L = R (analyzer.rs:183-185,218)
    assert stereo.tolist() == full.tolist()
    lr = np.zeros(2 * len(pcm), dtype=np.int16)
    lr[0::2] = pcm
    lr[1::2] = np.roll(pcm, 3)
    assert capi.fingerprint([lr], channels=2)[0].tolist() == O.fingerprint(lr, channels=2).tolist()


def test_fingerprint_edge_lengths():
    rng = np.random.default_rng(5)
    lens = [0, 1, 4095, 4096, 4096 + 18 * 1365, 4096 + 19 * 1365, 4096 + 19 * 1365 + 1364, 4096 + 40 * 1365 + 7]
    pcms = [(rng.standard_normal(n) * 3000).astype(np.int16) for n in lens]
    got = capi.fingerprint(pcms, step=1)
    for g, p in zip(got, pcms):
        want = O.fingerprint(p)
        assert len(g) == len(want) == O.num_items(len(p))
        assert g.tolist() == want.tolist()
    # digital silence and full-scale square wave
    silence = np.zeros(4096 + 30 * 1365, np.int16)
    assert capi.fingerprint([silence])[0].tolist() == O.fingerprint(silence).tolist()
    loud = np.where((np.arange(4096 + 30 * 1365) // 25) % 2 == 0, 32767, -32768).astype(np.int16)
    assert capi.fingerprint([loud])[0].tolist() == O.fingerprint(loud).tolist()


def test_fingerprint_ragged_batch_matches_single_calls(lib3):
    pcms = [lib3[0].pcm[: 7 * 11025], lib3[1].pcm[: 3 * 11025 + 1], np.zeros(10, np.int16), lib3[2].pcm[: 20 * 11025]]
    batch = capi.fingerprint(pcms, step=2)
    for b, p in zip(batch, pcms):
        assert b.tolist() == O.fingerprint(p)[::2].tolist()


# ---- search (hamming_runs) ------------------------------------------------------------------------------------
def _oracle_runs(src, dst, thr, min_len):
    cmp = O.Comparator(hash_match_threshold=thr, min_opening_duration=0)
    ents = O.longest_common_hash_match(cmp, [(int(h), i) for i, h in enumerate(src)],
                                       [(int(h), i) for i, h in enumerate(dst)], 0, 0)
    return sorted((e["src_end_idx"], e["dst_end_idx"], e["score"], e["src_match_hash"], e["dst_match_hash"])
                  for e in ents if e["score"] >= min_len)


def _gpu_runs(seqs, problems, thr):
    r = capi.hamming_runs(seqs, problems, thr)
    out = {}
    for x in r:
        out.setdefault(int(x["problem"]), []).append((int(x["src_end"]), int(x["dst_end"]), int(x["len"]),
                                                      int(x["src_match_hash"]), int(x["dst_match_hash"])))
    return {k: sorted(v) for k, v in out.items()}


@pytest.mark.parametrize("n,m", [(171, 171), (2897, 2897), (5441, 2715), (2, 2), (2, 300), (300, 2), (65, 257)])
def test_hamming_runs_equal_reference_dp(n, m):
    """Sizes from SURVEY.md §8 (90 s / 24 min / 45 min windows) plus degenerate shapes; random hashes with
    planted near-duplicate runs, min_len = 1 so EVERY maximal run must be reported (runs at table edges,
    single cells, rows/cols 0 excluded, comparator.rs:179-200)."""
    rng = np.random.default_rng(n * 131 + m)
    src, dst = _rand_hashes(rng, n), _rand_hashes(rng, m)
    if n > 150 and m > 150:
        for (a, b, L) in [(10, 40, 90), (n - 60, m - 60, 60), (0, 0, 30), (100, 5, 45)]:
            L = min(L, n - a, m - b)
            noise = (np.uint32(1) << rng.integers(0, 32, L).astype(np.uint32)) * (rng.random(L) < 0.6)
            dst[b:b + L] = src[a:a + L] ^ noise
    thr = 10 if n > 2000 else 12
    min_len = 3 if n * m > 4_000_000 else 1     # keep the big cases' run lists (2.5 % random matches) short
    got = _gpu_runs([src, dst], [(0, 1, min_len)], thr).get(0, [])
    assert got == _oracle_runs(src, dst, thr, min_len)


@pytest.fixture(params=["sampled", "sampled-rows", "sampled-sparse", "sampled-mfma", "band", "generic"])
def search_mode(request):
    """The launcher picks the aligned-window kernel for min_len >= 23, the band kernel for >= 21 and the
    one-lane-per-diagonal kernel otherwise; the environment switches force the slower ones so every kernel is
    checked on the same inputs -- and the aligned-window kernel with the survivors of a window's head rows always
    walked row by row (NEEDLE_HIP_SPARSE_MAX=0) or always finished one diagonal at a time (1000)."""
    keys = ("NEEDLE_HIP_BAND_SEARCH", "NEEDLE_HIP_GENERIC_SEARCH", "NEEDLE_HIP_SPARSE_MAX", "NEEDLE_HIP_SCAN_MFMA")
    old = {k: os.environ.pop(k, None) for k in keys}
    if request.param == "sampled-mfma":          # the aligned-window kernel's first stage on the matrix pipe (opt-in)
        os.environ["NEEDLE_HIP_SCAN_MFMA"] = "1"
    if request.param == "band":
        os.environ["NEEDLE_HIP_BAND_SEARCH"] = "1"
    elif request.param == "generic":
        os.environ["NEEDLE_HIP_GENERIC_SEARCH"] = "1"
    elif request.param == "sampled-rows":
        os.environ["NEEDLE_HIP_SPARSE_MAX"] = "0"
    elif request.param == "sampled-sparse":
        os.environ["NEEDLE_HIP_SPARSE_MAX"] = "1000"
    yield request.param
    for k in keys:
        os.environ.pop(k, None)
        if old[k] is not None:
            os.environ[k] = old[k]


@pytest.mark.parametrize("n,m,min_len", [(2897, 2897, 82), (5441, 2715, 41), (2715, 5441, 23), (500, 449, 23),
                                         (449, 460, 25), (30, 3000, 23), (3000, 30, 24), (24, 24, 23), (2, 2, 30),
                                         (1443, 1443, 82), (700, 650, 200)])
def test_fast_kernels_equal_reference_dp(n, m, min_len, search_mode):
    """Long minimum run lengths select the pruned kernels.  Planted runs sit on every kind of edge: starting
    at row/col 1, ending at n-1 / m-1, crossing checkpoint rows, exactly min_len and min_len-1 long, and
    preceded by matches in row/col 0 (which the reference never counts, comparator.rs:179-180)."""
    rng = np.random.default_rng(n * 7 + m * 3 + min_len)
    src, dst = _rand_hashes(rng, n), _rand_hashes(rng, m)

    def plant(a, b, L):
        L = min(L, n - a, m - b)
        if L > 0:
            noise = (np.uint32(1) << rng.integers(0, 32, L).astype(np.uint32)) * (rng.random(L) < 0.5)
            dst[b:b + L] = src[a:a + L] ^ noise
    for (a, b, L) in [(0, 0, min_len + 5), (n - min_len - 3, m - min_len - 3, min_len + 3), (0, m // 2, min_len),
                      (n // 2, 0, min_len + 1), (n // 3, m // 3, min_len - 1), (n // 4, m // 2 + 7, 3 * min_len),
                      (1, m - min_len - 1, min_len + 1), (n - min_len - 1, 1, min_len + 1)]:
        if a >= 0 and b >= 0:
            plant(a, b, L)
    thr = 10
    got = _gpu_runs([src, dst], [(0, 1, min_len)], thr).get(0, [])
    want = _oracle_runs(src, dst, thr, min_len)
    assert got == want
    if n > 400 and m > 400:
        assert len(want) >= 2
    if min(n, m) >= 2:                                           # which kernel really ran (needle_hip_scan_last_launch)
        form = capi.scan_last_launch()[0]
        assert form == {"sampled-mfma": 4, "band": 2, "generic": 1}.get(search_mode, 3), (form, search_mode, n, m, min_len)


@pytest.mark.parametrize("min_len,seed", [(82, 1), (82, 2), (41, 3), (163, 4), (23, 5), (300, 7), (700, 6)])
def test_ragged_runs_over_chains_of_windows(min_len, seed, search_mode):
    """One long common stretch broken by single mismatching rows, the way a shared intro looks in real audio -- what the
    matrix-pipe form resolves by CHAINS of whole windows (scan_mfma_kernel.h resolve()): mismatches in the gap between two
    aligned windows (both sides are runs, one walk reports both), inside an aligned window, two in one gap, two next to each
    other, stretches of exactly min_len and min_len - 1 rows, a window isolated by mismatches two rows off on both sides, a
    stretch that reaches the table's last row, slowly changing hashes (neighbouring diagonals match as well), and a run
    that goes on for more than 128 rows behind its chain's last window (min_len 163), windows further apart than a walk's
    block of 512 rows (min_len 700).  Every kernel form against the DP."""
    rng = np.random.default_rng(1000 + seed)
    P = min_len - 8 + 1
    a, b, L = 140, 87, max(2400, 14 * P + 300)                    # src[a + i] ~ dst[b + i]
    n, m = a + L + 560, b + L + 463                               # (3100 x 2950 for the usual window spacings)
    src, dst = _rand_hashes(rng, n), _rand_hashes(rng, m)
    src[a:a + L:2] = src[a + 1:a + L + 1:2]                       # pairs of equal hashes: diagonals d +- 1 match every other row
    noise = (np.uint32(1) << rng.integers(0, 32, L).astype(np.uint32)) * (rng.random(L) < 0.4)
    dst[b:b + L] = src[a:a + L] ^ noise
    tail = min(n, m) - 300                                        # a second stretch up to the last row of both
    src[n - 300:] = _rand_hashes(rng, 300)
    dst[m - 300:] = src[n - 300:]
    del tail

    def first_window_at_or_after(row):                            # aligned windows start at rows 1 + k P
        return 1 + -(-(row - 1) // P) * P
    w = first_window_at_or_after(a + 3 * P)
    breaks = [w + 8 + P // 2,                                     # in a gap
              w + P + 3,                                          # inside the next aligned window
              w + 2 * P + 10, w + 2 * P + P - 5,                  # two in one gap
              w + 4 * P + 20, w + 4 * P + 21,                     # neighbours
              w + 6 * P - 2, w + 6 * P + 9]                       # two rows off a window on both sides: a 10-row stretch
    x = w + 8 * P + 11
    breaks += [x, x + min_len + 1, x + 2 * min_len + 1]           # stretches of exactly min_len and min_len - 1 rows
    for row in breaks:
        if a <= row < a + L:
            dst[b + (row - a)] ^= np.uint32(0xFFFFF000)           # 20 bits: over any threshold used here
    thr = 10
    got = _gpu_runs([src, dst], [(0, 1, min_len)], thr).get(0, [])
    if (min_len, seed) not in _RAGGED_WANT:                       # (the DP over 10 000 x 10 000 cells takes the oracle a minute: once)
        _RAGGED_WANT[(min_len, seed)] = _oracle_runs(src, dst, thr, min_len)
    want = _RAGGED_WANT[(min_len, seed)]
    assert got == want
    assert len(want) >= 4


_RAGGED_WANT = {}


def test_fast_kernels_all_cells_match_and_many_problems(search_mode):
    rng = np.random.default_rng(77)
    # threshold >= 32: every cell with i, j >= 1 matches; each diagonal is one run as long as the diagonal
    src, dst = _rand_hashes(rng, 300), _rand_hashes(rng, 260)
    for thr in (32, 65535):
        assert _gpu_runs([src, dst], [(0, 1, 30)], thr).get(0, []) == _oracle_runs(src, dst, thr, 30)
    seqs = [_rand_hashes(rng, int(k)) for k in rng.integers(40, 900, 8)]
    for k in range(0, 8, 2):
        L = min(len(seqs[k]), len(seqs[k + 1])) - 3
        seqs[k + 1][2:2 + L] = seqs[k][1:1 + L]
    problems = [(a, b, 23 + (a * 5 + b) % 40) for a in range(8) for b in range(8) if a != b]
    got = _gpu_runs(seqs, problems, 9)
    for p, (a, b, ml) in enumerate(problems):
        assert got.get(p, []) == _oracle_runs(seqs[a], seqs[b], 9, ml), (p, a, b)


@pytest.mark.parametrize("waves,splits", [(4, 1), (8, 3), (16, 2), (12, 1), (8, 1)])
def test_matrix_pipe_form_in_its_shapes_with_crowds_of_chains(waves, splits, monkeypatch):
    """The matrix-pipe form's workgroup shapes (NEEDLE_HIP_MFMA_WAVES) and workgroups per group (NEEDLE_HIP_MFMA_SPLITS) on content
    that fills its queues: a ragged shared intro (chains of whole windows) and stretches of ONE repeated hash in three sequences --
    every diagonal of a block a chain: the workgroup's chain ring wraps, a wave in a crowd hands chains over, the waves that are
    through stay and take them, run buffers overflow (scan_mfma_kernel.h, round 6).  Complete run lists against the DP."""
    monkeypatch.setenv("NEEDLE_HIP_SCAN_MFMA", "1")
    monkeypatch.setenv("NEEDLE_HIP_MFMA_WAVES", str(waves))
    monkeypatch.setenv("NEEDLE_HIP_MFMA_SPLITS", str(splits))
    rng = np.random.default_rng(4242)
    lens = [1400, 1250, 1500, 1100, 1333]
    seqs = [_rand_hashes(rng, n) for n in lens]
    intro = _rand_hashes(rng, 260)
    for k, at in zip(range(4), (50, 400, 900, 17)):                # the intro, a bit of noise in every copy, a break in two of them
        noise = (np.uint32(1) << rng.integers(0, 32, 260).astype(np.uint32)) * (rng.random(260) < 0.4)
        seqs[k][at:at + 260] = intro ^ noise
    seqs[1][400 + 131] ^= np.uint32(0xFFFFF000)
    seqs[3][17 + 77] ^= np.uint32(0xFFFFF000)
    for k, at, length in ((1, 800, 150), (2, 200, 120), (4, 600, 170)):   # digital silence: one constant hash
        seqs[k][at:at + length] = np.uint32(0x1F07C1F0)
    seqs[4][900:1010] = np.uint32(0x1F07C1F1)                     # ... and a second constant one bit away (a sustained chord)
    problems = [(a, b, 40) for a in range(5) for b in range(a + 1, 5)]
    got = _gpu_runs(seqs, problems, 10)
    assert capi.scan_last_launch()[0] == 4
    most = 0
    for p, (a, b, ml) in enumerate(problems):
        want = _oracle_runs(seqs[a], seqs[b], 10, ml)
        assert got.get(p, []) == want, (p, a, b, waves, splits)
        most = max(most, len(want))
    assert most > 200, most                                       # a pair of silent stretches: hundreds of runs


def test_hamming_runs_thresholds_and_min_len():
    rng = np.random.default_rng(9)
    src, dst = _rand_hashes(rng, 120), _rand_hashes(rng, 140)
    dst[20:70] = src[30:80]
    for thr in [0, 1, 31, 32, 40, 65535]:      # >= 32 matches every cell with i,j >= 1
        assert _gpu_runs([src, dst], [(0, 1, 1)], thr).get(0, []) == _oracle_runs(src, dst, thr, 1)
    for min_len in [2, 49, 50, 51, 1000]:
        assert _gpu_runs([src, dst], [(0, 1, min_len)], 10).get(0, []) == _oracle_runs(src, dst, 10, min_len)


def test_hamming_runs_many_problems_in_one_launch():
    rng = np.random.default_rng(10)
    seqs = [_rand_hashes(rng, int(k)) for k in rng.integers(2, 400, 9)]
    seqs.append(np.zeros(0, np.uint32))
    seqs.append(_rand_hashes(rng, 1))
    for k in range(0, 8, 2):
        L = min(len(seqs[k]), len(seqs[k + 1])) // 2
        seqs[k + 1][:L] = seqs[k][-L:]
    problems = [(a, b, 1 + (a + b) % 3) for a in range(len(seqs)) for b in range(len(seqs)) if a != b]
    got = _gpu_runs(seqs, problems, 11)
    for p, (a, b, ml) in enumerate(problems):
        assert got.get(p, []) == _oracle_runs(seqs[a], seqs[b], 11, ml), (p, a, b)


def test_pairs_too_long_for_lds_are_scanned_from_hbm(search_mode):
    """A window longer than the scan kernels can stage in LDS (~39 000 hashes) must not fail the launch: such pairs go
    to the unstaged kernel, the others of the same launch stay on the fast one (search.hip build_plan).  The limit is
    lowered through the environment so that ordinary sizes exercise the split; results must equal the reference's
    table for every pair, whichever kernel scanned it."""
    rng = np.random.default_rng(4242)
    seqs = [_rand_hashes(rng, k) for k in (300, 2600, 700, 2900, 450)]
    for a, b in [(0, 1), (2, 3), (1, 3), (4, 0)]:
        L = min(len(seqs[a]), len(seqs[b])) // 2
        seqs[b][5:5 + L] = seqs[a][3:3 + L]
    problems = [(a, b, 23 + (3 * a + b) % 17) for a in range(5) for b in range(5) if a != b]
    old = os.environ.get("NEEDLE_HIP_SEARCH_LDS_LIMIT")
    os.environ["NEEDLE_HIP_SEARCH_LDS_LIMIT"] = str(4 * (2000 + 2 * 448))   # destinations > 2000 hashes do not "fit"
    try:
        got = _gpu_runs(seqs, problems, 10)
    finally:
        os.environ.pop("NEEDLE_HIP_SEARCH_LDS_LIMIT", None)
        if old is not None:
            os.environ["NEEDLE_HIP_SEARCH_LDS_LIMIT"] = old
    for p, (a, b, ml) in enumerate(problems):
        assert got.get(p, []) == _oracle_runs(seqs[a], seqs[b], 10, ml), (p, a, b)


def test_a_window_of_41500_hashes_is_searched():
    """The real thing once: 41 500 hashes (2.9 h of audio at step 1) against 1 500, beyond the 160 KiB of a CU (the
    oracle's table for this pair takes 0.5 GB)."""
    rng = np.random.default_rng(45)
    long_seq, short_seq = _rand_hashes(rng, 41500), _rand_hashes(rng, 1500)
    long_seq[40000:40400] = short_seq[100:500]
    long_seq[10:300] = short_seq[1200:1490]
    got = _gpu_runs([short_seq, long_seq], [(0, 1, 100), (1, 0, 100)], 10)
    want = _oracle_runs(short_seq, long_seq, 10, 100)
    assert got.get(0, []) == want and len(want) == 2
    assert got.get(1, []) == _oracle_runs(long_seq, short_seq, 10, 100)


# ---- Analyzer / Comparator through the C ABI ---------------------------------------------------------------------
def _same_results(got, want):
    g = [None if r is None else (r.opening, r.ending) for r in got]
    w = [None if r is None else (r.opening, r.ending) for r in want]
    assert g == w


def test_config1_analyze_search_matches_oracle(lib3, tmp_path):
    """configs[0]: 3 x 90 s, shared 20 s intro; default min_opening_duration (20 s) finds nothing
    (SURVEY.md §7.6), 10 s finds the intro in every episode; both must equal the reference path."""
    paths = [str(tmp_path / f"ep{k}.wav") for k in range(3)]
    fhs = capi.Analyzer.from_files(paths).run_pcm([e.pcm for e in lib3])
    hd = O.duration_from_secs_f32(0.3)
    ref = O.analyze_batch([e.pcm[: len(e.pcm) // 2] for e in lib3], 1, hd)
    for got, want in zip(fhs, ref):
        h, ts = got.opening_data()
        assert h.tolist() == [x for x, _ in want.opening]
        assert ts.tolist() == [t for _, t in want.opening]
        assert got.hash_duration() == 300_000_012 and len(got.ending_data()[0]) == 0
    for min_s in (20, 10, 0):
        res = capi.Comparator.from_files(paths).with_min_opening_duration(min_s).run_with_frame_hashes(fhs)
        want = O.run_with_frame_hashes(O.Comparator(min_opening_duration=min_s * NS), ref)
        _same_results(res, want)
    res10 = capi.Comparator.from_files(paths).with_min_opening_duration(10).run_with_frame_hashes(fhs)
    for r, e in zip(res10, lib3):
        assert r is not None and r.opening is not None
        # the reference stamps a hash with the END of its 2.6 s analysis span (delay 2600 ms, analyzer.rs:288,309)
        # and runs its clock at 123/123.81: detected times trail the planted intro by at most that span
        start, end = r.opening[0] / 1e9, r.opening[1] / 1e9
        assert e.intro_off / 11025 <= start <= e.intro_off / 11025 + 2.7
        assert abs(end - (e.intro_off + e.intro_len) / 11025) < 2.7


def test_endings_and_parameters_match_oracle(tmp_path):
    eps = synth.make_library(4, 120.0, 25.0, 22.0)
    paths = [str(tmp_path / f"e{k}.wav") for k in range(4)]
    an = capi.Analyzer.from_files(paths).with_include_endings(True).with_opening_search_percentage(0.4) \
        .with_ending_search_percentage(0.35)
    fhs = an.run_pcm([e.pcm for e in eps])
    hd = O.duration_from_secs_f32(0.3)
    ref = []
    for e, f in zip(eps, fhs):
        total = len(e.pcm)
        dur = O.duration_from_secs_f64(total * (1.0 / 11025.0))
        n_open = O.duration_mul_f32(dur, 0.4) * 11025 // NS
        seek = O.duration_mul_f32(dur, float(np.float32(1.0) - np.float32(0.35)))
        first = seek * 11025 // NS
        o = O.step_and_timestamp(O.fingerprint(e.pcm[:n_open]), hd)
        en = O.step_and_timestamp(O.fingerprint(e.pcm[first:]), hd, seek_to_ns=seek)
        ref.append(O.FrameHashes(o, en, hd))
        assert f.opening_data()[0].tolist() == [h for h, _ in o] and f.opening_data()[1].tolist() == [t for _, t in o]
        assert f.ending_data()[0].tolist() == [h for h, _ in en] and f.ending_data()[1].tolist() == [t for _, t in en]
    for kw in [dict(include_endings=True, min_opening_duration=12, min_ending_duration=12),
               dict(include_endings=True, min_opening_duration=12, min_ending_duration=12, time_padding=1.5),
               dict(include_endings=False, min_opening_duration=15, hash_match_threshold=6),
               dict(include_endings=True, min_opening_duration=5, min_ending_duration=30, hash_match_threshold=14)]:
        c = capi.Comparator(paths, **kw)
        got = c.run_with_frame_hashes(fhs)
        ocmp = O.Comparator(include_endings=kw.get("include_endings", False),
                            hash_match_threshold=kw.get("hash_match_threshold", 10),
                            min_opening_duration=kw.get("min_opening_duration", 20) * NS,
                            min_ending_duration=kw.get("min_ending_duration", 20) * NS,
                            time_padding=O.duration_from_secs_f32(kw.get("time_padding", 0.0)))
        _same_results(got, O.run_with_frame_hashes(ocmp, ref))
    # include_endings without ending data is the reference's FrameHashDataNoEnding error
    fh_no_end = capi.Analyzer.from_files(paths).run_pcm([e.pcm for e in eps])
    with pytest.raises(capi.NeedleError):
        capi.Comparator(paths, include_endings=True).run_with_frame_hashes(fh_no_end)


def test_file_based_c_abi_roundtrip(lib3, tmp_path, capfd):
    """The unmodified needle-capi call sequence on WAV files: analyze with persist (.needle.dat), search from
    disk with display + skip files, then the cached / skip-file paths (analyzer.rs:338-348, comparator.rs:600-605)."""
    paths = []
    for k, e in enumerate(lib3):
        p = str(tmp_path / f"show-ep{k}.wav")
        synth.write_wav(p, e.pcm, channels=2)            # stereo L = R like the reference's resampler output
        paths.append(p)
    an = capi.Analyzer.from_files(paths)
    fhs = an.run(0.3, persist=True)
    hd = O.duration_from_secs_f32(0.3)
    ref = O.analyze_batch([e.pcm[: len(e.pcm) // 2] for e in lib3], 1, hd)
    for p, got, want in zip(paths, fhs, ref):
        assert got.opening_data()[0].tolist() == [h for h, _ in want.opening]
        assert got.md5() == O.header_md5(p)
        rc, disk = O.frame_hashes_read(p[:-4] + ".needle.dat")     # our file, read by the oracle's bincode reader
        assert rc == 0 and disk.opening == want.opening and disk.md5 == got.md5() and disk.hash_duration == hd
    capfd.readouterr()
    an2 = capi.Analyzer.from_files(paths)
    an2.run(0.3, persist=False)                           # cached: "Skipping analysis for ..." (analyzer.rs:344)
    out = capfd.readouterr().out
    assert out.count("Skipping analysis for") == 3
    c = capi.Comparator(paths, min_opening_duration=10)
    c.run(analyze=False, display=True, write_skip_files=True)
    out = capfd.readouterr().out
    want = O.run_with_frame_hashes(O.Comparator(min_opening_duration=10 * NS), ref)
    for p, w in zip(paths, want):
        assert f"\n{p}\n" in out
        line = f'* Opening - "{O.format_time(w.opening[0])}"-"{O.format_time(w.opening[1])}"'
        assert line in out
        skip = open(p[:-4] + ".needle.skip.json").read()
        assert skip == O.skip_file_json(w, O.header_md5(p))
    c.run(analyze=False, display=True, use_skip_files=True)
    assert capfd.readouterr().out.count("Skipping due to existing skip file...") == 3
    # analyze-in-place variant of search (comparator.rs:650-654 -> data.rs:134-136)
    for p in paths:
        os.remove(p[:-4] + ".needle.dat")
    c.run(analyze=True, display=True)
    out = capfd.readouterr().out
    assert all(f'* Opening - "{O.format_time(w.opening[0])}"' in out for w in want)
    with pytest.raises(capi.NeedleError) as ei:
        c.run(analyze=False)
    assert ei.value.name == "FrameHashDataNotFound"


# ---- library-scale parity at BASELINE.json's full size ----------------------------------------------------------
@pytest.fixture(scope="module")
def lib28():
    """configs[1]: 28 episodes x 24 min, 90 s shared intro."""
    return synth.make_library(28, 24 * 60.0, 90.0)


def test_config2_full_size_parity(lib28):
    n = len(lib28)
    threads = min(os.cpu_count() or 1, 16)
    hd = O.duration_from_secs_f32(0.3)
    windows = [e.pcm[: len(e.pcm) // 2] for e in lib28]
    ref = O.analyze_batch(windows, 1, hd, threads=threads)
    lib = capi.Library(n)
    lib.set_pcm([e.pcm for e in lib28], [len(e.pcm) for e in lib28])
    lib.analyze()
    for v in range(n):
        h, ts = lib.frame_hashes(v).opening_data()
        assert len(h) == 2897
        assert h.tolist() == [x for x, _ in ref[v].opening], f"episode {v}"
        assert ts.tolist() == [t for _, t in ref[v].opening]
    # the first pass audited on the device: the f64 kernel over the same resident PCM, every kept item compared
    # (needle_hip_library_audit).  Accepted items must all be the f64 item; what the radius K S = 64 S is a guard against
    # -- |log v32 - log v64| / S of an accepted item -- must stay an order of magnitude inside it.
    audit = lib.audit()
    print("audit 28 x 24 min:", audit)
    assert audit["items"] == n * 2897 and audit["mismatches"] == 0 and audit["accepted_mismatches"] == 0
    assert audit["accepted"] > 0.99 * audit["items"] and audit["max_error_over_s"] <= 8.0
    cmp = capi.Comparator([f"ep{k}.wav" for k in range(n)])
    cmp.handle()
    cap = 1 << 16
    d_runs, d_count = capi.DeviceBuffer(cap * capi.RUN_DTYPE.itemsize), capi.DeviceBuffer(4)
    lib.search(cmp, 0, lib.num_pairs(), d_runs.ptr, cap, d_count.ptr)
    count = int(d_count.to_host(np.uint32, 1)[0])
    assert 0 < count <= cap
    runs = d_runs.to_host(capi.RUN_DTYPE, count)
    got = lib.finalize(cmp, runs)
    want = O.run_with_frame_hashes(O.Comparator(), ref, threads=threads)
    _same_results(got, want)
    # Ground truth.  GPU vs oracle is 0 ns (above); against the PLANTED edges the reference's own semantics are
    # offset: a hash is stamped 2.6 s + 0.123 s * item (its clock runs 0.65 % slow against the real 123.81 ms hop),
    # covers 2.72 s of audio and matches with <= 10 differing bits.  Measured on this library (28 episodes, intro
    # offsets 17 s .. 6 min): start - planted start in [-1.60, +1.08] s (mean -0.06), end - planted end in
    # [-0.89, +1.22] s (mean +0.19).  The bound below is that distribution with a little headroom, not a tolerance
    # of the implementation.
    d_start, d_end = [], []
    for r, e in zip(got, lib28):
        assert r is not None and r.opening is not None
        d_start.append(r.opening[0] / 1e9 - e.intro_off / 11025)
        d_end.append(r.opening[1] / 1e9 - (e.intro_off + e.intro_len) / 11025)
    print(f"opening start - planted: min {min(d_start):+.3f} mean {sum(d_start) / n:+.3f} max {max(d_start):+.3f} s; "
          f"end - planted: min {min(d_end):+.3f} mean {sum(d_end) / n:+.3f} max {max(d_end):+.3f} s")
    assert -1.75 < min(d_start) and max(d_start) < 1.25
    assert -1.0 < min(d_end) and max(d_end) < 1.4
    # sharded search (two "ranks" splitting the pair list) gives the same run set
    half = lib.num_pairs() // 2
    parts = []
    for first, cnt in [(0, half), (half, lib.num_pairs() - half)]:
        lib.search(cmp, first, cnt, d_runs.ptr, cap, d_count.ptr)
        c = int(d_count.to_host(np.uint32, 1)[0])
        parts.append(d_runs.to_host(capi.RUN_DTYPE, c))
    merged = np.concatenate(parts)
    assert sorted(map(tuple, merged.tolist())) == sorted(map(tuple, runs.tolist()))
    _same_results(lib.finalize(cmp, merged), want)


def test_gpu_reproduces_committed_golden_vectors(lib3, tmp_path):
    """tests/golden/config1.json (oracle outputs for the seeded configs[0] library) through the C ABI."""
    import json
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "config1.json")))
    paths = [str(tmp_path / f"g{k}.wav") for k in range(3)]
    fhs = capi.Analyzer.from_files(paths).run_pcm([e.pcm for e in lib3])
    for f, want in zip(fhs, g["opening"]):
        h, ts = f.opening_data()
        assert [[int(a), int(b)] for a, b in zip(h, ts)] == want
    assert capi.fingerprint([lib3[0].pcm[: len(lib3[0].pcm) // 2]])[0].tolist() == g["raw_items_first_episode"]
    for min_s, want in g["results"].items():
        res = capi.Comparator(paths, min_opening_duration=int(min_s)).run_with_frame_hashes(fhs)
        assert [None if r is None else list(r.opening) if r.opening else [] for r in res] == want


def test_config3_search_only_from_needle_dat_files(tmp_path, capfd):
    """BASELINE.json configs[2] at reduced count: precomputed .needle.dat files (written by the ORACLE's bincode
    writer, 24-min-sized: 2897 opening + 1443 ending hashes), search-only through needle_audio_comparator_run
    with endings, display and skip files; every printed line and skip file must equal the reference path."""
    n = 24
    rng = np.random.default_rng(2024)
    hd = O.duration_from_secs_f32(0.3)
    intro = _rand_hashes(rng, 360)
    outro = _rand_hashes(rng, 250)
    fhs, paths = [], []
    for v in range(n):
        op = _rand_hashes(rng, 2897)
        en = _rand_hashes(rng, 1443)
        if v % 6 != 5:                                    # every sixth episode has no intro / outro
            a = 40 + 37 * (v % 7)
            op[a:a + 360] = intro ^ ((np.uint32(1) << rng.integers(0, 32, 360).astype(np.uint32)) * (rng.random(360) < 0.7))
            b = 300 + 29 * (v % 5)
            en[b:b + 250] = outro ^ ((np.uint32(1) << rng.integers(0, 32, 250).astype(np.uint32)) * (rng.random(250) < 0.7))
        o = O.step_and_timestamp(np.repeat(op, 2)[: 2 * len(op)], hd)[: len(op)]
        o = [(int(h), t) for h, (_, t) in zip(op, o)]
        e = [(int(h), t + 1080 * NS) for h, (_, t) in zip(en, O.step_and_timestamp(np.repeat(en, 2), hd))]
        p = tmp_path / f"library-ep{v:02d}.wav"
        p.write_bytes(bytes(range(256)) * 32 + bytes([v]) * 64)          # >= 8 KiB so the header MD5 exists
        f = O.FrameHashes(o, e, hd, O.header_md5(str(p)))
        assert O.frame_hashes_write(str(p)[:-4] + ".needle.dat", f) == 0
        fhs.append(f)
        paths.append(str(p))
    cmp = capi.Comparator(paths, include_endings=True, min_opening_duration=30, min_ending_duration=25)
    cmp.run(analyze=False, display=True, write_skip_files=True)
    out = capfd.readouterr().out
    want = O.run_with_frame_hashes(O.Comparator(include_endings=True, min_opening_duration=30 * NS,
                                                min_ending_duration=25 * NS), fhs, threads=min(os.cpu_count() or 1, 16))
    blocks = out.split("\n\n")
    expected = []
    for p, w in zip(paths, want):
        expected.append(f"\n{p}\n")
        if w is None:
            expected.append("No opening or ending found.")
        else:
            o = f'* Opening - "{O.format_time(w.opening[0])}"-"{O.format_time(w.opening[1])}"' if w.opening else "* Opening - N/A"
            e = f'* Ending - "{O.format_time(w.ending[0])}"-"{O.format_time(w.ending[1])}"' if w.ending else "* Ending - N/A"
            expected.append(o + "\n" + e)
    assert out == "\n".join(expected) + "\n", blocks[:3]
    found = 0
    for p, w in zip(paths, want):
        skip = p[:-4] + ".needle.skip.json"
        if w is None or (w.opening is None and w.ending is None):
            assert not os.path.exists(skip)
        else:
            assert open(skip).read() == O.skip_file_json(w, O.header_md5(p))
            found += 1
    assert found >= 18
    # in-memory call returns the same per-video results
    got = cmp.run_with_frame_hashes([capi.FrameHashes.from_path(p[:-4] + ".needle.dat") for p in paths])
    _same_results(got, want)
    # The search-only call keeps the objects that hold the videos' hashes between calls and fills its hash arena in pinned
    # memory: the same call again (objects reused), from pageable memory, and -- the objects of the 24 now holding other
    # videos' hashes -- a library of fewer, shorter videos must print what the first call / the oracle print.
    cmp.run(analyze=False, display=True)
    assert capfd.readouterr().out == out
    os.environ["NEEDLE_HIP_PAGEABLE_ARENA"] = "1"
    try:
        cmp.run(analyze=False, display=True)
    finally:
        del os.environ["NEEDLE_HIP_PAGEABLE_ARENA"]
    assert capfd.readouterr().out == out
    small_fhs, small_paths = [], []
    for v in range(7):
        op = _rand_hashes(rng, 900 + 31 * v)
        if v != 3:
            op[50 + 11 * v:50 + 11 * v + 300] = intro[:300] ^ ((np.uint32(1) << rng.integers(0, 32, 300).astype(np.uint32)) * (rng.random(300) < 0.5))
        o = [(int(h), t) for h, (_, t) in zip(op, O.step_and_timestamp(np.repeat(op, 2), hd))]
        p = tmp_path / f"small-ep{v:02d}.wav"
        p.write_bytes(bytes(range(256)) * 32 + bytes([100 + v]) * 64)
        f = O.FrameHashes(o, [], hd, O.header_md5(str(p)))
        assert O.frame_hashes_write(str(p)[:-4] + ".needle.dat", f) == 0
        small_fhs.append(f)
        small_paths.append(str(p))
    small = capi.Comparator(small_paths, min_opening_duration=30)
    small.run(analyze=False, display=True)
    small_out = capfd.readouterr().out
    small_want = O.run_with_frame_hashes(O.Comparator(min_opening_duration=30 * NS), small_fhs, threads=4)
    expected = []
    for p, w in zip(small_paths, small_want):                       # (no endings asked for: comparator.rs:356-381 prints no ending line)
        expected.append(f"\n{p}\n")
        if w is None:
            expected.append("No opening found.")
        else:
            expected.append(f'* Opening - "{O.format_time(w.opening[0])}"-"{O.format_time(w.opening[1])}"' if w.opening else "* Opening - N/A")
    assert small_out == "\n".join(expected) + "\n"
    assert sum(w is not None and w.opening is not None for w in small_want) >= 5


def test_chromaprint_compat_streaming_equals_oracle(lib3):
    """libchromaprint's streaming calls as needle issues them (start(rate, 2); feed per resampled frame; finish;
    get_raw_fingerprint) give the oracle's raw fingerprint, whatever the feed chunking."""
    import ctypes as C
    from .test_capi_cpu import _chromaprint_lib
    L = _chromaprint_lib()
    pcm = np.ascontiguousarray(lib3[2].pcm[: 50 * 11025])
    want = O.fingerprint(pcm)
    stereo = np.ascontiguousarray(np.repeat(pcm, 2))
    for channels, data, chunk in [(1, pcm, 4096), (2, stereo, 2 * 1152), (2, stereo, 2 * 7), (1, pcm, len(pcm))]:
        ctx = L.chromaprint_new(1)
        assert L.chromaprint_start(ctx, L.chromaprint_get_sample_rate(ctx), channels) == 1
        step = chunk if chunk > 64 else 2 * 50001                              # tiny chunks only for the head
        off = 0
        first = True
        while off < len(data):
            size = min(chunk if first else step, len(data) - off)
            assert L.chromaprint_feed(ctx, data[off:].ctypes.data, size) == 1
            off += size
            first = False
        assert L.chromaprint_finish(ctx) == 1
        fp = C.POINTER(C.c_uint32)()
        n = C.c_int(0)
        assert L.chromaprint_get_raw_fingerprint(ctx, C.byref(fp), C.byref(n)) == 1
        assert [fp[i] for i in range(n.value)] == want.tolist()
        L.chromaprint_dealloc(fp)
        L.chromaprint_free(ctx)


def test_library_with_endings_matches_oracle():
    """The HBM-resident Library with both search windows (Analyzer.with_include_endings) and a comparator that
    asks for endings: hashes, timestamps (seek offset included) and results equal the reference path."""
    eps = synth.make_library(6, 150.0, 30.0, 28.0)
    hd = O.duration_from_secs_f32(0.3)
    lib = capi.Library(len(eps)).include_endings(0.25)
    assert lib.rows_per_video() == 2
    lib.set_pcm([e.pcm for e in eps], [len(e.pcm) for e in eps])
    lib.analyze()
    ref = []
    for v, e in enumerate(eps):
        dur = O.duration_from_secs_f64(len(e.pcm) * (1.0 / 11025.0))
        n_open = O.duration_mul_f32(dur, 0.5) * 11025 // NS
        seek = O.duration_mul_f32(dur, float(np.float32(1.0) - np.float32(0.25)))
        first = seek * 11025 // NS
        o = O.step_and_timestamp(O.fingerprint(e.pcm[:n_open]), hd)
        en = O.step_and_timestamp(O.fingerprint(e.pcm[first:]), hd, seek_to_ns=seek)
        ref.append(O.FrameHashes(o, en, hd))
        f = lib.frame_hashes(v)
        assert [list(x) for x in zip(*[a.tolist() for a in f.opening_data()])] == [list(x) for x in o]
        assert [list(x) for x in zip(*[a.tolist() for a in f.ending_data()])] == [list(x) for x in en]
    for kw in (dict(include_endings=True, min_opening_duration=20, min_ending_duration=15),
               dict(include_endings=False, min_opening_duration=25)):
        cmp = capi.Comparator([f"e{k}.wav" for k in range(len(eps))], **kw)
        cmp.handle()
        cap = 1 << 14
        d_runs, d_count = capi.DeviceBuffer(cap * capi.RUN_DTYPE.itemsize), capi.DeviceBuffer(4)
        lib.search(cmp, 0, lib.num_pairs(), d_runs.ptr, cap, d_count.ptr)
        runs = d_runs.to_host(capi.RUN_DTYPE, int(d_count.to_host(np.uint32, 1)[0]))
        got = lib.finalize(cmp, runs)
        want = O.run_with_frame_hashes(O.Comparator(include_endings=kw["include_endings"],
                                                    min_opening_duration=kw["min_opening_duration"] * NS,
                                                    min_ending_duration=kw.get("min_ending_duration", 20) * NS), ref)
        _same_results(got, want)
        if kw["include_endings"]:
            assert all(r is not None and r.opening is not None and r.ending is not None for r in got)
    # a comparator that wants endings on a library analysed without them is the reference's NoEnding error
    plain = capi.Library(2)
    plain.set_pcm([eps[0].pcm, eps[1].pcm], [len(eps[0].pcm), len(eps[1].pcm)])
    plain.analyze()
    c2 = capi.Comparator(["a.wav", "b.wav"], include_endings=True)
    c2.handle()
    with pytest.raises(capi.NeedleError):
        plain.search(c2, 0, 1, d_runs.ptr, cap, d_count.ptr)


def test_resampler_bit_exact_and_analyzer_accepts_decode_rates(tmp_path):
    """The device resampler equals its oracle bit for bit (integer / f32 arithmetic in a fixed order), for the
    usual decode rates, mono and stereo, ragged batches; and a 44.1 kHz stereo WAV analysed through the unchanged
    needle-capi call gives the hashes of the oracle chain resample -> fingerprint."""
    rng = np.random.default_rng(99)
    # 44.1k: contiguous LDS layout, one phase; 48k/32k/16k/8k: row layout; 22.05k and 12345: lanes with different tap
    # alignment (coefficients straight from global memory); 96k: longest filter; 11025: identity + down-mix
    # 192k / 176.4k: filters too long for the per-wave coefficient scratch (global-memory coefficient path, fewer rows)
    # round 2: 44.1k / 22.05k take the integer-decimation kernel (scalar coefficients), the row-layout rates the kernel with
    # four outputs per lane and DPP-broadcast coefficients, 88.2k / 12345 / 192k / 176.4k / 11025 the first kernel
    for rate, ch in [(44100, 2), (44100, 1), (48000, 2), (48000, 1), (22050, 1), (22050, 2), (32000, 2), (8000, 1),
                     (11025, 2), (96000, 2), (16000, 1), (12345, 1), (88200, 2), (192000, 2), (176400, 1)]:
        pcms = []
        for n in (0, 1, 777, rate * 3 + 17):
            t = np.arange(n) / rate
            x = 6000 * np.sin(2 * np.pi * (300 + 40 * len(pcms)) * t) + 2000 * rng.standard_normal(n)
            x = np.clip(x, -32768, 32767).astype(np.int16)
            if ch == 2:
                x = np.stack([x, np.roll(x, 5)], axis=1).reshape(-1)
            pcms.append(x)
        got = capi.resample(pcms, ch, rate)
        for g, p in zip(got, pcms):
            assert g.tolist() == O.resample(p, ch, rate).tolist(), (rate, ch, len(p))
    # full-scale square wave: clamping and rounding agree
    sq = np.where((np.arange(48000) // 13) % 2 == 0, 32767, -32768).astype(np.int16)
    assert capi.resample([sq], 1, 48000)[0].tolist() == O.resample(sq, 1, 48000).tolist()

    # analyzer on a 44.1 kHz stereo WAV (what a decoder typically hands over)
    e = synth.make_episode(3, 40.0, 15.0)
    up = np.repeat(e.pcm, 4)                                    # crude 4x zero-order hold: any 44.1 kHz content will do
    up = (up.astype(np.int32) + np.roll(up, 1)) // 2
    stereo = np.stack([up, up], axis=1).reshape(-1).astype(np.int16)
    p = str(tmp_path / "ep-44k.wav")
    synth.write_wav(p, up.astype(np.int16), channels=2, rate=44100)
    fh = capi.Analyzer.from_files([p]).run(0.3)[0]
    total = len(up)
    dur = O.duration_from_secs_f64(total * (1.0 / 44100.0))
    n_open = O.duration_mul_f32(dur, 0.5) * 44100 // NS
    mono = O.resample(stereo[: 2 * n_open], 2, 44100)
    want = O.step_and_timestamp(O.fingerprint(mono), O.duration_from_secs_f32(0.3))
    h, ts = fh.opening_data()
    assert h.tolist() == [x for x, _ in want] and ts.tolist() == [t for _, t in want]
    fh2 = capi.Analyzer.from_files([p]).run_pcm([stereo], channels=2, sample_rate=44100)[0]
    assert fh2.opening_data()[0].tolist() == h.tolist()


def test_cli_analyze_then_search_matches_oracle(lib3, tmp_path):
    """The reference's command lines (README: `needle analyze <dir>`, `needle search <dir>`) through the needle
    binary: .needle.dat files, the displayed results and the skip files all equal the oracle's."""
    import subprocess
    exe = os.path.join(os.path.dirname(capi.LIB_PATH), "..", "bin", "needle")
    paths = []
    for k, e in enumerate(lib3):
        p = str(tmp_path / f"s01e0{k}.wav")
        synth.write_wav(p, e.pcm, channels=2)
        paths.append(p)
    (tmp_path / "readme.txt").write_text("not media")
    r = subprocess.run([exe, "analyze", str(tmp_path)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    hd = O.duration_from_secs_f32(0.3)
    ref = O.analyze_batch([e.pcm[: len(e.pcm) // 2] for e in lib3], 1, hd)
    for p, want in zip(paths, ref):
        rc, disk = O.frame_hashes_read(p[:-4] + ".needle.dat")
        assert rc == 0 and disk.opening == want.opening and disk.md5 == O.header_md5(p)
    r = subprocess.run([exe, "search", str(tmp_path), "--min-opening-duration", "10", "--write-skip-files"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    want = O.run_with_frame_hashes(O.Comparator(min_opening_duration=10 * NS), ref)
    for p, w in zip(paths, want):
        assert f'{p}\n\n* Opening - "{O.format_time(w.opening[0])}"-"{O.format_time(w.opening[1])}"' in r.stdout
        assert open(p[:-4] + ".needle.skip.json").read() == O.skip_file_json(w, O.header_md5(p))
    # default minimum (20 s) on a 20 s intro: nothing found, by both; --no-display prints nothing
    r = subprocess.run([exe, "search", str(tmp_path), "--no-display"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "Opening" not in r.stdout
    # a second analyze finds the cached data (analyzer.rs:338-348); --force recomputes
    r = subprocess.run([exe, "analyze", *paths], capture_output=True, text=True, timeout=300)
    assert r.stdout.count("Skipping analysis for") == 3
    r = subprocess.run([exe, "analyze", "--force", *paths], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "Skipping analysis" not in r.stdout


def test_gpu_reproduces_chromaprints_own_silence_vector():
    """libchromaprint's API test Test2SilenceRawFp replayed call for call against libneedle_chromaprint.so
    (tests/golden/chromaprint_silence.json): start(44100, 1), 130 feeds of 1024 zeros, finish, raw fingerprint =
    three items 627964279 -- through the device resampler and the three fingerprint kernels."""
    import ctypes as C
    import json
    from .test_capi_cpu import _chromaprint_lib
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "chromaprint_silence.json")))
    L = _chromaprint_lib()
    ctx = L.chromaprint_new(1)                                    # CHROMAPRINT_ALGORITHM_TEST2
    assert L.chromaprint_start(ctx, g["sample_rate"], g["channels"]) == 1
    zeroes = np.zeros(g["samples_per_feed"], dtype=np.int16)
    for _ in range(g["feeds"]):
        assert L.chromaprint_feed(ctx, zeroes.ctypes.data, len(zeroes)) == 1
    assert L.chromaprint_finish(ctx) == 1
    fp = C.POINTER(C.c_uint32)()
    n = C.c_int(0)
    assert L.chromaprint_get_raw_fingerprint(ctx, C.byref(fp), C.byref(n)) == 1
    assert [fp[i] for i in range(n.value)] == g["raw_fingerprint"]
    L.chromaprint_dealloc(fp)
    L.chromaprint_free(ctx)
    # and straight through the fingerprint entry point at 11025 Hz
    assert capi.fingerprint([np.zeros(33280, dtype=np.int16)], 1)[0].tolist() == g["raw_fingerprint"]


def test_cli_endings_padding_and_threshold_flags(tmp_path):
    """`needle analyze --include-endings --opening-search-percentage ... --ending-search-percentage ...` then
    `needle search --include-endings --min-*-duration --time-padding --hash-match-threshold`: the flags reach the
    library (same skip files as the in-process Comparator with the same settings, which is checked against the
    oracle in test_endings_and_parameters_match_oracle)."""
    import subprocess
    exe = os.path.join(os.path.dirname(capi.LIB_PATH), "..", "bin", "needle")
    eps = synth.make_library(4, 120.0, 25.0, 22.0)
    paths = []
    for k, e in enumerate(eps):
        p = str(tmp_path / f"e{k}.wav")
        synth.write_wav(p, e.pcm, channels=1)
        paths.append(p)
    r = subprocess.run([exe, "analyze", *paths, "--include-endings", "--opening-search-percentage", "0.4",
                        "--ending-search-percentage=0.35"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    fhs = capi.Analyzer.from_files(paths).with_include_endings(True).with_opening_search_percentage(0.4) \
        .with_ending_search_percentage(0.35).run_pcm([e.pcm for e in eps])
    for p, f in zip(paths, fhs):
        rc, disk = O.frame_hashes_read(p[:-4] + ".needle.dat")
        assert rc == 0
        assert [h for h, _ in disk.opening] == f.opening_data()[0].tolist()
        assert [h for h, _ in disk.ending] == f.ending_data()[0].tolist() and len(disk.ending) > 0
    r = subprocess.run([exe, "search", *paths, "--include-endings", "--min-opening-duration", "12",
                        "--min-ending-duration", "12", "--time-padding", "1.5", "--hash-match-threshold", "9",
                        "--write-skip-files", "--no-display"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip() == "", r.stderr
    cli_skip = [open(p[:-4] + ".needle.skip.json").read() for p in paths]
    for p in paths:
        os.remove(p[:-4] + ".needle.skip.json")
    c = capi.Comparator(paths, include_endings=True, min_opening_duration=12, min_ending_duration=12, time_padding=1.5,
                        hash_match_threshold=9)
    c.run(analyze=False, display=False, write_skip_files=True)
    assert cli_skip == [open(p[:-4] + ".needle.skip.json").read() for p in paths]
    assert all('"ending":[' in s for s in cli_skip)


def test_best_match_on_host_threads_equals_sequential(monkeypatch):
    """At library scale find_best_match (quadratic in a video's candidates) runs for all videos on host threads;
    the results must be those of the sequential walk.  160 videos sharing one planted run: every pair matches, so
    each video has 2 x 159 candidates and the threaded path is taken."""
    rng = np.random.default_rng(5)
    n, length, intro = 160, 400, 120
    hd = O.duration_from_secs_f32(0.3)
    ts = [t for _, t in O.step_and_timestamp(np.zeros(2 * length, dtype=np.uint32), hd)][:length]
    shared = rng.integers(0, 2 ** 32, intro, dtype=np.uint64).astype(np.uint32)
    fhs = []
    for v in range(n):
        h = rng.integers(0, 2 ** 32, length, dtype=np.uint64).astype(np.uint32)
        a = 10 + (13 * v) % 200
        flips = (np.uint32(1) << rng.integers(0, 32, intro).astype(np.uint32)) * (rng.random(intro) < 0.5)
        h[a:a + intro] = shared ^ flips
        fhs.append(capi.FrameHashes.new(list(zip(h.tolist(), ts)), (), hd, ""))
    cmp = capi.Comparator([f"v{v}.wav" for v in range(n)], min_opening_duration=20)
    monkeypatch.setenv("NEEDLE_HOST_THREADS", "1")
    seq = cmp.run_with_frame_hashes(fhs)
    monkeypatch.delenv("NEEDLE_HOST_THREADS")
    par = cmp.run_with_frame_hashes(fhs)
    assert all(r is not None and r.opening is not None for r in seq)
    assert [(r.opening, r.ending) for r in seq] == [(r.opening, r.ending) for r in par]


def test_fingerprint_signal_zoo_bit_exact():
    """Hashes equal the oracle's on signal kinds the synthetic episodes do not contain: noise from +-1 LSB (where the
    feature norm sits around chromaprint's 0.01 cut-off and rows flip between zeroed and normalised) to clipping,
    DC, chirps, impulse trains, a signal that fades in from digital silence."""
    rng = np.random.default_rng(11)
    n = 14 * 11025
    t = np.arange(n) / 11025.0
    zoo = {}
    for amp in (0.6, 1, 2, 5, 20, 100, 1000, 12000, 60000):
        zoo[f"noise{amp}"] = np.clip(np.rint(rng.standard_normal(n) * amp), -32768, 32767)
    zoo["dc"] = np.full(n, 1234.0)
    zoo["dc+lsb"] = 20000 + (rng.random(n) < 0.5)
    zoo["chirp"] = 9000 * np.sin(2 * np.pi * (40 * t + 0.5 * 380 * t * t))
    zoo["impulses"] = np.where(np.arange(n) % 997 == 0, 30000.0, 0.0)
    zoo["nyquist"] = np.where(np.arange(n) % 2 == 0, 32767.0, -32768.0)
    zoo["fade-in"] = 8000 * np.sin(2 * np.pi * 440 * t) * np.clip((t - 5.0) / 6.0, 0, 1) ** 4
    zoo["two-tones"] = 3000 * np.sin(2 * np.pi * 261.63 * t) + 3000 * np.sin(2 * np.pi * 2093.0 * t)
    pcms = [np.asarray(v).astype(np.int16) for v in zoo.values()]
    got = capi.fingerprint(pcms, step=1)
    for name, g, p in zip(zoo, got, pcms):
        assert g.tolist() == O.fingerprint(p).tolist(), name


@pytest.mark.parametrize("n,world", [(7, 2), (4, 3)])       # (4, 3): blocks of 2, 2 and 0 episodes -- a rank that owns none
def test_two_simulated_ranks_on_one_device_equal_single_library(n, world):
    """The multi-GPU plan of tests/dist_plan.py with two Library objects standing in for two ranks on one device:
    each holds the PCM of its own episode block only, fingerprints it into a caller-owned arena, the arenas exchange
    row blocks (what the all-gather does), each scans its own range of the pair list, the run lists are
    concatenated and rank 0 finalises.  Result = the single-library result = the oracle's."""
    from tests import dist_plan as ndist
    eps = synth.make_library(n, 90.0, 20.0)
    lens = [len(e.pcm) for e in eps]
    cmp = capi.Comparator([f"ep{k}.wav" for k in range(n)], min_opening_duration=10)
    L = capi.lib()
    b = ndist.block(n, world)
    libs, arenas, stride = [], [], None
    for rank in range(world):
        first, count = ndist.shard(n, world, rank)
        lib = capi.Library(n)
        lib.set_pcm([e.pcm if first <= k < first + count else None for k, e in enumerate(eps)], lens)
        _, stride = lib.hash_arena()
        buf = capi.DeviceBuffer(b * world * stride * 4)
        zeros = np.zeros(b * world * stride, dtype=np.uint32)
        capi.check(L.needle_hip_memcpy_h2d(buf.ptr, zeros.ctypes.data, zeros.nbytes))
        lib.use_hash_arena(buf.ptr, b * world, stride)
        if count:
            lib.analyze(first, count)
        libs.append(lib)
        arenas.append(buf)
    # the all-gather of row blocks, by hand
    host = [a.to_host(np.uint32, b * world * stride).reshape(b * world, stride) for a in arenas]
    full = np.zeros_like(host[0])
    for rank in range(world):
        full[rank * b:(rank + 1) * b] = host[rank][rank * b:(rank + 1) * b]
    assert not full[n:].any()
    for a in arenas:
        capi.check(L.needle_hip_memcpy_h2d(a.ptr, full.ctypes.data, full.nbytes))
    # pair ranges
    cap = 4096
    runs = []
    for rank in range(world):
        pfirst, pcount = ndist.shard(ndist.pair_count(n), world, rank)
        d_runs, d_count = capi.DeviceBuffer(cap * capi.RUN_DTYPE.itemsize), capi.DeviceBuffer(4)
        libs[rank].search(cmp, pfirst, pcount, d_runs.ptr, cap, d_count.ptr, sync=True)
        found = int(d_count.to_host(np.uint32, 1)[0])
        part = d_runs.to_host(capi.RUN_DTYPE, found)
        assert all(pfirst <= int(p) < pfirst + pcount for p in part["problem"])
        runs.append(part)
    got = libs[0].finalize(cmp, np.concatenate(runs))
    # single library
    one = capi.Library(n)
    one.set_pcm([e.pcm for e in eps], lens)
    one.analyze(0, n)
    d_runs, d_count = capi.DeviceBuffer(cap * capi.RUN_DTYPE.itemsize), capi.DeviceBuffer(4)
    one.search(cmp, 0, one.num_pairs(), d_runs.ptr, cap, d_count.ptr, sync=True)
    want = one.finalize(cmp, d_runs.to_host(capi.RUN_DTYPE, int(d_count.to_host(np.uint32, 1)[0])))
    assert [(r.opening, r.ending) for r in got] == [(r.opening, r.ending) for r in want]
    hd = O.duration_from_secs_f32(0.3)
    ref = O.run_with_frame_hashes(O.Comparator(min_opening_duration=10 * NS),
                                  O.analyze_batch([e.pcm[: len(e.pcm) // 2] for e in eps], 1, hd))
    _same_results(got, ref)
    # a rank can also hand out the FrameHashes of a video another rank fingerprinted (rows arrived by the gather)
    assert libs[0].frame_hashes(n - 1).opening_data()[0].tolist() == [h for h, _ in O.analyze_batch(
        [eps[n - 1].pcm[: lens[n - 1] // 2]], 1, hd)[0].opening]


@pytest.mark.parametrize("bands_per_wave,sparse_max", [(1, None), (2, "0"), (3, "1000"), (8, "2"), (1, "1"), (2, None)])
def test_sampled_kernel_on_sustained_hashes_and_multi_band_workgroups(bands_per_wave, sparse_max, monkeypatch):
    """Hashes that repeat for a few frames, as sustained notes produce, give short chance runs on many diagonals:
    windows survive the early-out far more often than on random hashes and candidates that fail the length test
    get resolved.  Checked against the oracle's table-free scan for every pair of 14 sequences, with one band
    per wave (small launches) and several (large launches: one workgroup walks many bands of its pair), and with
    the survivors of a window's head rows finished one diagonal at a time (the default up to 6 of them), never
    (0: row by row) or always (1000)."""
    monkeypatch.setenv("NEEDLE_HIP_BANDS_PER_WAVE", str(bands_per_wave))
    if sparse_max is not None:
        monkeypatch.setenv("NEEDLE_HIP_SPARSE_MAX", sparse_max)
    rng = np.random.default_rng(100 + bands_per_wave)
    palette = rng.integers(0, 2 ** 32, 40, dtype=np.uint64).astype(np.uint32)     # few distinct "notes"
    seqs = []
    for v in range(14):
        n = int(rng.integers(900, 2400))
        notes = rng.integers(0, len(palette), n // 3 + 2)
        h = np.repeat(palette[notes], rng.integers(1, 7, len(notes)))[:n].copy()
        h ^= (np.uint32(1) << rng.integers(0, 32, n).astype(np.uint32)) * (rng.random(n) < 0.8)   # a flipped bit or none
        seqs.append(h)
    shared = seqs[0][100:400].copy()
    for v in range(1, 14, 2):
        a = 50 + 37 * v
        seqs[v][a:a + 300] = shared
    min_len = 30
    problems = [(i, j, min_len) for i in range(14) for j in range(i + 1, 14)]
    got = _gpu_runs(seqs, problems, 10)
    total, runs = O.diagonal_runs_all_pairs(seqs, 10, min_len, threads=4, capacity=200000)
    assert total == len(runs) and total > 20
    want = {}
    for p, a, b, L in runs.tolist():
        want.setdefault(p, []).append((a, b, L))
    assert {p: sorted((a, b, L) for a, b, L, _, _ in v) for p, v in got.items()} == {p: sorted(v) for p, v in want.items()}


def test_fingerprint_batch_split_into_chunks_equals_one_launch(lib3, monkeypatch):
    """Huge batches are fingerprinted in chunks that reuse the workspaces and the descriptor buffer; with the chunk
    bound forced down to a few hundred frames the same ragged batch must give the same items."""
    pcms = [lib3[0].pcm[: 20 * 11025], lib3[1].pcm[: 3 * 11025 + 1], np.zeros(10, np.int16), lib3[2].pcm[: 31 * 11025],
            lib3[1].pcm[: 9 * 11025], lib3[0].pcm[: 4096 + 19 * 1365]]
    whole = capi.fingerprint(pcms, step=2)
    for bound in ("1", "100", "300"):                      # 1: every stream its own chunk
        monkeypatch.setenv("NEEDLE_HIP_MAX_FRAMES_PER_CHUNK", bound)
        split = capi.fingerprint(pcms, step=2)
        assert [s.tolist() for s in split] == [w.tolist() for w in whole]
    monkeypatch.delenv("NEEDLE_HIP_MAX_FRAMES_PER_CHUNK")
    assert whole[0].tolist() == O.fingerprint(pcms[0])[::2].tolist()
    # and the host entry point's upload batches (2 GiB of PCM each by default), also through the resampler
    for bound in ("1", "250000"):
        monkeypatch.setenv("NEEDLE_HIP_MAX_BATCH_VALUES", bound)
        assert [s.tolist() for s in capi.fingerprint(pcms, step=2)] == [w.tolist() for w in whole]
    monkeypatch.delenv("NEEDLE_HIP_MAX_BATCH_VALUES")


def test_run_list_larger_than_the_first_buffer():
    """More than 65 536 runs in one call: the host entry point sizes its buffer from the count and scans again.
    Threshold 40 makes every cell match, so each diagonal of each of 140 problems is one run."""
    rng = np.random.default_rng(12)
    seqs = [_rand_hashes(rng, 300), _rand_hashes(rng, 260)]
    problems = [(0, 1, 1)] * 140
    got = _gpu_runs(seqs, problems, 40)
    want = _oracle_runs(seqs[0], seqs[1], 40, 1)
    assert len(want) == 300 + 260 - 3 and 140 * len(want) > 65536
    assert len(got) == 140 and all(got[p] == want for p in range(140))


def test_device_resident_fingerprint_entry_point(lib3):
    """needle_hip_fingerprint_device: PCM already in HBM (offsets in s16 values), kept items written to a device
    buffer at caller-chosen offsets -- the entry point a decoder that writes into device memory would use."""
    import ctypes as C
    L = capi.lib()
    pcms = [np.ascontiguousarray(lib3[0].pcm[: 9 * 11025]), np.ascontiguousarray(lib3[2].pcm[: 6 * 11025 + 3])]
    offs = [0, (len(pcms[0]) + 7) & ~7]
    total = offs[1] + len(pcms[1])
    d_pcm = capi.DeviceBuffer(total * 2)
    for p, o in zip(pcms, offs):
        capi.check(L.needle_hip_memcpy_h2d(d_pcm.ptr + 2 * o, p.ctypes.data, p.nbytes))
    step = 2
    kept = [L.needle_hip_fingerprint_num_kept(len(p), step) for p in pcms]
    item_offs = [5, 5 + kept[0] + 11]                       # gaps on purpose
    d_items = capi.DeviceBuffer(4 * (item_offs[1] + kept[1] + 3))
    u64 = C.c_uint64 * 2
    capi.check(L.needle_hip_fingerprint_device(d_pcm.ptr, u64(*offs), u64(*[len(p) for p in pcms]), 2, 1, step,
                                               d_items.ptr, u64(*item_offs), True))
    items = d_items.to_host(np.uint32, item_offs[1] + kept[1] + 3)
    for p, o, k in zip(pcms, item_offs, kept):
        assert items[o:o + k].tolist() == O.fingerprint(p)[::step].tolist()


def test_concurrent_callers_get_their_own_results(lib3):
    """Several host threads inside the library at once (ctypes releases the GIL): the shared per-device workspaces
    and staging buffers are serialised by the library, so every caller gets what it would get alone."""
    import threading
    rng = np.random.default_rng(4)
    jobs = []
    for k in range(6):
        pcm = lib3[k % 3].pcm[k * 11025: (k + 9) * 11025]
        src, dst = _rand_hashes(rng, 300 + 40 * k), _rand_hashes(rng, 500)
        dst[30:130] = src[20:120]
        jobs.append((pcm, src, dst))
    want = [(capi.fingerprint([p])[0].tolist(), _gpu_runs([s, d], [(0, 1, 25)], 10)) for p, s, d in jobs]
    got = [None] * len(jobs)

    def work(i):
        p, s, d = jobs[i]
        for _ in range(5):
            got[i] = (capi.fingerprint([p])[0].tolist(), _gpu_runs([s, d], [(0, 1, 25)], 10))

    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(jobs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert got == want


def test_wav_sample_formats_reach_the_same_hashes(tmp_path):
    """The WAV reader converts 8/24/32-bit integer and float PCM to s16 the way the reference's resampler output
    format does (top 16 bits; floats scaled by 2^15, rounded, clipped): analysing such a file equals analysing the
    s16 stream that rule gives."""
    e = synth.make_episode(5, 40.0, 12.0)
    pcm = e.pcm[: 30 * 11025]

    def write(path, fmt, bits, payload):
        with open(path, "wb") as f:
            n = len(payload)
            f.write(b"RIFF" + (36 + n).to_bytes(4, "little") + b"WAVE")
            f.write(b"fmt " + (16).to_bytes(4, "little") + fmt.to_bytes(2, "little") + (1).to_bytes(2, "little")
                    + (11025).to_bytes(4, "little") + (11025 * bits // 8).to_bytes(4, "little")
                    + (bits // 8).to_bytes(2, "little") + bits.to_bytes(2, "little"))
            f.write(b"data" + n.to_bytes(4, "little") + payload)

    rng = np.random.default_rng(8)
    x = pcm.astype(np.int64)
    s24 = (x << 8) + rng.integers(0, 256, len(x))                 # low byte is dropped by the reader
    s32 = (x << 16) + rng.integers(0, 65536, len(x))
    u8 = ((x >> 8) + 128).astype(np.uint8)
    flt = (pcm.astype(np.float32) / np.float32(32768.0))
    cases = {
        "s24": (1, 24, b"".join(int(v).to_bytes(3, "little", signed=True) for v in s24[:40000]), pcm[:40000]),
        "s32": (1, 32, s32.astype("<i4").tobytes(), pcm),
        "u8": (1, 8, u8.tobytes(), (((u8.astype(np.int32) - 128) << 8)).astype(np.int16)),
        "f32": (3, 32, flt.astype("<f4").tobytes(), pcm),
        "f64": (3, 64, (pcm.astype(np.float64) * 1.7 / 32768.0).astype("<f8").tobytes(),
                np.clip(np.rint(pcm.astype(np.float64) * 1.7), -32768, 32767).astype(np.int16)),
    }
    hd = O.duration_from_secs_f32(0.3)
    for name, (fmt, bits, payload, expect) in cases.items():
        p = str(tmp_path / f"{name}.wav")
        write(p, fmt, bits, payload)
        fh = capi.Analyzer.from_files([p]).run(0.3)[0]
        want = O.analyze_batch([expect[: len(expect) // 2]], 1, hd)[0]
        assert fh.opening_data()[0].tolist() == [h for h, _ in want.opening], name


def test_file_analyzer_mixed_library_bounded_batches_and_threads(tmp_path, monkeypatch):
    """Analyzer::run over files of different channel counts, sample rates and encodings in ONE call: only the
    search windows are read from each file, by reader threads, into a ring of pinned slabs that feeds the device, one
    pass per distinct (channels, rate).  Every video must come out as if analysed alone from its whole
    PCM (run_pcm, itself pinned to the oracle above), whatever the batch size and reader count; one resampled
    stereo file is also checked against the oracle chain directly, opening and ending."""
    lib = synth.make_library(6, 50.0, 12.0)
    spec = [(1, 11025), (2, 44100), (1, 11025), (1, 48000), (2, 11025), (2, 44100)]
    paths, streams = [], []
    for k, (ch, rate) in enumerate(spec):
        x = lib[k].pcm
        if rate != 11025:
            m = {44100: 4, 48000: 5}[rate]                      # zero-order hold then a 2-tap smoother: any content will do
            x = np.repeat(x, m)[: len(x) * m - 7 * k]
            x = ((x.astype(np.int32) + np.roll(x, 1)) // 2).astype(np.int16)
        inter = x if ch == 1 else np.stack([x, np.roll(x, 3)], axis=1).reshape(-1).astype(np.int16)
        p = str(tmp_path / f"ep-{k}.wav")
        with open(p, "wb") as f:                                  # junk chunk before fmt, odd-sized chunk before data
            body = (b"LIST" + (5).to_bytes(4, "little") + b"hello\0"
                    + b"fmt " + (16).to_bytes(4, "little") + (1).to_bytes(2, "little") + ch.to_bytes(2, "little")
                    + rate.to_bytes(4, "little") + (rate * ch * 2).to_bytes(4, "little")
                    + (ch * 2).to_bytes(2, "little") + (16).to_bytes(2, "little")
                    + b"data" + (inter.nbytes).to_bytes(4, "little") + inter.astype("<i2").tobytes())
            f.write(b"RIFF" + (4 + len(body)).to_bytes(4, "little") + b"WAVE" + body)
        paths.append(p)
        streams.append(inter)

    def alone(k):
        ch, rate = spec[k]
        a = capi.Analyzer.from_files([paths[k]]).with_include_endings(True)
        return a.run_pcm([streams[k]], channels=ch, sample_rate=rate)[0]

    want = [alone(k) for k in range(len(spec))]

    def same(got):
        for k, (g, w) in enumerate(zip(got, want)):
            for part in ("opening_data", "ending_data"):
                gh, gt = getattr(g, part)()
                wh, wt = getattr(w, part)()
                assert gh.tolist() == wh.tolist() and gt.tolist() == wt.tolist(), (k, part)
            assert g.md5() == O.header_md5(paths[k])

    # slab size of the upload ring (default 8 MiB; 4 KiB = hundreds of segments per window, many trips round the
    # ring; 48 bytes = stereo frames split across slabs as finely as alignment allows), device batch size, readers
    for slab, batch, threading in [(None, None, True), ("4096", None, True), ("4096", "300000", False),
                                   ("48", "1", True)]:
        for name, v in (("NEEDLE_HIP_UPLOAD_SLAB_BYTES", slab), ("NEEDLE_HIP_MAX_BATCH_VALUES", batch)):
            if v is None:
                monkeypatch.delenv(name, raising=False)
            else:
                monkeypatch.setenv(name, v)
        if slab == "48":   # keep the finest split affordable: three short files
            sub = [0, 1, 4]
            got = capi.Analyzer.from_files([paths[k] for k in sub]).with_include_endings(True).run(0.3, threading=threading)
            for g, k in zip(got, sub):
                assert g.opening_data()[0].tolist() == want[k].opening_data()[0].tolist()
                assert g.ending_data()[1].tolist() == want[k].ending_data()[1].tolist()
            continue
        same(capi.Analyzer.from_files(paths).with_include_endings(True).run(0.3, threading=threading))
    monkeypatch.delenv("NEEDLE_HIP_UPLOAD_SLAB_BYTES", raising=False)
    monkeypatch.delenv("NEEDLE_HIP_MAX_BATCH_VALUES", raising=False)

    # the oracle chain for video 1 (44.1 kHz stereo): windows at the stream's rate, resample, fingerprint, timestamps
    k, ch, rate = 1, 2, 44100
    total = len(streams[k]) // ch
    dur = O.duration_from_secs_f64(total * (1.0 / rate))
    n_open = O.duration_mul_f32(dur, 0.5) * rate // NS
    seek = O.duration_mul_f32(dur, 1.0 - 0.25)
    first = seek * rate // NS
    hd = O.duration_from_secs_f32(0.3)
    op = O.step_and_timestamp(O.fingerprint(O.resample(streams[k][: ch * n_open], ch, rate)), hd)
    en = O.step_and_timestamp(O.fingerprint(O.resample(streams[k][ch * first:], ch, rate)), hd, seek)
    h, ts = want[k].opening_data()
    assert h.tolist() == [x for x, _ in op] and ts.tolist() == [t for _, t in op]
    h, ts = want[k].ending_data()
    assert h.tolist() == [x for x, _ in en] and ts.tolist() == [t for _, t in en]


def test_file_analyzer_truncated_and_unreadable_files(tmp_path):
    """A data chunk that claims more bytes than the file holds ends at the end of the file (the reader never reads
    past it); a file that is not RIFF/WAVE is an error, not a crash."""
    e = synth.make_episode(2, 40.0, 10.0)
    p = str(tmp_path / "short.wav")
    synth.write_wav(p, e.pcm)
    raw = open(p, "rb").read()
    cut = 44 + 2 * (len(e.pcm) * 3 // 4) + 1                     # ends in the middle of a sample
    open(p, "wb").write(raw[:cut])
    fh = capi.Analyzer.from_files([p]).run(0.3)[0]
    kept = e.pcm[: len(e.pcm) * 3 // 4]
    want = O.analyze_batch([kept[: O.duration_mul_f32(O.duration_from_secs_f64(len(kept) * (1.0 / 11025.0)), 0.5)
                                  * 11025 // NS]], 1, O.duration_from_secs_f32(0.3))[0]
    assert fh.opening_data()[0].tolist() == [h for h, _ in want.opening]
    bad = str(tmp_path / "bad.wav")
    open(bad, "wb").write(b"not a wave file at all" * 10)
    with pytest.raises(capi.NeedleError):
        capi.Analyzer.from_files([bad]).run(0.3)


@pytest.mark.timeout(120)
def test_upload_ring_survives_a_failing_reader(tmp_path, monkeypatch):
    """A read error in the middle of a streamed upload (injected at a chosen segment) ends the call with an error
    code -- readers, uploader and ring all wind down -- and the next call works and gives the right hashes: first,
    last and a middle segment, with slabs small enough that the ring wraps many times."""
    eps = synth.make_library(4, 60.0, 10.0)
    paths = []
    for k, e in enumerate(eps):
        p = str(tmp_path / f"e{k}.wav")
        synth.write_wav(p, e.pcm)
        paths.append(p)
    want = [fh.opening_data()[0].tolist() for fh in capi.Analyzer.from_files(paths).run(0.3)]
    monkeypatch.setenv("NEEDLE_HIP_UPLOAD_SLAB_BYTES", "8192")
    total_segments = sum(-(-(len(e.pcm) // 2 * 2) // 8192) for e in eps)
    for fail_at in (0, 37, total_segments - 1):
        monkeypatch.setenv("NEEDLE_HIP_TEST_FAIL_READ_AT", str(fail_at))
        with pytest.raises(capi.NeedleError) as err:
            capi.Analyzer.from_files(paths).run(0.3)
        assert err.value.name == "IOError"
        monkeypatch.delenv("NEEDLE_HIP_TEST_FAIL_READ_AT")
        got = [fh.opening_data()[0].tolist() for fh in capi.Analyzer.from_files(paths).run(0.3)]
        assert got == want
