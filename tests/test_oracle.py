"""CPU tests that pin the oracle (oracle/*.c) before anything is compared with it.

What the reference itself offers as pins for this path is thin: one golden value —
the header MD5 in its analyzer snapshot — plus the code of comparator.rs/data.rs.  So besides that
golden value the oracle is held against (a) known values of std::time::Duration documented by Rust,
(b) a hand-traced known answer for the DP, (c) independent restatements written
here in Python/numpy (np_chromaprint.py; a brute-force diagonal scan), and (d) format round trips.
"""
import hashlib
import json
import os
import struct

import numpy as np
import pytest

from needle_amd import synth
from oracle import oracle as O

from . import np_chromaprint as NP

HERE = os.path.dirname(os.path.abspath(__file__))
NS = O.NS


# ---- golden value held by the reference's own tests ----------------------------------------------------
def test_header_md5_matches_reference_snapshot(tmp_path):
    """needle/src/audio/snapshots/needle__audio__analyzer__test__analyzer.snap:43,75 records
    md5: "759c6a520c5ce70359fdff38c4be6b98" for needle/resources/sample-5s.mp4 (util.rs:99-105: MD5 of
    the first 8192 bytes).  The fixture is those 8192 bytes of the reference's test media."""
    head = open(os.path.join(HERE, "golden", "sample-5s.header8k.bin"), "rb").read()
    assert len(head) == 8192
    assert O.md5_hex(head) == "759c6a520c5ce70359fdff38c4be6b98"
    p = tmp_path / "video.mp4"
    p.write_bytes(head + b"tail bytes beyond the header do not matter")
    assert O.header_md5(str(p)) == "759c6a520c5ce70359fdff38c4be6b98"
    short = tmp_path / "short.mp4"
    short.write_bytes(head[:8191])
    assert O.header_md5(str(short)) is None  # read_exact fails on files shorter than 8 KiB


def test_md5_against_hashlib():
    rng = np.random.default_rng(1)
    for n in [0, 1, 55, 56, 57, 63, 64, 65, 119, 120, 8192, 10001]:
        data = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert O.md5_hex(data) == hashlib.md5(data).hexdigest()


# ---- std::time::Duration ---------------------------------------------------------------------------------
def test_duration_known_values():
    # documented in std: Duration::from_secs_f32(2.7) == Duration::new(2, 700_000_048)
    assert O.duration_from_secs_f32(2.7) == 2_700_000_048
    assert O.duration_from_secs_f32(0.999e-9) == 1          # rounds up to 1 ns (std doc example)
    assert O.duration_from_secs_f32(1e-20) == 0
    assert O.duration_from_secs_f32(3e9) == 3_000_000_000 * NS   # no fractional part (exp >= mantissa bits)
    assert O.duration_from_secs_f64(2.7) == 2_700_000_000
    assert O.duration_from_secs_f64(0.999e-9) == 1
    # the reference's default hash duration: from_secs_f32(0.3) (data.rs:135, lib.rs:478)
    assert O.duration_from_secs_f32(0.3) == 300_000_012


def test_duration_from_f32_is_exact_rounding():
    rng = np.random.default_rng(2)
    for v in np.concatenate([rng.random(200).astype(np.float32) * 2000, rng.random(100).astype(np.float32)]):
        # exact: f32 * 1e9 fits a double exactly (24 + 21 significant bits); rint = round-half-even
        want = int(np.rint(np.float64(v) * 1e9))
        assert O.duration_from_secs_f32(float(v)) == want


def test_timestamps_follow_analyzer_rule():
    """analyzer.rs:293-311: step = 300 ms / 123 ms = 2; ts_i = 2.6 s + Duration(0.123 s).mul_f32(i)."""
    assert O.delay_ms() == 2600 and O.item_duration_ms() == 123
    raw = np.arange(11, dtype=np.uint32) + 100
    out = O.step_and_timestamp(raw, 300_000_012)
    assert [h for h, _ in out] == [100, 102, 104, 106, 108, 110]
    for k, (_, ts) in enumerate(out):
        i = 2 * k
        prod = np.float32(i) * np.float32(0.123)
        assert ts == 2_600_000_000 + int(np.rint(np.float64(prod) * 1e9))
    assert out[1][1] == 2_846_000_007
    with_seek = O.step_and_timestamp(raw, 300_000_012, seek_to_ns=1080 * NS)
    assert [t for _, t in with_seek] == [t + 1080 * NS for _, t in out]
    with pytest.raises(ValueError):
        O.step_and_timestamp(raw, 100_000_000)  # hash_duration < item duration: step_by(0) panics upstream
    assert len(O.step_and_timestamp(raw, 123_000_000)) == 11  # step 1


# ---- chromaprint restatement -------------------------------------------------------------------------------
def _tone(seconds, seed):
    rng = np.random.default_rng(seed)
    t = np.arange(int(seconds * 11025)) / 11025.0
    x = np.zeros_like(t)
    for f in rng.choice([220.0, 277.18, 329.63, 440.0, 523.25, 659.25, 880.0], 4, replace=False):
        x += rng.uniform(0.1, 0.25) * np.sin(2 * np.pi * f * t * (1 + 0.02 * np.sin(t)))
    x += 0.01 * rng.standard_normal(len(t))
    return np.clip(np.rint(x * 32767), -32768, 32767).astype(np.int16)


def test_counts():
    assert O.num_frames(4095) == 0 and O.num_frames(4096) == 1 and O.num_frames(4096 + 1364) == 1
    assert O.num_frames(4096 + 1365) == 2
    assert O.num_items(4096 + 18 * 1365) == 0 and O.num_items(4096 + 19 * 1365) == 1
    # SURVEY.md §8 size table
    assert (O.num_frames(496125), O.num_items(496125)) == (361, 342)
    assert (O.num_frames(7938000), O.num_items(7938000)) == (5813, 5794)
    assert (O.num_frames(14883750), O.num_items(14883750)) == (10901, 10882)


def test_oracle_fingerprint_matches_independent_numpy_statement():
    for seed in (3, 4):
        pcm = _tone(12.0, seed)
        items, chroma, feats, margin = O.fingerprint(pcm, debug=True)
        ref_chroma = NP.chroma_features(pcm)
        assert np.allclose(chroma, ref_chroma, rtol=1e-11, atol=0)
        ref_items = NP.fingerprint(pcm)
        assert margin > 1e-9, "a quantiser decision sits on a threshold: pick another seed"
        assert items.tolist() == ref_items.tolist()
        assert len(items) == O.num_items(len(pcm))


def test_oracle_stereo_downmix_and_edges():
    pcm = _tone(6.0, 5)
    mono = O.fingerprint(pcm)
    stereo = O.fingerprint(np.repeat(pcm, 2), channels=2)   # L = R, as the reference feeds (analyzer.rs:183-185,218)
    assert mono.tolist() == stereo.tolist()
    # (L + R) / 2 truncates toward zero like C: (-3 + 0) / 2 = -1
    lr = np.zeros(2 * len(pcm), dtype=np.int16)
    lr[0::2] = pcm
    lr[1::2] = np.roll(pcm, 1)
    want = ((pcm.astype(np.int32) + np.roll(pcm, 1).astype(np.int32)) / 2).astype(np.int32)  # trunc toward zero
    assert O.fingerprint(lr, channels=2).tolist() == O.fingerprint(want.astype(np.int16)).tolist()
    assert len(O.fingerprint(np.zeros(4095, np.int16))) == 0
    assert len(O.fingerprint(np.zeros(0, np.int16))) == 0
    # digital silence: norm < 0.01 -> zero rows -> every filter value log(1/1) = 0
    silent = O.fingerprint(np.zeros(4096 + 30 * 1365, np.int16))
    assert len(silent) == 12 and len(set(silent.tolist())) == 1


def test_simhash32_known_answers():
    assert O.simhash32([0x44444444, 0x55555555]) == 0x44444444   # a tie (+1-1) leaves the bit clear
    assert O.simhash32([0xFFFFFFFF]) == 0xFFFFFFFF
    assert O.simhash32([]) == 0
    assert O.simhash32([0xF0F0F0F0, 0xFF00FF00, 0xFFFF0000]) == 0xFFF0F000


# ---- FrameHashes on disk (data.rs + bincode 1.3) -------------------------------------------------------------
def test_needle_dat_layout_matches_hand_assembled_bytes(tmp_path):
    """SURVEY.md Appendix B, assembled by hand: u32 version index, u32 data tag, u64 len + n*(u32,u64,u32),
    same for ending, (u64,u32) hash_duration, u64 len + md5 bytes."""
    fh = O.FrameHashes([(0xDEADBEEF, 2_600_000_000), (7, 2_846_000_007)], [(9, 1_082_600_000_000)],
                       300_000_012, "759c6a520c5ce70359fdff38c4be6b98")
    path = str(tmp_path / "ep.needle.dat")
    assert O.frame_hashes_write(path, fh) == 0
    want = struct.pack("<II", 0, 0)
    want += struct.pack("<Q", 2) + struct.pack("<IQI", 0xDEADBEEF, 2, 600_000_000) + struct.pack("<IQI", 7, 2, 846_000_007)
    want += struct.pack("<Q", 1) + struct.pack("<IQI", 9, 1082, 600_000_000)
    want += struct.pack("<QI", 0, 300_000_012)
    want += struct.pack("<Q", 32) + b"759c6a520c5ce70359fdff38c4be6b98"
    got = open(path, "rb").read()
    assert got == want and len(got) == 76 + 16 * 3
    rc, back = O.frame_hashes_read(path)
    assert rc == 0 and back == fh
    open(path, "wb").write(want[:-5])
    assert O.frame_hashes_read(path)[0] == 2          # truncated: bincode error
    open(path, "wb").write(struct.pack("<I", 1) + want[4:])
    assert O.frame_hashes_read(path)[0] == 2          # unknown enum variant index
    assert O.frame_hashes_read(str(tmp_path / "missing.needle.dat"))[0] == 1


# ---- comparator.rs ---------------------------------------------------------------------------------------------
def _brute_runs(src, dst, thr):
    """Every maximal diagonal run over cells i>=1, j>=1, as (i_end, j_end, L) — written independently of the DP."""
    out = []
    n, m = len(src), len(dst)
    for d in range(-(n - 2), m - 1):
        run = 0
        i = max(1, 1 - d)
        while i <= n - 1 and i + d <= m - 1:
            if bin(src[i] ^ dst[i + d]).count("1") <= thr:
                run += 1
            else:
                if run:
                    out.append((i - 1, i - 1 + d, run))
                run = 0
            i += 1
        if run:
            out.append((i - 1, i - 1 + d, run))
    return sorted(out)


def test_lcs_hand_traced_known_answer():
    """SURVEY.md Appendix C (derived by reading comparator.rs:157-250)."""
    src = [0xAAAA0000, 0x11111111, 0x22222222, 0x33333333, 0x44444444, 0x55555555, 0x0F0F0F0F]
    dst = [0x11111111, 0xFFFF0000, 0x11111111, 0x22222222, 0x33333333, 0x44444445, 0xF0F0F0F0, 0x55555555]
    cmp = O.Comparator(hash_match_threshold=1, min_opening_duration=0)
    ents = O.longest_common_hash_match(cmp, [(h, i * NS) for i, h in enumerate(src)],
                                       [(h, i * NS) for i, h in enumerate(dst)], 0, 0)
    # BinaryHeap array order: pushes were (L=1) then (L=4); the larger sifts to the root
    assert [(e["score"], e["src_end_idx"], e["dst_end_idx"]) for e in ents] == [(4, 4, 5), (1, 5, 7)]
    assert (ents[0]["src_start"], ents[0]["src_end"]) == (0, 4 * NS)       # start index is one BEFORE the run
    assert (ents[0]["dst_start"], ents[0]["dst_end"]) == (1 * NS, 5 * NS)
    assert (ents[0]["src_match_hash"], ents[0]["dst_match_hash"]) == (0x22220000, 0x33330001)
    assert (ents[1]["src_match_hash"], ents[1]["dst_match_hash"]) == (0x44444444, 0x50505050)
    # src[1] == dst[0] is not a match: column 0 is forced to 0 (comparator.rs:179-180)
    assert all(not (e["dst_end_idx"] == 0) for e in ents)


def test_lcs_dp_equals_brute_force_diagonals():
    rng = np.random.default_rng(7)
    for n, m, thr in [(40, 55, 10), (64, 64, 12), (3, 90, 14), (2, 2, 32), (1, 10, 10), (25, 1, 10)]:
        src = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
        dst = rng.integers(0, 2 ** 32, m, dtype=np.uint64).astype(np.uint32)
        if n > 20 and m > 20:
            dst[5:18] = src[3:16]  # plant an exact run
        cmp = O.Comparator(hash_match_threshold=thr, min_opening_duration=0)
        ents = O.longest_common_hash_match(cmp, [(int(h), i * NS) for i, h in enumerate(src)],
                                           [(int(h), i * NS) for i, h in enumerate(dst)], 0, 0)
        got = sorted((e["src_end_idx"], e["dst_end_idx"], e["score"]) for e in ents)
        assert got == _brute_runs(src.tolist(), dst.tolist(), thr)
        # heap property of the returned array (std BinaryHeap backing vector)
        keys = [(e["score"], e["src_start"], e["src_end"], e["dst_start"], e["dst_end"]) for e in ents]
        assert all(keys[(k - 1) // 2] >= keys[k] for k in range(1, len(keys)))


def test_lcs_duration_filter_uses_start_before_run():
    """A run of L cells spans L steps of timestamp (start idx = i - L), comparator.rs:206-223."""
    n = 60
    src = [(i * 7919 + 1) & 0xFFFFFFFF for i in range(n)]
    dst = [(~h) & 0xFFFFFFFF for h in src]
    for k in range(20, 31):  # 11 matched cells at i = j = 20..30
        dst[k] = src[k]
    ts = [(h, i * 1_000_000_000) for i, h in enumerate(src)]
    td = [(h, i * 1_000_000_000) for i, h in enumerate(dst)]
    for min_s, expect in [(11, 1), (12, 0)]:
        cmp = O.Comparator(hash_match_threshold=0, min_opening_duration=min_s * NS)
        ents = O.longest_common_hash_match(cmp, ts, td, 0, 0)
        assert len(ents) == expect
    e = O.longest_common_hash_match(O.Comparator(hash_match_threshold=0, min_opening_duration=0), ts, td, 0, 0)[0]
    assert (e["score"], e["src_start"], e["src_end"]) == (11, 19 * NS, 30 * NS)


def _planted_library(n_videos, n_hashes, intro_len, seed, spacing=246_000_000):
    rng = np.random.default_rng(seed)
    intro = rng.integers(0, 2 ** 32, intro_len, dtype=np.uint64).astype(np.uint32)
    fhs = []
    for v in range(n_videos):
        h = rng.integers(0, 2 ** 32, n_hashes, dtype=np.uint64).astype(np.uint32)
        off = 5 + 3 * v
        noisy = intro.copy()
        flips = rng.integers(0, 32, intro_len)
        noisy ^= (np.uint32(1) << flips.astype(np.uint32)) * (rng.random(intro_len) < 0.5)
        h[off:off + intro_len] = noisy
        fhs.append(O.FrameHashes([(int(x), 2_600_000_000 + i * spacing) for i, x in enumerate(h)], [], 300_000_012))
    return fhs


def test_run_with_frame_hashes_finds_planted_intro_and_skips_unmatched_videos():
    fhs = _planted_library(4, 200, 100, seed=11)
    rng = np.random.default_rng(12)
    fhs.append(O.FrameHashes([(int(x), 2_600_000_000 + i * 246_000_000)
                              for i, x in enumerate(rng.integers(0, 2 ** 32, 200, dtype=np.uint64))], [], 300_000_012))
    res = O.run_with_frame_hashes(O.Comparator(), fhs)
    assert res[4] is None                      # no match: the reference pushes no result (comparator.rs:608-617)
    for v in range(4):
        assert res[v] is not None and res[v].opening is not None and res[v].ending is None
        start, end = res[v].opening
        off = 5 + 3 * v
        # run cells off..off+99 -> start idx off-1 (or later if the first cells mismatch), end idx off+99
        assert abs(start - (2_600_000_000 + (off - 1) * 246_000_000)) <= 2 * 246_000_000
        assert abs(end - (2_600_000_000 + (off + 99) * 246_000_000 - 300_000_012)) <= 2 * 246_000_000
    # threads change nothing (rayon par_iter preserves order, comparator.rs:553-563)
    assert O.run_with_frame_hashes(O.Comparator(), fhs, threads=4) == res


def test_run_with_frame_hashes_endings_and_errors():
    fhs = _planted_library(3, 150, 90, seed=13)
    with pytest.raises(RuntimeError):          # FrameHashDataNoEnding (comparator.rs:271-273)
        O.run_with_frame_hashes(O.Comparator(include_endings=True), fhs)
    for f in fhs:
        f.ending = [(h, t + 1000 * NS) for h, t in f.opening]
    res = O.run_with_frame_hashes(O.Comparator(include_endings=True), fhs)
    for r in res:
        assert r.opening is not None and r.ending is not None
        assert r.ending[0] == r.opening[0] + 1000 * NS
    with pytest.raises(OverflowError):         # end - padding - hash_duration underflows: Rust panics
        O.run_with_frame_hashes(O.Comparator(time_padding=10_000 * NS), fhs)
    # threshold 0: the bias bound is 0, no candidate links to itself, opening stays None but a result exists
    exact = _planted_library(2, 150, 90, seed=14)
    exact[1].opening[8:98] = [(h, exact[1].opening[8 + k][1]) for k, (h, _) in enumerate(exact[0].opening[5:95])]
    res0 = O.run_with_frame_hashes(O.Comparator(hash_match_threshold=0), exact)
    assert res0[0] is not None and res0[0].opening is None


def test_skip_file_json_and_format_time():
    # README.md:51: {"opening":null,"ending":[1331.6644,1419.0249],"md5":"..."}
    r = O.SearchResult(None, (O.duration_from_secs_f32(1331.6644), O.duration_from_secs_f32(1419.0249)))
    text = O.skip_file_json(r, "14bfa97f85d86f74e1ab5a26066f9181")
    assert text == '{"opening":null,"ending":[1331.6644,1419.0249],"md5":"14bfa97f85d86f74e1ab5a26066f9181"}'
    assert json.loads(text)["ending"] == [1331.6644, 1419.0249]
    r2 = O.SearchResult((20 * NS, 3_092_000_014), None)
    assert O.skip_file_json(r2, "x") == '{"opening":[20.0,3.092],"ending":null,"md5":"x"}'
    assert O.skip_file_json(O.SearchResult(None, None), "x") == ""      # comparator.rs:336-338: nothing written
    assert O.format_time(43 * NS) == "00:43s" and O.format_time(132_900_000_000) == "02:12s"
    assert O.format_time(3_723 * NS) == "62:03s"


# ---- committed golden vectors (tests/golden/make_golden.py) ---------------------------------------------------
def test_oracle_reproduces_committed_golden_vectors():
    from needle_amd import synth
    g = json.load(open(os.path.join(HERE, "golden", "config1.json")))
    eps = synth.make_library(3, 90.0, 20.0)
    assert [int(e.pcm.astype("int64").sum()) for e in eps] == g["pcm_crc"], "synthetic generator drifted"
    fhs = O.analyze_batch([e.pcm[: len(e.pcm) // 2] for e in eps], 1, g["hash_duration_ns"])
    assert [[[h, t] for h, t in f.opening] for f in fhs] == g["opening"]
    assert O.fingerprint(eps[0].pcm[: len(eps[0].pcm) // 2]).tolist() == g["raw_items_first_episode"]
    for min_s, want in g["results"].items():
        res = O.run_with_frame_hashes(O.Comparator(min_opening_duration=int(min_s) * NS), fhs)
        assert [None if r is None else list(r.opening) if r.opening else [] for r in res] == want


# ---- resampler front-end (own specification, oracle/ora_resample.h) -----------------------------------------------
def test_resampler_oracle_is_a_sane_low_pass_to_11025():
    for rate in (44100, 48000, 22050, 32000, 8000):
        n = rate * 2
        t = np.arange(n) / rate
        hi = 9000.0 if rate > 20000 else 0.0                 # above the new Nyquist: must vanish
        x = (8000 * np.sin(2 * np.pi * 440 * t) + 3000 * np.sin(2 * np.pi * hi * t)).astype(np.int16)
        y = O.resample(x, 1, rate)
        assert len(y) == -(-n * 11025 // rate)
        ref = 8000 * np.sin(2 * np.pi * 440 * np.arange(len(y)) / 11025)
        assert np.abs(y[300:-300] - ref[300:-300]).max() <= 3
        stereo = np.repeat(x, 2)
        assert O.resample(stereo, 2, rate).tolist() == y.tolist()
    x = (np.arange(5000) % 700 - 350).astype(np.int16)
    assert O.resample(x, 1, 11025).tolist() == x.tolist()     # same rate: identity
    lr = np.stack([x, -x // 2], axis=1).reshape(-1)
    assert O.resample(lr, 2, 11025).tolist() == ((x.astype(np.int32) + (-x // 2).astype(np.int32)) / 2).astype(np.int32).tolist()


def test_oracle_reproduces_chromaprints_own_silence_vector():
    """The one known answer of libchromaprint's own test-suite that needs no audio file (tests/test_api.cpp,
    Test2SilenceFp / Test2SilenceRawFp): 130 x 1024 zero samples at 44100 Hz -> three items 627964279.  It pins,
    for the all-zero feature vector, every classifier's quantiser (which side of its thresholds 0 falls), the
    Gray code and the bit packing order, and through the item count the frame / latency arithmetic (4096-sample
    frames, hop 1365, 19 frames of latency).  The compressed form upstream prints decodes to the same items."""
    import base64
    import json
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "chromaprint_silence.json")))
    n_in = g["feeds"] * g["samples_per_feed"]
    mono = O.resample(np.zeros(n_in, dtype=np.int16), g["channels"], g["sample_rate"])
    assert len(mono) == n_in // 4 and not mono.any()
    items = O.fingerprint(mono)
    assert items.tolist() == g["raw_fingerprint"]
    assert O.simhash32(items) == g["fingerprint_hash"]
    # chromaprint's FingerprintCompressor: header (algorithm, 24-bit length), then 3-bit gaps between set bits of
    # each XOR-delta, 0 = end of item (no gap >= 7 occurs here)
    s = g["compressed_base64"]
    raw = base64.urlsafe_b64decode(s + "=" * (-len(s) % 4))
    assert raw[0] == 1 and int.from_bytes(raw[1:4], "big") == len(items)
    bits = int.from_bytes(raw[4:], "little")
    decoded, value, last = [], 0, 0
    while len(decoded) < len(items):
        gap, bits = bits & 7, bits >> 3
        assert gap != 7
        if gap == 0:
            decoded.append(value)
            value, last = 0, 0
        else:
            last += gap
            value |= 1 << (last - 1)
    for i in range(1, len(decoded)):
        decoded[i] ^= decoded[i - 1]
    assert decoded == g["raw_fingerprint"]


def test_optimised_cpu_scan_reports_the_table_walks_runs():
    """The second CPU baseline (diagonal scan, no table) finds exactly the runs the literal table walk reports."""
    rng = np.random.default_rng(21)
    seqs = [rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32) for n in (90, 140, 61, 2, 1)]
    seqs[1][20:75] = seqs[0][30:85]
    seqs[2][5:40] = seqs[0][50:85] ^ np.uint32(1 << 7)
    seqs[2][41:60] = seqs[1][100:119]
    for thr, min_len in [(10, 1), (10, 12), (0, 5), (32, 3)]:
        total, runs = O.diagonal_runs_all_pairs(seqs, thr, min_len, threads=3, capacity=100000)
        assert total == len(runs)
        got = sorted(map(tuple, runs.tolist()))
        want = []
        pair = 0
        for i in range(len(seqs)):
            for j in range(i + 1, len(seqs)):
                ents = O.longest_common_hash_match(
                    O.Comparator(hash_match_threshold=thr, min_opening_duration=0),
                    [(int(h), k) for k, h in enumerate(seqs[i])], [(int(h), k) for k, h in enumerate(seqs[j])], 0, 0)
                want += [(pair, e["src_end_idx"], e["dst_end_idx"], e["score"]) for e in ents if e["score"] >= min_len]
                pair += 1
        assert got == sorted(want)


def test_oracle_stages_reproduce_chromaprints_unit_test_vectors():
    """libchromaprint's own unit tests hold known answers for the stages of the pipeline taken one at a time: the
    Hamming window (denominator size - 1), the bin -> pitch-class map (with and without interpolation: the six-digit expectations fix base frequency, rounding
    and class origin of Chroma::PrepareNotes), the temporal FIR (which coefficient meets the oldest row), the
    Euclidean normaliser with its 0.01 threshold, and the quantiser's `<` at the thresholds.  The oracle's stage
    functions -- the same ones ora_chromaprint_fingerprint runs -- must reproduce every one of them."""
    import json
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "chromaprint_unit_vectors.json")))
    h = g["hamming_window"]
    assert np.allclose(O.hamming_window(h["size"]), h["expected"], rtol=0, atol=h["tolerance"])
    c = g["chroma"]
    k = c["constructor"]
    for case in c["cases"]:
        frame = [0.0] * c["frame_bins"]
        frame[case["bin"]] = 1.0
        got = O.chroma_features(k["min_freq"], k["max_freq"], k["frame_size"], k["sample_rate"], frame,
                                case["interpolate"])
        assert np.allclose(got, case["expected"], rtol=0, atol=c["tolerance"]), case["name"]
    for case in g["chroma_filter"]["cases"]:
        rows = [r + [0.0] * 10 for r in case["rows"]]
        got = O.chroma_filter(case["coefficients"], rows)
        assert len(got) == len(case["expected"]), case["name"]
        for g_row, e_row in zip(got, case["expected"]):
            assert np.allclose(g_row[:2], e_row, rtol=1e-6, atol=0) and not any(g_row[2:]), case["name"]
    n = g["normalize_vector"]
    for case in n["cases"]:
        got = O.normalize_vector(case["input"], n["threshold"])
        assert np.allclose(got, case["expected"], rtol=0, atol=n["tolerance"]), case["name"]
    q = g["quantizer"]
    for value, level in q["cases"]:
        assert O.quantize(value, *q["thresholds"]) == level, value
    # and the default fingerprinter's own constants go through the same map: 28..3520 Hz of a 4096-point frame at
    # 11025 Hz = bins 10..1307, A4 = 440 Hz (bin 163.5) on the class-0 / class-11 border
    frame = [0.0] * 2049
    frame[164] = 1.0
    assert O.chroma_features(28, 3520, 4096, 11025, frame)[0] == 1.0
    frame = [0.0] * 2049
    frame[163] = frame[9] = frame[1308] = 1.0
    feats = O.chroma_features(28, 3520, 4096, 11025, frame)
    assert feats[11] == 1.0 and sum(feats) == 1.0


# ---- the table-free restatement used for checks at library scale ---------------------------------------------------
def test_tablefree_pair_function_equals_the_literal_one():
    """ora_longest_common_hash_match_tablefree walks diagonals instead of filling comparator.rs:175's table; it must
    return the same entries in the same BinaryHeap array order -- on planted runs, chance runs at a low minimum
    duration (many entries: ties in score, walk order matters), runs that end on the table's last row / column, and
    with the index-0 row and column excluded (:179-180)."""
    rng = np.random.default_rng(5)
    hd = O.duration_from_secs_f32(0.3)

    def seq(hashes):
        return O.step_and_timestamp(np.repeat(np.asarray(hashes, dtype=np.uint32), 2)[: 2 * len(hashes) - 1], hd)

    base = rng.integers(0, 2 ** 32, 400, dtype=np.uint64).astype(np.uint32)
    shared = rng.integers(0, 2 ** 32, 120, dtype=np.uint64).astype(np.uint32)
    a, b = base.copy(), rng.integers(0, 2 ** 32, 380, dtype=np.uint64).astype(np.uint32)
    a[30:150] = shared
    b[200:320] = shared ^ (1 << rng.integers(0, 32, 120)).astype(np.uint32)          # one bit off: still a match
    a[-40:] = b[-40:]                                                               # a run that ends at both table edges
    b[0:50] = a[0:50]                                                               # a run through index 0
    cases = [(O.Comparator(), a, b), (O.Comparator(min_opening_duration=2 * NS), a, b),
             (O.Comparator(min_opening_duration=0), a[:90], b[:70]),
             (O.Comparator(hash_match_threshold=14, min_opening_duration=1 * NS), a, b),
             (O.Comparator(min_opening_duration=5 * NS), a, a)]
    for cmp, x, y in cases:
        want = O.longest_common_hash_match(cmp, seq(x), seq(y), hd, hd)
        got = O.longest_common_hash_match(cmp, seq(x), seq(y), hd, hd, tablefree=True)
        assert got == want and len(want) > 0
    want = O.longest_common_hash_match(O.Comparator(min_ending_duration=3 * NS), seq(a), seq(b), hd, hd, is_opening=False)
    assert O.longest_common_hash_match(O.Comparator(min_ending_duration=3 * NS), seq(a), seq(b), hd, hd, False, True) == want


def test_selected_videos_equal_the_full_call():
    eps = synth.make_library(6, 90.0, 20.0)
    hd = O.duration_from_secs_f32(0.3)
    fhs = O.analyze_batch([e.pcm[: len(e.pcm) // 2] for e in eps], 1, hd)
    cmp = O.Comparator(min_opening_duration=10 * NS)
    full = O.run_with_frame_hashes(cmp, fhs)
    hashes = [np.array([h for h, _ in f.opening], dtype=np.uint32) for f in fhs]
    ts = [np.array([t for _, t in f.opening], dtype=np.uint64) for f in fhs]
    sel = [4, 0, 5, 2]
    got = O.run_selected_videos(cmp, hashes, ts, hd, sel, threads=3)
    assert [None if r is None else (r.opening, r.ending) for r in got] == \
           [None if full[v] is None else (full[v].opening, full[v].ending) for v in sel]
    assert all(r is not None and r.opening is not None for r in got)
