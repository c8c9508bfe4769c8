"""The per-thread code of the HIP kernels (needle_amd/csrc/fp_core.h, the loop body of search.hip), compiled with
g++ and stepped serially with the kernel's barrier structure, against the oracle.  Catches indexing mistakes in
the CPU suite; the GPU parity tests then only have to confirm the hardware executes the same program."""
import ctypes as C
import os

import numpy as np
import pytest

from needle_amd import synth
from oracle import oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))


class EmuRun(C.Structure):
    _fields_ = [("src_end", C.c_uint32), ("dst_end", C.c_uint32), ("len", C.c_uint32)]


@pytest.fixture(scope="module")
def emu():
    L = C.CDLL(os.path.join(HERE, "cpu_emu", "libemu.so"))
    L.emu_stft_chroma_pair.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.emu_classify.argtypes = [C.c_void_p]
    L.emu_classify.restype = C.c_uint32
    L.emu_hamming_runs.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t]
    L.emu_hamming_runs.restype = C.c_size_t
    return L


def test_stft_pair_schedule_matches_oracle_chroma(emu):
    e = synth.make_episode(0, 40.0, 10.0)
    pcm = e.pcm[: 20 * 11025]
    _, chroma, _, _ = O.fingerprint(pcm, debug=True)
    a, b = np.zeros(12), np.zeros(12)
    for f in (0, 1, 50, len(chroma) - 2):
        fa = np.ascontiguousarray(pcm[f * 1365: f * 1365 + 4096])
        fb = np.ascontiguousarray(pcm[(f + 1) * 1365: (f + 1) * 1365 + 4096])
        emu.emu_stft_chroma_pair(fa.ctypes.data, fb.ctypes.data, 1, a.ctypes.data, b.ctypes.data)
        assert np.max(np.abs(a - chroma[f]) / chroma[f]) < 1e-12
        assert np.max(np.abs(b - chroma[f + 1]) / chroma[f + 1]) < 1e-12
    # odd frame count: frame B absent; stereo L = R
    fa = np.ascontiguousarray(pcm[:4096])
    emu.emu_stft_chroma_pair(fa.ctypes.data, None, 1, a.ctypes.data, None)
    assert np.max(np.abs(a - chroma[0]) / chroma[0]) < 1e-12
    st = np.ascontiguousarray(np.repeat(fa, 2))
    emu.emu_stft_chroma_pair(st.ctypes.data, None, 2, a.ctypes.data, None)
    assert np.max(np.abs(a - chroma[0]) / chroma[0]) < 1e-12


def test_power_layout_of_the_stft_kernel_is_consistent(emu):
    """Structure of the round-2 schedule that no numerical test would localise: wave-local power slots, fold lanes that
    read every bin exactly once, the lane-group permutation that keeps a bin's partner in its wave."""
    emu.emu_power_layout_check.restype = C.c_int
    assert emu.emu_power_layout_check() == 0


def test_classifier_unrolled_regions_match_oracle_items(emu):
    e = synth.make_episode(1, 40.0, 10.0)
    items, _, feats, margin = O.fingerprint(e.pcm[: 20 * 11025], debug=True)
    assert margin > 1e-9
    for x in range(len(items)):
        w = np.ascontiguousarray(feats[x: x + 16])
        assert emu.emu_classify(w.ctypes.data) == items[x]


def test_diagonal_scan_loop_matches_oracle_dp(emu):
    rng = np.random.default_rng(3)
    for n, m, thr, min_len in [(50, 70, 11, 1), (64, 64, 32, 1), (2, 9, 10, 1), (120, 90, 9, 5)]:
        s = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
        t = rng.integers(0, 2 ** 32, m, dtype=np.uint64).astype(np.uint32)
        if n > 40:
            t[7:37] = s[11:41]
        buf = (EmuRun * 8192)()
        k = emu.emu_hamming_runs(s.ctypes.data, n, t.ctypes.data, m, thr, min_len, buf, 8192)
        got = sorted((buf[i].src_end, buf[i].dst_end, buf[i].len) for i in range(k))
        ents = O.longest_common_hash_match(O.Comparator(hash_match_threshold=thr, min_opening_duration=0),
                                           [(int(h), i) for i, h in enumerate(s)], [(int(h), i) for i, h in enumerate(t)], 0, 0)
        want = sorted((e["src_end_idx"], e["dst_end_idx"], e["score"]) for e in ents if e["score"] >= min_len)
        assert got == want


def test_f32_first_pass_schedule_chroma_and_energy(emu):
    """stft_chroma32_kernel's per-thread code stepped on the CPU (same schedule, f32 arithmetic, table window and
    twiddles): its chroma is the oracle's within f32 round-off, and the four energy partials the fourth wave folds
    add up to the sum of squares of the two windowed frames of the pair (times the kernel's 1/2 input scale, squared)."""
    emu.emu_stft_chroma_pair_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    e = synth.make_episode(2, 40.0, 10.0)
    pcm = e.pcm[: 20 * 11025]
    _, chroma, _, _ = O.fingerprint(pcm, debug=True)
    window = 0.5 * (0.54 - 0.46 * np.cos(2.0 * np.pi * np.arange(4096) / 4095)) / 32767.0
    a, b = np.zeros(12), np.zeros(12)
    ea, eb = np.zeros(4, dtype=np.float32), np.zeros(4, dtype=np.float32)
    for f in (0, 1, 77, len(chroma) - 2):
        fa = np.ascontiguousarray(pcm[f * 1365: f * 1365 + 4096])
        fb = np.ascontiguousarray(pcm[(f + 1) * 1365: (f + 1) * 1365 + 4096])
        emu.emu_stft_chroma_pair_f32(fa.ctypes.data, fb.ctypes.data, 1, a.ctypes.data, b.ctypes.data, ea.ctypes.data, eb.ctypes.data)
        assert a[0] >= 0, f"layout check failed with code {a[0]}"
        assert np.max(np.abs(a - chroma[f])) / chroma[f].max() < 2e-6
        assert np.max(np.abs(b - chroma[f + 1])) / chroma[f + 1].max() < 2e-6
        pair = float(((fa * window) ** 2).sum()) + float(((fb * window) ** 2).sum())   # both frames carry the PAIR's energy
        assert abs(float(ea.sum()) / pair - 1.0) < 1e-5
        assert abs(float(eb.sum()) / pair - 1.0) < 1e-5
    # odd frame count: frame B absent -> its energy is exactly zero
    emu.emu_stft_chroma_pair_f32(fa.ctypes.data, None, 1, a.ctypes.data, None, ea.ctypes.data, eb.ctypes.data)
    assert float(eb.sum()) == 0.0 and np.max(np.abs(a - chroma[len(chroma) - 2])) / chroma[len(chroma) - 2].max() < 2e-6
