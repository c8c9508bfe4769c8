"""Deterministic synthetic episode libraries for tests and bench.py (harness tool).

Layout follows SURVEY.md §8(d): mono s16 at 11025 Hz, a unique tonal body per episode, one shared
intro inside the opening search window (first 50 %, needle/src/audio/mod.rs:19) at an offset that
is not a multiple of chromaprint's 1365-sample hop, one shared outro inside the ending window
(last 25 %, mod.rs:24), episode-specific noise on top.  The sample generator is C
(csrc/synth.c, no libm => bit-reproducible); this module only lays episodes out.
"""
from __future__ import annotations

import ctypes
import os
from concurrent.futures import ThreadPoolExecutor
from dataclasses import dataclass
from typing import List

import numpy as np

RATE = 11025
EPISODE_SEED = 0x6E6565646C65
INTRO_SEED = 1
OUTRO_SEED = 2

_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libneedle_synth.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        lib = ctypes.CDLL(path)
        lib.needle_synth_episode.argtypes = [
            ctypes.c_uint64, ctypes.c_size_t, ctypes.c_uint64, ctypes.c_size_t, ctypes.c_size_t,
            ctypes.c_uint64, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p]
        lib.needle_synth_episode.restype = None
        _LIB = lib
    return _LIB


@dataclass
class Episode:
    index: int
    pcm: np.ndarray          # int16 mono, full episode
    intro_off: int           # samples
    intro_len: int
    outro_off: int
    outro_len: int

    @property
    def duration_s(self) -> float:
        return len(self.pcm) / RATE


def episode_layout(k: int, total: int, intro_s: float, outro_s: float):
    """Offsets of the shared segments in episode k (samples)."""
    intro_len = int(intro_s * RATE)
    outro_len = int(outro_s * RATE)
    half = total // 2
    # spread intro offsets over the opening window, never hop-aligned
    room = max(half - intro_len, 1)
    base = int((0.04 + 0.09 * (k % 5)) * room)
    intro_off = min(base + 137 * k, max(room - 1, 0)) if intro_len else 0
    tail_start = total - total // 4
    room_t = max(total - tail_start - outro_len, 1)
    base_t = int((0.10 + 0.15 * (k % 4)) * room_t)
    outro_off = tail_start + min(base_t + 211 * (k % 16), max(room_t - 1, 0)) if outro_len else 0
    return intro_off, intro_len, outro_off, outro_len


def make_episode(k: int, seconds: float, intro_s: float, outro_s: float = 0.0,
                 seed_base: int = EPISODE_SEED) -> Episode:
    total = int(round(seconds * RATE))
    intro_off, intro_len, outro_off, outro_len = episode_layout(k, total, intro_s, outro_s)
    scratch = np.empty(total, dtype=np.float64)
    out = np.empty(total, dtype=np.int16)
    _lib().needle_synth_episode(
        ctypes.c_uint64(seed_base ^ k), total, INTRO_SEED, intro_off, intro_len,
        OUTRO_SEED, outro_off, outro_len, scratch.ctypes.data, out.ctypes.data)
    return Episode(k, out, intro_off, intro_len, outro_off, outro_len)


def make_library(n_episodes: int, seconds: float, intro_s: float, outro_s: float = 0.0,
                 threads: int | None = None, seed_base: int = EPISODE_SEED) -> List[Episode]:
    threads = threads or min(os.cpu_count() or 1, n_episodes, 16)
    with ThreadPoolExecutor(max_workers=threads) as pool:
        return list(pool.map(lambda k: make_episode(k, seconds, intro_s, outro_s, seed_base),
                             range(n_episodes)))


def write_wav(path: str, pcm: np.ndarray, channels: int = 1, rate: int = RATE) -> None:
    """Minimal RIFF/WAVE PCM s16 writer (channels=2 duplicates the mono signal, L = R, which is
    what the reference's resampler hands chromaprint for a mono source, analyzer.rs:183-185)."""
    data = pcm if channels == 1 else np.repeat(pcm, channels)
    data = np.ascontiguousarray(data, dtype="<i2")
    nbytes = data.nbytes
    with open(path, "wb") as f:
        f.write(b"RIFF" + (36 + nbytes).to_bytes(4, "little") + b"WAVE")
        f.write(b"fmt " + (16).to_bytes(4, "little") + (1).to_bytes(2, "little")
                + channels.to_bytes(2, "little") + rate.to_bytes(4, "little")
                + (rate * channels * 2).to_bytes(4, "little") + (channels * 2).to_bytes(2, "little")
                + (16).to_bytes(2, "little"))
        f.write(b"data" + nbytes.to_bytes(4, "little"))
        f.write(data.tobytes())


# ---- libraries generated in HBM (csrc/synth_hip.hip): tests and measurements at library scale -------------------------
_HIP_LIB = None


def _hip_lib():
    global _HIP_LIB
    if _HIP_LIB is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libneedle_synth_hip.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        lib = ctypes.CDLL(path)
        lib.needle_synth_hip_library.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32,
                                                 ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint64,
                                                 ctypes.c_uint64, ctypes.c_void_p]
        lib.needle_synth_hip_library.restype = ctypes.c_int
        lib.needle_synth_hip_library_hostile.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32,
                                                         ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p,
                                                         ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p]
        lib.needle_synth_hip_library_hostile.restype = ctypes.c_int
        _HIP_LIB = lib
    return _HIP_LIB


def hostile_segments(k: int, samples: int, intro_off: int, intro_len: int):
    """Where episode k of the HOSTILE corpus has its stretch of digital silence and its sustained chord (samples within the
    search window; length 0 = none): every second episode is silent for 25 - 60 s, every third holds the chord for 25 - 60 s,
    both in the larger part of the window beside the shared intro, never overlapping it or each other."""
    before, after = intro_off, samples - (intro_off + intro_len)
    start, room = (0, before) if before >= after else (intro_off + intro_len, after)
    sil_len = int((25.0 + 35.0 * ((k * 7) % 11) / 10.0) * RATE) if k % 2 == 0 else 0
    chord_len = int((25.0 + 35.0 * ((k * 5) % 7) / 6.0) * RATE) if k % 3 == 0 else 0
    if sil_len + chord_len + 2 * RATE > room:                    # short windows (tests): scale both down
        scale = max(room - 2 * RATE, 0) / max(sil_len + chord_len, 1)
        sil_len, chord_len = int(sil_len * scale), int(chord_len * scale)
    slack = room - sil_len - chord_len - RATE
    sil_off = start + (slack * ((k * 13) % 17)) // 17
    chord_off = sil_off + sil_len + RATE
    return sil_off, sil_len, chord_off, chord_len


class DeviceLibrary:
    """`n` episodes of `samples` mono s16 values each, generated on the device: a unique tonal body per episode and
    one shared intro of `intro_s` seconds at the offsets episode_layout() gives (the whole stream is meant to be the
    opening search window: use with opening_search_percentage = 1.0).  Not the host generator's samples -- its own
    (csrc/synth_hip.hip); a checker reads the PCM it wants back with episode()."""

    def __init__(self, n: int, samples: int, intro_s: float, first_episode: int = 0, seed_base: int = EPISODE_SEED,
                 hostile: bool = False):
        """hostile: the round-6 corpus of csrc/synth_hip.hip -- broadband speech-like bodies, noise 20 dB under the programme,
        stretches of digital silence and of one sustained chord (hostile_segments) -- instead of note sequences."""
        from . import capi
        self.n, self.samples = n, samples
        self.stride = (samples + 7) & ~7                       # every episode 16-byte aligned
        self.intro_len = int(intro_s * RATE)
        # the window generated is the first half of an episode twice as long: same offsets as make_episode(k, 2 * ...)
        self.intro_off = np.array([episode_layout(first_episode + k, 2 * samples, intro_s, 0.0)[0] for k in range(n)],
                                  dtype=np.uint32)
        assert int(self.intro_off.max()) + self.intro_len <= samples
        self._buf = capi.DeviceBuffer(self.stride * 2 * n)
        off = capi.DeviceBuffer(4 * n)
        capi.check(capi.lib().needle_hip_memcpy_h2d(off.ptr, self.intro_off.ctypes.data, self.intro_off.nbytes))
        self.hostile = hostile
        if hostile:
            self.segments = np.array([hostile_segments(first_episode + k, samples, int(self.intro_off[k]), self.intro_len)
                                      for k in range(n)], dtype=np.uint32)
            seg = capi.DeviceBuffer(16 * n)
            capi.check(capi.lib().needle_hip_memcpy_h2d(seg.ptr, self.segments.ctypes.data, self.segments.nbytes))
            rc = _hip_lib().needle_synth_hip_library_hostile(self._buf.ptr, self.stride, n, first_episode, samples, off.ptr,
                                                             self.intro_len, seg.ptr, seed_base, INTRO_SEED, capi.stream_ptr())
        else:
            rc = _hip_lib().needle_synth_hip_library(self._buf.ptr, self.stride, n, first_episode, samples, off.ptr,
                                                     self.intro_len, seed_base, INTRO_SEED, capi.stream_ptr())
        if rc != 0:
            raise RuntimeError(f"needle_synth_hip_library: HIP error {rc}")
        capi.synchronize()

    def pointers(self):
        return [self._buf.ptr + 2 * self.stride * k for k in range(self.n)]

    def episode(self, k: int) -> np.ndarray:
        """Episode k's PCM, copied to the host."""
        from . import capi
        out = np.zeros(self.samples, dtype=np.int16)
        capi.check(capi.lib().needle_hip_memcpy_d2h(out.ctypes.data, self._buf.ptr + 2 * self.stride * k, out.nbytes))
        return out

    def free(self) -> None:
        self._buf = None
