"""ctypes binding of liboracle.so — ORACLE, test infrastructure only.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module, and
only as the checker / timed CPU baseline.  The product (needle_amd/, libneedle_capi.so) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

NS = 1_000_000_000


class HashTs(C.Structure):
    _fields_ = [("hash", C.c_uint32), ("ts", C.c_uint64)]


class CFrameHashes(C.Structure):
    _fields_ = [("opening", C.POINTER(HashTs)), ("n_opening", C.c_size_t),
                ("ending", C.POINTER(HashTs)), ("n_ending", C.c_size_t),
                ("hash_duration", C.c_uint64), ("md5", C.c_char * 33)]


class CEntry(C.Structure):
    _fields_ = [("score", C.c_size_t),
                ("src_start", C.c_uint64), ("src_end", C.c_uint64),
                ("dst_start", C.c_uint64), ("dst_end", C.c_uint64),
                ("src_match_hash", C.c_uint32), ("dst_match_hash", C.c_uint32),
                ("is_src_opening", C.c_bool), ("is_src_ending", C.c_bool),
                ("is_dst_opening", C.c_bool), ("is_dst_ending", C.c_bool),
                ("src_hash_duration", C.c_uint64), ("dst_hash_duration", C.c_uint64),
                ("src_end_idx", C.c_uint32), ("dst_end_idx", C.c_uint32)]


class CComparator(C.Structure):
    _fields_ = [("include_endings", C.c_bool), ("hash_match_threshold", C.c_uint32),
                ("min_opening_duration", C.c_uint64), ("min_ending_duration", C.c_uint64),
                ("time_padding", C.c_uint64)]


class CSearchResult(C.Structure):
    _fields_ = [("has_result", C.c_bool), ("has_opening", C.c_bool), ("has_ending", C.c_bool),
                ("opening_start", C.c_uint64), ("opening_end", C.c_uint64),
                ("ending_start", C.c_uint64), ("ending_end", C.c_uint64)]


def build() -> str:
    subprocess.run(["make", "-s", "-C", _HERE], check=True)
    return os.path.join(_HERE, "liboracle.so")


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = os.environ.get("NEEDLE_ORACLE_LIB") or os.path.join(_HERE, "liboracle.so")   # another build of the checker (ASan)
    if not os.path.exists(path):
        build()
    L = C.CDLL(path)
    L.ora_chromaprint_delay_ms.restype = C.c_int
    L.ora_chromaprint_item_duration_ms.restype = C.c_int
    L.ora_chromaprint_num_frames.argtypes = [C.c_size_t]
    L.ora_chromaprint_num_frames.restype = C.c_size_t
    L.ora_chromaprint_num_items.argtypes = [C.c_size_t]
    L.ora_chromaprint_num_items.restype = C.c_size_t
    L.ora_chromaprint_fingerprint.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t,
                                              C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]
    L.ora_chromaprint_fingerprint.restype = C.c_size_t
    L.ora_prepare_hamming_window.argtypes = [C.c_void_p, C.c_int, C.c_double]
    L.ora_prepare_hamming_window.restype = None
    L.ora_chroma_prepare_notes.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p]
    L.ora_chroma_prepare_notes.restype = None
    L.ora_chroma_consume.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.ora_chroma_consume.restype = None
    L.ora_chroma_filter_row.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.ora_chroma_filter_row.restype = None
    L.ora_normalize_vector.argtypes = [C.c_void_p, C.c_int, C.c_double]
    L.ora_normalize_vector.restype = None
    L.ora_quantize.argtypes = [C.c_double, C.c_double, C.c_double, C.c_double]
    L.ora_quantize.restype = C.c_int
    L.ora_simhash32.argtypes = [C.c_void_p, C.c_size_t]
    L.ora_simhash32.restype = C.c_uint32
    L.ora_duration_from_secs_f32.argtypes = [C.c_float]
    L.ora_duration_from_secs_f32.restype = C.c_uint64
    L.ora_duration_from_secs_f64.argtypes = [C.c_double]
    L.ora_duration_from_secs_f64.restype = C.c_uint64
    L.ora_duration_as_secs_f32.argtypes = [C.c_uint64]
    L.ora_duration_as_secs_f32.restype = C.c_float
    L.ora_duration_mul_f32.argtypes = [C.c_uint64, C.c_float]
    L.ora_duration_mul_f32.restype = C.c_uint64
    L.ora_step_and_timestamp.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64, C.c_int, C.c_int, C.c_uint64,
                                         C.c_int, C.POINTER(HashTs), C.c_size_t, C.POINTER(C.c_int)]
    L.ora_step_and_timestamp.restype = C.c_size_t
    L.ora_md5_hex.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p]
    L.ora_header_md5.argtypes = [C.c_char_p, C.c_char_p]
    L.ora_header_md5.restype = C.c_int
    L.ora_frame_hashes_write.argtypes = [C.c_char_p, C.POINTER(CFrameHashes)]
    L.ora_frame_hashes_write.restype = C.c_int
    L.ora_frame_hashes_read.argtypes = [C.c_char_p, C.POINTER(CFrameHashes)]
    L.ora_frame_hashes_read.restype = C.c_int
    L.ora_frame_hashes_free.argtypes = [C.POINTER(CFrameHashes)]
    L.ora_comparator_default.argtypes = [C.POINTER(CComparator)]
    L.ora_longest_common_hash_match.argtypes = [C.POINTER(CComparator), C.POINTER(HashTs), C.c_size_t,
                                                C.POINTER(HashTs), C.c_size_t, C.c_uint64, C.c_uint64,
                                                C.c_bool, C.POINTER(C.c_size_t)]
    L.ora_longest_common_hash_match.restype = C.POINTER(CEntry)
    L.ora_run_with_frame_hashes.argtypes = [C.POINTER(CComparator), C.POINTER(CFrameHashes), C.c_size_t,
                                            C.POINTER(CSearchResult)]
    L.ora_run_with_frame_hashes.restype = C.c_int
    L.ora_longest_common_hash_match_tablefree.argtypes = L.ora_longest_common_hash_match.argtypes
    L.ora_longest_common_hash_match_tablefree.restype = C.POINTER(CEntry)
    L.ora_run_selected_videos.argtypes = [C.POINTER(CComparator), C.POINTER(CFrameHashes), C.c_size_t,
                                          C.POINTER(C.c_size_t), C.c_size_t, C.POINTER(CSearchResult)]
    L.ora_run_selected_videos.restype = C.c_int
    L.ora_set_threads.argtypes = [C.c_int]
    L.ora_get_threads.restype = C.c_int
    L.ora_analyze_batch.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_int, C.c_size_t,
                                    C.c_uint64, C.POINTER(CFrameHashes)]
    L.ora_analyze_batch.restype = C.c_int
    L.ora_skip_file_json.argtypes = [C.POINTER(CSearchResult), C.c_char_p, C.c_char_p, C.c_size_t]
    L.ora_skip_file_json.restype = C.c_size_t
    L.ora_format_time.argtypes = [C.c_uint64, C.c_char_p]
    L.ora_resample_out_len.argtypes = [C.c_size_t, C.c_int]
    L.ora_resample_out_len.restype = C.c_size_t
    L.ora_resample.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
    L.ora_resample.restype = C.c_size_t
    L.free = C.CDLL(None).free
    L.free.argtypes = [C.c_void_p]
    _LIB = L
    return L


# ---- pythonic views ---------------------------------------------------------------------------------
@dataclass
class FrameHashes:
    opening: List[Tuple[int, int]]            # (hash, ts_ns)
    ending: List[Tuple[int, int]]
    hash_duration: int                        # ns
    md5: str = ""

    def to_c(self) -> CFrameHashes:
        c = CFrameHashes()
        self._o = (HashTs * max(len(self.opening), 1))(*[HashTs(h, t) for h, t in self.opening])
        self._e = (HashTs * max(len(self.ending), 1))(*[HashTs(h, t) for h, t in self.ending])
        c.opening = C.cast(self._o, C.POINTER(HashTs))
        c.n_opening = len(self.opening)
        c.ending = C.cast(self._e, C.POINTER(HashTs))
        c.n_ending = len(self.ending)
        c.hash_duration = self.hash_duration
        c.md5 = self.md5.encode()
        return c

    @staticmethod
    def from_c(c: CFrameHashes) -> "FrameHashes":
        return FrameHashes([(c.opening[i].hash, c.opening[i].ts) for i in range(c.n_opening)],
                           [(c.ending[i].hash, c.ending[i].ts) for i in range(c.n_ending)],
                           c.hash_duration, c.md5.decode())


@dataclass
class Comparator:
    include_endings: bool = False
    hash_match_threshold: int = 10
    min_opening_duration: int = 20 * NS
    min_ending_duration: int = 20 * NS
    time_padding: int = 0

    def to_c(self) -> CComparator:
        return CComparator(self.include_endings, self.hash_match_threshold, self.min_opening_duration,
                           self.min_ending_duration, self.time_padding)


@dataclass
class SearchResult:
    opening: Optional[Tuple[int, int]]
    ending: Optional[Tuple[int, int]]


def delay_ms() -> int:
    return lib().ora_chromaprint_delay_ms()


def item_duration_ms() -> int:
    return lib().ora_chromaprint_item_duration_ms()


def num_frames(samples: int) -> int:
    return lib().ora_chromaprint_num_frames(samples)


def num_items(samples: int) -> int:
    return lib().ora_chromaprint_num_items(samples)


def fingerprint(pcm: np.ndarray, channels: int = 1, debug: bool = False):
    """Raw chromaprint items for interleaved s16 PCM.  debug=True also returns (chroma, features, min_margin)."""
    pcm = np.ascontiguousarray(pcm, dtype=np.int16)
    samples = pcm.size // channels
    n = num_items(samples)
    out = np.zeros(max(n, 1), dtype=np.uint32)
    if not debug:
        got = lib().ora_chromaprint_fingerprint(pcm.ctypes.data, pcm.size, channels, out.ctypes.data, n,
                                                None, None, None)
        assert got == n
        return out[:n]
    frames = num_frames(samples)
    chroma = np.zeros((max(frames, 1), 12), dtype=np.float64)
    feats = np.zeros((max(frames - 4, 1), 12), dtype=np.float64)
    margin = C.c_double(0.0)
    got = lib().ora_chromaprint_fingerprint(pcm.ctypes.data, pcm.size, channels, out.ctypes.data, n,
                                            chroma.ctypes.data, feats.ctypes.data, C.byref(margin))
    assert got == n
    return out[:n], chroma[:frames], feats[:max(frames - 4, 0)], margin.value


# ---- the pipeline's stages one by one (what libchromaprint's unit tests exercise) --------------------------------
def hamming_window(size: int, scale: float = 1.0) -> List[float]:
    w = (C.c_double * size)()
    lib().ora_prepare_hamming_window(w, size, C.c_double(scale))
    return list(w)


def chroma_features(min_freq: int, max_freq: int, frame_size: int, sample_rate: int, frame: Sequence[float],
                    interpolate: bool = False) -> List[float]:
    """Chroma(min_freq, max_freq, frame_size, sample_rate).Consume(frame) -> 12 pitch-class energies."""
    L = lib()
    half = frame_size // 2
    notes = (C.c_byte * (half + 1))()
    frac = (C.c_double * (half + 1))()
    lo, hi = C.c_int(0), C.c_int(0)
    L.ora_chroma_prepare_notes(min_freq, max_freq, frame_size, sample_rate, notes, frac, C.byref(lo), C.byref(hi))
    f = (C.c_double * max(len(frame), half + 1))(*frame)
    out = (C.c_double * 12)()
    L.ora_chroma_consume(notes, frac, lo.value, hi.value, int(interpolate), f, out)
    return list(out)


def chroma_filter(coefficients: Sequence[float], rows: Sequence[Sequence[float]]) -> List[List[float]]:
    """ChromaFilter(coefficients): one output row per input row once len(coefficients) rows are buffered."""
    L = lib()
    taps, bands = len(coefficients), len(rows[0])
    coef = (C.c_double * taps)(*coefficients)
    bufs = [(C.c_double * bands)(*r) for r in rows]
    outs = []
    for i in range(len(rows) - taps + 1):
        ptrs = (C.POINTER(C.c_double) * taps)(*[C.cast(bufs[i + j], C.POINTER(C.c_double)) for j in range(taps)])
        out = (C.c_double * bands)()
        L.ora_chroma_filter_row(ptrs, coef, taps, bands, out)
        outs.append(list(out))
    return outs


def normalize_vector(v: Sequence[float], threshold: float = 0.01) -> List[float]:
    a = (C.c_double * len(v))(*v)
    lib().ora_normalize_vector(a, len(v), C.c_double(threshold))
    return list(a)


def quantize(value: float, t0: float, t1: float, t2: float) -> int:
    return lib().ora_quantize(C.c_double(value), C.c_double(t0), C.c_double(t1), C.c_double(t2))


def simhash32(data: Sequence[int]) -> int:
    a = np.ascontiguousarray(data, dtype=np.uint32)
    return lib().ora_simhash32(a.ctypes.data, a.size)


def duration_from_secs_f32(s: float) -> int:
    return lib().ora_duration_from_secs_f32(s)


def duration_from_secs_f64(s: float) -> int:
    return lib().ora_duration_from_secs_f64(s)


def duration_as_secs_f32(ns: int) -> float:
    return lib().ora_duration_as_secs_f32(ns)


def duration_mul_f32(ns: int, rhs: float) -> int:
    return lib().ora_duration_mul_f32(ns, rhs)


def step_and_timestamp(raw: np.ndarray, hash_duration_ns: int, seek_to_ns: Optional[int] = None):
    raw = np.ascontiguousarray(raw, dtype=np.uint32)
    out = (HashTs * max(raw.size, 1))()
    err = C.c_int(0)
    k = lib().ora_step_and_timestamp(raw.ctypes.data, raw.size, hash_duration_ns, delay_ms(), item_duration_ms(),
                                     seek_to_ns or 0, 1 if seek_to_ns is not None else 0, out, raw.size,
                                     C.byref(err))
    if err.value:
        raise ValueError("step_by == 0 (the reference panics)")
    return [(out[i].hash, out[i].ts) for i in range(k)]


def md5_hex(data: bytes) -> str:
    buf = C.create_string_buffer(33)
    lib().ora_md5_hex(data, len(data), buf)
    return buf.value.decode()


def header_md5(path: str) -> Optional[str]:
    buf = C.create_string_buffer(33)
    return None if lib().ora_header_md5(path.encode(), buf) else buf.value.decode()


def frame_hashes_write(path: str, fh: FrameHashes) -> int:
    c = fh.to_c()
    return lib().ora_frame_hashes_write(path.encode(), C.byref(c))


def frame_hashes_read(path: str):
    c = CFrameHashes()
    rc = lib().ora_frame_hashes_read(path.encode(), C.byref(c))
    if rc:
        return rc, None
    fh = FrameHashes.from_c(c)
    lib().ora_frame_hashes_free(C.byref(c))
    return 0, fh


def longest_common_hash_match(cmp: Comparator, src, dst, src_hd: int, dst_hd: int, is_opening: bool = True,
                              tablefree: bool = False):
    """Entries in BinaryHeap backing-array order, as dicts.  tablefree: the diagonal-walk form of the same function
    (ora_longest_common_hash_match_tablefree), used for checks at library scale."""
    cc = cmp.to_c()
    s = (HashTs * max(len(src), 1))(*[HashTs(h, t) for h, t in src])
    d = (HashTs * max(len(dst), 1))(*[HashTs(h, t) for h, t in dst])
    n = C.c_size_t(0)
    fn = lib().ora_longest_common_hash_match_tablefree if tablefree else lib().ora_longest_common_hash_match
    p = fn(C.byref(cc), s, len(src), d, len(dst), src_hd, dst_hd, is_opening, C.byref(n))
    out = []
    for i in range(n.value):
        e = p[i]
        out.append({f: getattr(e, f) for f, _ in CEntry._fields_})
    if n.value:
        lib().free(C.cast(p, C.c_void_p))
    return out


def run_with_frame_hashes(cmp: Comparator, fhs: Sequence[FrameHashes], threads: int = 1):
    """Per-video Optional[SearchResult] (None = the reference skips the video, comparator.rs:608-617)."""
    cc = cmp.to_c()
    arr = (CFrameHashes * max(len(fhs), 1))(*[f.to_c() for f in fhs])
    res = (CSearchResult * max(len(fhs), 1))()
    lib().ora_set_threads(threads)
    rc = lib().ora_run_with_frame_hashes(C.byref(cc), arr, len(fhs), res)
    lib().ora_set_threads(1)
    if rc == 1:
        raise RuntimeError("FrameHashDataNoEnding")
    if rc == 2:
        raise OverflowError("overflow when subtracting durations")
    out = []
    for i in range(len(fhs)):
        r = res[i]
        if not r.has_result:
            out.append(None)
        else:
            out.append(SearchResult((r.opening_start, r.opening_end) if r.has_opening else None,
                                    (r.ending_start, r.ending_end) if r.has_ending else None))
    return out


HASH_TS_DTYPE = np.dtype([("hash", "<u4"), ("pad", "<u4"), ("ts", "<u8")])   # struct HashTs as the C compiler lays it out


def run_selected_videos(cmp: Comparator, hashes: Sequence[np.ndarray], timestamps: Sequence[np.ndarray],
                        hash_duration_ns: int, videos: Sequence[int], threads: int = 1):
    """comparator.rs:524-629 for `videos` of a library given as numpy arrays (opening window only): what
    run_with_frame_hashes returns for those videos, through the table-free pair function
    (ora_run_selected_videos) -- the form that stays feasible at 2000 episodes."""
    assert C.sizeof(HashTs) == HASH_TS_DTYPE.itemsize
    keep, arr = [], (CFrameHashes * max(len(hashes), 1))()
    empty = (HashTs * 1)()
    for v, (h, t) in enumerate(zip(hashes, timestamps)):
        buf = np.zeros(len(h), dtype=HASH_TS_DTYPE)
        buf["hash"], buf["ts"] = h, t
        keep.append(buf)
        arr[v].opening = C.cast(buf.ctypes.data, C.POINTER(HashTs))
        arr[v].n_opening = len(h)
        arr[v].ending = C.cast(empty, C.POINTER(HashTs))
        arr[v].n_ending = 0
        arr[v].hash_duration = hash_duration_ns
    cc = cmp.to_c()
    sel = (C.c_size_t * max(len(videos), 1))(*[int(v) for v in videos])
    res = (CSearchResult * max(len(videos), 1))()
    lib().ora_set_threads(threads)
    rc = lib().ora_run_selected_videos(C.byref(cc), arr, len(hashes), sel, len(videos), res)
    lib().ora_set_threads(1)
    if rc:
        raise RuntimeError(f"ora_run_selected_videos failed: {rc}")
    return [None if not r.has_result else
            SearchResult((r.opening_start, r.opening_end) if r.has_opening else None,
                         (r.ending_start, r.ending_end) if r.has_ending else None) for r in res[:len(videos)]]


def diagonal_runs_all_pairs(seqs: Sequence[np.ndarray], threshold: int, min_len: int, threads: int = 1,
                            capacity: int = 0):
    """The optimised CPU variant of the pair scan (no table; ora_needle.h).  Returns (total, runs ndarray [k, 4] of
    (pair, src_end, dst_end, len))."""
    seqs = [np.ascontiguousarray(q, dtype=np.uint32) for q in seqs]
    ptrs = (C.c_void_p * len(seqs))(*[q.ctypes.data for q in seqs])
    lens = (C.c_size_t * len(seqs))(*[q.size for q in seqs])
    out = np.zeros((max(capacity, 1), 4), dtype=np.uint32)
    L = lib()
    L.ora_diagonal_runs_all_pairs.restype = C.c_size_t
    L.ora_diagonal_runs_all_pairs.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_void_p,
                                              C.c_size_t]
    L.ora_set_threads(threads)
    total = L.ora_diagonal_runs_all_pairs(ptrs, lens, len(seqs), threshold, min_len,
                                          out.ctypes.data if capacity else None, capacity)
    L.ora_set_threads(1)
    return int(total), out[: min(total, capacity)]


def analyze_batch(pcms: Sequence[np.ndarray], channels: int, hash_duration_ns: int, threads: int = 1):
    """Opening-window analyze of already-cropped PCM streams -> list[FrameHashes] (md5 empty)."""
    pcms = [np.ascontiguousarray(p, dtype=np.int16) for p in pcms]
    ptrs = (C.c_void_p * len(pcms))(*[p.ctypes.data for p in pcms])
    lens = (C.c_size_t * len(pcms))(*[p.size for p in pcms])
    out = (CFrameHashes * len(pcms))()
    lib().ora_set_threads(threads)
    rc = lib().ora_analyze_batch(ptrs, lens, channels, len(pcms), hash_duration_ns, out)
    lib().ora_set_threads(1)
    if rc:
        raise ValueError("step_by == 0 (the reference panics)")
    res = []
    for i in range(len(pcms)):
        res.append(FrameHashes.from_c(out[i]))
        lib().ora_frame_hashes_free(C.byref(out[i]))
    return res


def resample(pcm: np.ndarray, channels: int, rate: int) -> np.ndarray:
    """ora_resample: interleaved s16 at `rate` -> mono s16 at 11025 Hz (this project's own front-end spec)."""
    a = np.ascontiguousarray(pcm, dtype=np.int16)
    n = lib().ora_resample_out_len(a.size // channels, rate)
    out = np.zeros(max(n, 1), dtype=np.int16)
    got = lib().ora_resample(a.ctypes.data, a.size, channels, rate, out.ctypes.data, n)
    assert got == n
    return out[:n]


def skip_file_json(r: SearchResult, md5: str) -> str:
    c = CSearchResult(True, r.opening is not None, r.ending is not None,
                      *(r.opening or (0, 0)), *(r.ending or (0, 0)))
    buf = C.create_string_buffer(512)
    n = lib().ora_skip_file_json(C.byref(c), md5.encode(), buf, 512)
    return buf.value.decode() if n else ""


def format_time(ns: int) -> str:
    buf = C.create_string_buffer(32)
    lib().ora_format_time(ns, buf)
    return buf.value.decode()
