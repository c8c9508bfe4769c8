"""ctypes binding of libneedle_capi.so (include/needle.h + include/needle_hip.h).

The classes mirror needle::audio::{Analyzer, Comparator, FrameHashes} (needle/src/audio/*.rs) over the
C ABI, so tests read like the reference's own usage: build with paths, chain with_* setters, run.
All compute happens in the shared library on the GPU; this module holds no arithmetic and no CPU
fallback — if the library or a HIP device is missing the calls raise.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

NS = 1_000_000_000
# NEEDLE_CAPI_LIB: another build of the same library (A/B timing of kernel variants)
LIB_PATH = os.environ.get("NEEDLE_CAPI_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libneedle_capi.so")

ERROR_NAMES = ["Ok", "InvalidUtf8String", "NullArgument", "InvalidArgument", "FrameHashDataNotFound",
               "FrameHashDataInvalidVersion", "InvalidFrameHashData", "ComparatorMinimumPaths",
               "AnalyzerInvalidHashPeriod", "AnalyzerInvalidHashDuration", "IOError", "Unknown"]

# audio/mod.rs:14-45
DEFAULT_HASH_MATCH_THRESHOLD = 10
DEFAULT_OPENING_SEARCH_PERCENTAGE = 0.50
DEFAULT_ENDING_SEARCH_PERCENTAGE = 0.25
DEFAULT_MIN_OPENING_DURATION = 20
DEFAULT_MIN_ENDING_DURATION = 20
DEFAULT_HASH_DURATION = 0.3


class NeedleError(RuntimeError):
    def __init__(self, code: int, detail: str = ""):
        self.code = code
        self.name = ERROR_NAMES[code] if 0 <= code < len(ERROR_NAMES) else str(code)
        super().__init__(f"NeedleError_{self.name}: {detail}")


class Seq(C.Structure):
    _fields_ = [("offset", C.c_uint32), ("len", C.c_uint32)]


class Problem(C.Structure):
    _fields_ = [("src_seq", C.c_uint32), ("dst_seq", C.c_uint32), ("min_len", C.c_uint32), ("tag", C.c_uint32)]


class Run(C.Structure):
    _fields_ = [("problem", C.c_uint32), ("src_end", C.c_uint32), ("dst_end", C.c_uint32), ("len", C.c_uint32),
                ("src_match_hash", C.c_uint32), ("dst_match_hash", C.c_uint32)]


class CSearchResult(C.Structure):
    _fields_ = [("has_result", C.c_bool), ("has_opening", C.c_bool), ("has_ending", C.c_bool),
                ("opening_start_ns", C.c_uint64), ("opening_end_ns", C.c_uint64),
                ("ending_start_ns", C.c_uint64), ("ending_end_ns", C.c_uint64)]


RUN_DTYPE = np.dtype([("problem", "<u4"), ("src_end", "<u4"), ("dst_end", "<u4"), ("len", "<u4"),
                      ("src_match_hash", "<u4"), ("dst_match_hash", "<u4")])
RUN_WORDS = 6

# Every symbol include/needle.h and include/needle_hip.h declare (tests check the library exports all).
NEEDLE_H_SYMBOLS = [
    "needle_error_to_str", "needle_util_find_video_files", "needle_util_video_files_free",
    "needle_audio_analyzer_new_default", "needle_audio_analyzer_new", "needle_audio_analyzer_get_frame_hashes",
    "needle_audio_analyzer_free", "needle_audio_analyzer_print_paths", "needle_audio_analyzer_run",
    "needle_audio_comparator_new_default", "needle_audio_comparator_new", "needle_audio_comparator_free",
    "needle_audio_comparator_run"]
NEEDLE_HIP_H_SYMBOLS = [
    "needle_hip_device_count", "needle_hip_set_device", "needle_hip_synchronize", "needle_hip_stream",
    "needle_hip_device_pci_bus_id", "needle_hip_fingerprint_cert_stats", "needle_hip_scan_issued_evaluations",
    "needle_hip_last_error_message",
    "needle_hip_version", "needle_hip_malloc", "needle_hip_free", "needle_hip_memcpy_h2d", "needle_hip_memcpy_d2h",
    "needle_hip_host_free", "needle_hip_last_kernel_ms", "needle_hip_set_kernel_timing", "needle_hip_fingerprint_sample_rate",
    "needle_hip_fingerprint_delay_ms", "needle_hip_fingerprint_item_duration_ms", "needle_hip_fingerprint_num_items",
    "needle_hip_fingerprint_num_kept", "needle_hip_fingerprint_host", "needle_hip_fingerprint_device",
    "needle_hip_fingerprint_debug", "needle_hip_resample_out_len", "needle_hip_resample_host",
    "needle_hip_hamming_runs_device", "needle_hip_hamming_runs_host",
    "needle_hip_frame_hashes_new", "needle_hip_frame_hashes_free", "needle_hip_frame_hashes_len",
    "needle_hip_frame_hashes_copy", "needle_hip_frame_hashes_hash_duration_ns", "needle_hip_frame_hashes_md5",
    "needle_hip_frame_hashes_read", "needle_hip_frame_hashes_write", "needle_hip_header_md5",
    "needle_hip_analyzer_run_pcm", "needle_hip_comparator_run_with_frame_hashes", "needle_hip_library_new",
    "needle_hip_library_free", "needle_hip_library_include_endings", "needle_hip_library_rows_per_video",
    "needle_hip_library_set_pcm", "needle_hip_library_set_pcm_device", "needle_hip_library_rank_videos",
    "needle_hip_library_analyze",
    "needle_hip_library_hash_arena", "needle_hip_library_use_hash_arena", "needle_hip_library_num_pairs", "needle_hip_library_search",
    "needle_hip_library_fetch_runs_begin", "needle_hip_library_fetch_runs_end",
    "needle_hip_library_finalize", "needle_hip_library_frame_hashes",
    "needle_hip_comm_create_id", "needle_hip_comm_init", "needle_hip_comm_finalize", "needle_hip_comm_rank",
    "needle_hip_comm_world_size", "needle_hip_comm_backend", "needle_hip_comm_barrier",
    "needle_hip_comm_all_gather_host", "needle_hip_comm_shard", "needle_hip_library_job_begin",
    "needle_hip_library_job_end", "needle_hip_library_stream_pcm", "needle_hip_host_alloc",
    "needle_hip_host_alloc_free", "needle_hip_int_valu_ceiling",
    "needle_hip_comparator_results_from_runs", "needle_hip_library_job_runs", "needle_hip_library_job_comm_bytes",
    "needle_hip_library_job_form",
    "needle_hip_host_threads", "needle_hip_fingerprint_audit_device", "needle_hip_library_audit",
    "needle_hip_scan_counts", "needle_hip_scan_last_launch", "needle_hip_epilogue_host_fallbacks"]

_LIB = None


def lib():
    """Loads libneedle_capi.so; raises if it has not been built (no fallback)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'`")
    L = C.CDLL(LIB_PATH)
    vp, sz, u32, u64, b, f32 = C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint64, C.c_bool, C.c_float
    pp = C.POINTER(C.c_char_p)
    L.needle_error_to_str.argtypes = [C.c_int]
    L.needle_error_to_str.restype = C.c_char_p
    L.needle_util_find_video_files.argtypes = [pp, sz, b, b, C.POINTER(pp), C.POINTER(sz)]
    L.needle_util_video_files_free.argtypes = [pp, sz]
    L.needle_util_video_files_free.restype = None
    L.needle_audio_analyzer_new_default.argtypes = [pp, sz, C.POINTER(vp)]
    L.needle_audio_analyzer_new.argtypes = [pp, sz, f32, f32, b, b, b, C.POINTER(vp)]
    L.needle_audio_analyzer_get_frame_hashes.argtypes = [vp, sz, C.POINTER(vp)]
    L.needle_audio_analyzer_free.argtypes = [vp]
    L.needle_audio_analyzer_free.restype = None
    L.needle_audio_analyzer_print_paths.argtypes = [vp]
    L.needle_audio_analyzer_print_paths.restype = None
    L.needle_audio_analyzer_run.argtypes = [vp, f32, b, b]
    L.needle_audio_comparator_new_default.argtypes = [pp, sz, C.POINTER(vp)]
    L.needle_audio_comparator_new.argtypes = [pp, sz, b, C.c_uint16, C.c_uint16, C.c_uint16, f32, C.POINTER(vp)]
    L.needle_audio_comparator_free.argtypes = [vp]
    L.needle_audio_comparator_free.restype = None
    L.needle_audio_comparator_run.argtypes = [vp, b, b, b, b, b]

    L.needle_hip_device_count.argtypes = [C.POINTER(C.c_int)]
    L.needle_hip_set_device.argtypes = [C.c_int]
    L.needle_hip_last_error_message.restype = C.c_char_p
    L.needle_hip_version.restype = C.c_char_p
    L.needle_hip_malloc.argtypes = [C.POINTER(vp), sz]
    L.needle_hip_free.argtypes = [vp]
    L.needle_hip_memcpy_h2d.argtypes = [vp, vp, sz]
    L.needle_hip_memcpy_d2h.argtypes = [vp, vp, sz]
    L.needle_hip_host_free.argtypes = [vp]
    L.needle_hip_host_free.restype = None
    L.needle_hip_last_kernel_ms.argtypes = [C.c_char_p]
    L.needle_hip_last_kernel_ms.restype = C.c_double
    L.needle_hip_set_kernel_timing.argtypes = [C.c_char_p]
    L.needle_hip_set_kernel_timing.restype = None
    L.needle_hip_fingerprint_sample_rate.restype = C.c_int
    L.needle_hip_fingerprint_delay_ms.restype = C.c_int
    L.needle_hip_fingerprint_item_duration_ms.restype = C.c_int
    L.needle_hip_fingerprint_num_items.argtypes = [sz]
    L.needle_hip_fingerprint_num_items.restype = sz
    L.needle_hip_fingerprint_num_kept.argtypes = [sz, u32]
    L.needle_hip_fingerprint_num_kept.restype = sz
    L.needle_hip_fingerprint_host.argtypes = [C.POINTER(vp), C.POINTER(sz), sz, C.c_int, u32, C.POINTER(vp)]
    L.needle_hip_fingerprint_device.argtypes = [vp, C.POINTER(u64), C.POINTER(u64), sz, C.c_int, u32, vp,
                                                C.POINTER(u64), b]
    L.needle_hip_fingerprint_debug.argtypes = [vp, sz, C.c_int, vp, vp]
    L.needle_hip_resample_out_len.argtypes = [sz, C.c_int]
    L.needle_hip_resample_out_len.restype = sz
    L.needle_hip_resample_host.argtypes = [C.POINTER(vp), C.POINTER(sz), sz, C.c_int, C.c_int, C.POINTER(vp)]
    L.needle_hip_hamming_runs_device.argtypes = [vp, C.POINTER(Seq), sz, C.POINTER(Problem), sz, u32, vp, u32, vp, b]
    L.needle_hip_hamming_runs_host.argtypes = [vp, sz, C.POINTER(Seq), sz, C.POINTER(Problem), sz, u32,
                                               C.POINTER(C.POINTER(Run)), C.POINTER(sz)]
    L.needle_hip_frame_hashes_new.argtypes = [vp, vp, sz, vp, vp, sz, u64, C.c_char_p, C.POINTER(vp)]
    L.needle_hip_frame_hashes_free.argtypes = [vp]
    L.needle_hip_frame_hashes_free.restype = None
    L.needle_hip_frame_hashes_len.argtypes = [vp, b]
    L.needle_hip_frame_hashes_len.restype = sz
    L.needle_hip_frame_hashes_copy.argtypes = [vp, b, vp, vp, sz]
    L.needle_hip_frame_hashes_hash_duration_ns.argtypes = [vp]
    L.needle_hip_frame_hashes_hash_duration_ns.restype = u64
    L.needle_hip_frame_hashes_md5.argtypes = [vp]
    L.needle_hip_frame_hashes_md5.restype = C.c_char_p
    L.needle_hip_frame_hashes_read.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.needle_hip_frame_hashes_write.argtypes = [vp, C.c_char_p]
    L.needle_hip_header_md5.argtypes = [C.c_char_p, C.c_char_p]
    L.needle_hip_analyzer_run_pcm.argtypes = [vp, C.POINTER(vp), C.POINTER(sz), C.c_int, C.c_int, f32, b]
    L.needle_hip_comparator_run_with_frame_hashes.argtypes = [vp, C.POINTER(vp), sz, b, b, b,
                                                              C.POINTER(CSearchResult)]
    L.needle_hip_comparator_results_from_runs.argtypes = [vp, C.POINTER(vp), sz, vp, sz, sz, sz,
                                                          C.POINTER(CSearchResult)]
    L.needle_hip_library_new.argtypes = [sz, f32, f32, C.POINTER(vp)]
    L.needle_hip_library_free.argtypes = [vp]
    L.needle_hip_library_free.restype = None
    L.needle_hip_library_include_endings.argtypes = [vp, f32]
    L.needle_hip_library_rows_per_video.argtypes = [vp]
    L.needle_hip_library_rows_per_video.restype = sz
    L.needle_hip_library_set_pcm.argtypes = [vp, C.POINTER(vp), C.POINTER(sz), C.c_int]
    L.needle_hip_library_analyze.argtypes = [vp, sz, sz, b]
    L.needle_hip_library_hash_arena.argtypes = [vp, C.POINTER(vp), C.POINTER(sz)]
    L.needle_hip_library_use_hash_arena.argtypes = [vp, vp, sz, sz]
    L.needle_hip_library_num_pairs.argtypes = [vp]
    L.needle_hip_library_num_pairs.restype = sz
    L.needle_hip_library_search.argtypes = [vp, vp, sz, sz, vp, u32, vp, b]
    L.needle_hip_library_fetch_runs_begin.argtypes = [vp, C.c_int, vp, vp, u32]
    L.needle_hip_library_fetch_runs_end.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(u32)]
    L.needle_hip_library_finalize.argtypes = [vp, vp, vp, sz, C.POINTER(CSearchResult)]
    L.needle_hip_library_frame_hashes.argtypes = [vp, sz, C.POINTER(vp)]
    L.needle_hip_comm_create_id.argtypes = [vp]
    L.needle_hip_comm_init.argtypes = [vp, C.c_int, C.c_int]
    L.needle_hip_comm_finalize.restype = None
    L.needle_hip_comm_backend.restype = C.c_char_p
    L.needle_hip_comm_all_gather_host.argtypes = [vp, vp, sz]
    L.needle_hip_comm_shard.argtypes = [sz, C.c_int, C.c_int, C.POINTER(sz), C.POINTER(sz)]
    L.needle_hip_comm_shard.restype = None
    L.needle_hip_library_stream_pcm.argtypes = [vp, C.POINTER(vp), C.POINTER(sz), C.c_int]
    L.needle_hip_host_alloc.argtypes = [C.POINTER(vp), sz]
    L.needle_hip_host_alloc_free.argtypes = [vp]
    L.needle_hip_library_job_begin.argtypes = [vp, vp, C.c_int]
    L.needle_hip_library_job_end.argtypes = [vp, vp, C.c_int, C.POINTER(CSearchResult), C.POINTER(sz)]
    _LIB = L
    return L


def check(code: int) -> None:
    if code != 0:
        raise NeedleError(code, (lib().needle_hip_last_error_message() or b"").decode(errors="replace"))


def error_to_str(code: int) -> str:
    return lib().needle_error_to_str(code).decode()


def device_count() -> int:
    n = C.c_int(0)
    check(lib().needle_hip_device_count(C.byref(n)))
    return n.value


def set_device(ordinal: int) -> None:
    check(lib().needle_hip_set_device(ordinal))


def device_pci_bus_id() -> str:
    buf = C.create_string_buffer(32)
    lib().needle_hip_device_pci_bus_id.argtypes = [C.c_char_p]
    check(lib().needle_hip_device_pci_bus_id(buf))
    return buf.value.decode()


def synchronize() -> None:
    check(lib().needle_hip_synchronize())


def stream_ptr() -> int:
    """The library's hipStream_t as an integer (e.g. for torch.cuda.ExternalStream)."""
    L = lib()
    L.needle_hip_stream.restype = C.c_void_p
    p = L.needle_hip_stream()
    if not p:
        raise RuntimeError("no HIP device: the library has no stream")
    return int(p)


def cert_stats(reset: bool = False) -> dict:
    """Counts of the certified f32 first pass (needle_hip_fingerprint_cert_stats)."""
    v = (C.c_uint64 * 4)()
    lib().needle_hip_fingerprint_cert_stats.argtypes = [C.POINTER(C.c_uint64), C.c_bool]
    check(lib().needle_hip_fingerprint_cert_stats(v, reset))
    return {"items": int(v[0]), "items_recomputed": int(v[1]), "chunks": int(v[2]), "chunks_recomputed": int(v[3])}


def scan_issued_evaluations(reset: bool = False) -> int:
    """Cell evaluations issued by the counting scan launches (NEEDLE_HIP_SCAN_COUNT=1) since the last reset."""
    v = C.c_uint64(0)
    lib().needle_hip_scan_issued_evaluations.argtypes = [C.POINTER(C.c_uint64), C.c_bool]
    check(lib().needle_hip_scan_issued_evaluations(C.byref(v), reset))
    return int(v.value)


def scan_counts(reset: bool = False) -> Tuple[int, int]:
    """(issued lane evaluations, diagonals that survived the head rows) of the counting scan launches."""
    c = (C.c_uint64 * 2)()
    lib().needle_hip_scan_counts.argtypes = [C.POINTER(C.c_uint64), C.c_bool]
    check(lib().needle_hip_scan_counts(c, reset))
    return int(c[0]), int(c[1])


def scan_last_launch() -> Tuple[int, int]:
    """(form, matrix products) of this process's last scan launch: form 3 = aligned windows on the vector ALU, 4 = with the
    head rows on the matrix pipe (then the second value counts its v_mfma_i32_32x32x32_i8 instructions)."""
    form, products = C.c_int32(0), C.c_uint64(0)
    lib().needle_hip_scan_last_launch.argtypes = [C.POINTER(C.c_int32), C.POINTER(C.c_uint64)]
    check(lib().needle_hip_scan_last_launch(C.byref(form), C.byref(products)))
    return int(form.value), int(products.value)


def epilogue_host_fallbacks(reset: bool = False) -> int:
    """Jobs whose device epilogue fell back to the host form (a pair's bucket of runs too large for one lane)."""
    jobs = C.c_uint64(0)
    lib().needle_hip_epilogue_host_fallbacks.argtypes = [C.POINTER(C.c_uint64), C.c_bool]
    check(lib().needle_hip_epilogue_host_fallbacks(C.byref(jobs), reset))
    return int(jobs.value)


def host_threads() -> int:
    """Host threads this process uses for its parallel host phases (needle_hip_host_threads)."""
    lib().needle_hip_host_threads.restype = C.c_int
    return int(lib().needle_hip_host_threads())


def int_valu_ceiling() -> float:
    """Measured cells/s of the scan's 4-instruction cell on registers (needle_hip_int_valu_ceiling)."""
    v = C.c_double(0.0)
    lib().needle_hip_int_valu_ceiling.argtypes = [C.POINTER(C.c_double)]
    check(lib().needle_hip_int_valu_ceiling(C.byref(v)))
    return v.value


def last_kernel_ms(name: str) -> float:
    return lib().needle_hip_last_kernel_ms(name.encode())


def set_kernel_timing(kernels: Optional[str]) -> None:
    """"all", a comma-separated list of kernel names, or None: which launches get HIP timing events."""
    lib().needle_hip_set_kernel_timing(kernels.encode() if kernels else None)


def _paths(paths: Sequence[str]):
    arr = (C.c_char_p * max(len(paths), 1))(*[p.encode() if isinstance(p, str) else p for p in paths])
    return C.cast(arr, C.POINTER(C.c_char_p)), arr


def header_md5(path: str) -> str:
    buf = C.create_string_buffer(33)
    check(lib().needle_hip_header_md5(path.encode(), buf))
    return buf.value.decode()


# ---- FrameHashes --------------------------------------------------------------------------------------
class FrameHashes:
    """needle::audio::FrameHashes (data.rs:74-168).  Wraps a library-owned or borrowed handle."""

    def __init__(self, handle: int, owned: bool, keepalive=None):
        self._h = C.c_void_p(handle)
        self._owned = owned
        self._keep = keepalive

    @staticmethod
    def new(opening: Sequence[Tuple[int, int]], ending: Sequence[Tuple[int, int]] = (), hash_duration_ns: int = 0,
            md5: str = "") -> "FrameHashes":
        oh = np.array([h for h, _ in opening], dtype=np.uint32)
        ot = np.array([t for _, t in opening], dtype=np.uint64)
        eh = np.array([h for h, _ in ending], dtype=np.uint32)
        et = np.array([t for _, t in ending], dtype=np.uint64)
        out = C.c_void_p()
        check(lib().needle_hip_frame_hashes_new(oh.ctypes.data, ot.ctypes.data, len(oh), eh.ctypes.data,
                                                et.ctypes.data, len(eh), hash_duration_ns, md5.encode(),
                                                C.byref(out)))
        return FrameHashes(out.value, True)

    @staticmethod
    def from_path(path: str) -> "FrameHashes":          # data.rs:104-115
        out = C.c_void_p()
        check(lib().needle_hip_frame_hashes_read(path.encode(), C.byref(out)))
        return FrameHashes(out.value, True)

    def write(self, path: str) -> None:
        check(lib().needle_hip_frame_hashes_write(self._h, path.encode()))

    def _data(self, ending: bool):
        n = lib().needle_hip_frame_hashes_len(self._h, ending)
        hashes = np.zeros(max(n, 1), dtype=np.uint32)
        ts = np.zeros(max(n, 1), dtype=np.uint64)
        check(lib().needle_hip_frame_hashes_copy(self._h, ending, hashes.ctypes.data, ts.ctypes.data, n))
        return hashes[:n], ts[:n]

    def opening_data(self):                              # data.rs:143
        return self._data(False)

    def ending_data(self):                               # data.rs:150
        return self._data(True)

    def hash_duration(self) -> int:                      # data.rs:157 (ns)
        return lib().needle_hip_frame_hashes_hash_duration_ns(self._h)

    def md5(self) -> str:                                # data.rs:164
        return lib().needle_hip_frame_hashes_md5(self._h).decode()

    def __del__(self):
        if getattr(self, "_owned", False) and self._h:
            lib().needle_hip_frame_hashes_free(self._h)
            self._h = None


@dataclass
class SearchResult:                                      # comparator.rs:65-69 (times in ns)
    opening: Optional[Tuple[int, int]]
    ending: Optional[Tuple[int, int]]


def _results(arr, n) -> List[Optional[SearchResult]]:
    out: List[Optional[SearchResult]] = []
    for i in range(n):
        r = arr[i]
        if not r.has_result:
            out.append(None)
        else:
            out.append(SearchResult((r.opening_start_ns, r.opening_end_ns) if r.has_opening else None,
                                    (r.ending_start_ns, r.ending_end_ns) if r.has_ending else None))
    return out


# ---- Analyzer -------------------------------------------------------------------------------------------
class Analyzer:
    """needle::audio::Analyzer (analyzer.rs:86-151) over needle_audio_analyzer_* (needle-capi/src/lib.rs:354-491)."""

    def __init__(self, videos: Sequence[str], threaded_decoding: bool = False, force: bool = False,
                 opening_search_percentage: float = DEFAULT_OPENING_SEARCH_PERCENTAGE,
                 ending_search_percentage: float = DEFAULT_ENDING_SEARCH_PERCENTAGE, include_endings: bool = False):
        self.videos = list(videos)
        self._cfg = dict(opening=opening_search_percentage, ending=ending_search_percentage,
                         include_endings=include_endings, threaded=threaded_decoding, force=force)
        self._h = None

    @staticmethod
    def from_files(videos: Sequence[str], threaded_decoding: bool = False, force: bool = False) -> "Analyzer":
        return Analyzer(videos, threaded_decoding, force)

    def with_opening_search_percentage(self, v: float) -> "Analyzer":
        self._cfg["opening"] = v
        return self

    def with_ending_search_percentage(self, v: float) -> "Analyzer":
        self._cfg["ending"] = v
        return self

    def with_include_endings(self, v: bool) -> "Analyzer":
        self._cfg["include_endings"] = v
        return self

    def with_threaded_decoding(self, v: bool) -> "Analyzer":
        self._cfg["threaded"] = v
        return self

    def with_force(self, v: bool) -> "Analyzer":
        self._cfg["force"] = v
        return self

    def _handle(self):
        if self._h:
            lib().needle_audio_analyzer_free(self._h)
        ptr, keep = _paths(self.videos)
        out = C.c_void_p()
        c = self._cfg
        check(lib().needle_audio_analyzer_new(ptr, len(self.videos), c["opening"], c["ending"], c["include_endings"],
                                              c["threaded"], c["force"], C.byref(out)))
        self._h = out
        return out

    def _collect(self) -> List[FrameHashes]:
        out = []
        for i in range(len(self.videos)):
            fh = C.c_void_p()
            check(lib().needle_audio_analyzer_get_frame_hashes(self._h, i, C.byref(fh)))
            out.append(FrameHashes(fh.value, False, keepalive=self))
        return out

    def run(self, hash_duration: float = DEFAULT_HASH_DURATION, persist: bool = False,
            threading: bool = True) -> List[FrameHashes]:          # analyzer.rs:425
        h = self._handle()
        check(lib().needle_audio_analyzer_run(h, hash_duration, persist, threading))
        return self._collect()

    def run_pcm(self, pcm: Sequence[np.ndarray], channels: int = 1, sample_rate: int = 11025,
                hash_duration: float = DEFAULT_HASH_DURATION, persist: bool = False) -> List[FrameHashes]:
        """Same with decode already done: pcm[i] is the whole stream of video i (interleaved s16)."""
        h = self._handle()
        arrs = [np.ascontiguousarray(p, dtype=np.int16) for p in pcm]
        ptrs = (C.c_void_p * max(len(arrs), 1))(*[a.ctypes.data for a in arrs])
        lens = (C.c_size_t * max(len(arrs), 1))(*[a.size for a in arrs])
        check(lib().needle_hip_analyzer_run_pcm(h, ptrs, lens, channels, sample_rate, hash_duration, persist))
        return self._collect()

    def __del__(self):
        if getattr(self, "_h", None):
            lib().needle_audio_analyzer_free(self._h)
            self._h = None


# ---- Comparator ------------------------------------------------------------------------------------------
class Comparator:
    """needle::audio::Comparator (comparator.rs:74-147) over needle_audio_comparator_* (lib.rs:537-637)."""

    def __init__(self, videos: Sequence[str], include_endings: bool = False,
                 hash_match_threshold: int = DEFAULT_HASH_MATCH_THRESHOLD,
                 min_opening_duration: int = DEFAULT_MIN_OPENING_DURATION,
                 min_ending_duration: int = DEFAULT_MIN_ENDING_DURATION, time_padding: float = 0.0):
        self.videos = list(videos)
        self._cfg = dict(include_endings=include_endings, threshold=hash_match_threshold,
                         min_opening=min_opening_duration, min_ending=min_ending_duration, padding=time_padding)
        self._h = None

    @staticmethod
    def from_files(videos: Sequence[str]) -> "Comparator":
        return Comparator(videos)

    def with_include_endings(self, v: bool) -> "Comparator":
        self._cfg["include_endings"] = v
        return self

    def with_hash_match_threshold(self, v: int) -> "Comparator":
        self._cfg["threshold"] = v
        return self

    def with_min_opening_duration(self, secs: int) -> "Comparator":
        self._cfg["min_opening"] = secs
        return self

    def with_min_ending_duration(self, secs: int) -> "Comparator":
        self._cfg["min_ending"] = secs
        return self

    def with_time_padding(self, secs: float) -> "Comparator":
        self._cfg["padding"] = secs
        return self

    def handle(self):
        if self._h:
            lib().needle_audio_comparator_free(self._h)
        ptr, keep = _paths(self.videos)
        out = C.c_void_p()
        c = self._cfg
        check(lib().needle_audio_comparator_new(ptr, len(self.videos), c["include_endings"], c["threshold"],
                                                c["min_opening"], c["min_ending"], c["padding"], C.byref(out)))
        self._h = out
        return out

    def run_with_frame_hashes(self, frame_hashes: Sequence[FrameHashes], display: bool = False,
                              use_skip_files: bool = False, write_skip_files: bool = False
                              ) -> List[Optional[SearchResult]]:       # comparator.rs:524
        """One entry per video; None where the reference pushes no result (comparator.rs:608-617)."""
        h = self.handle()
        n = len(frame_hashes)
        ptrs = (C.c_void_p * max(n, 1))(*[f._h for f in frame_hashes])
        res = (CSearchResult * max(n, 1))()
        check(lib().needle_hip_comparator_run_with_frame_hashes(h, ptrs, n, display, use_skip_files,
                                                                write_skip_files, res))
        return _results(res, n)

    def results_from_runs(self, frame_hashes: Sequence[FrameHashes], runs: np.ndarray, first_video: int = 0,
                          video_count: Optional[int] = None) -> List[Optional[SearchResult]]:
        """The host epilogue alone (needle_hip_comparator_results_from_runs): no device work."""
        n = len(frame_hashes)
        runs = np.ascontiguousarray(runs, dtype=RUN_DTYPE)
        arr = (C.c_void_p * max(n, 1))(*[f._h for f in frame_hashes])
        res = (CSearchResult * max(n, 1))()
        check(lib().needle_hip_comparator_results_from_runs(
            self._h or self.handle(), arr, n, runs.ctypes.data, runs.size, first_video,
            n - first_video if video_count is None else video_count, res))
        return _results(res, n)

    def run(self, analyze: bool, display: bool = False, use_skip_files: bool = False,
            write_skip_files: bool = False, threading: bool = True) -> None:   # comparator.rs:637 / lib.rs:612
        h = self.handle()
        check(lib().needle_audio_comparator_run(h, analyze, display, use_skip_files, write_skip_files, threading))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().needle_audio_comparator_free(self._h)
            self._h = None


# ---- kernel-level entry points ------------------------------------------------------------------------------
def fingerprint(pcms: Sequence[np.ndarray], channels: int = 1, step: int = 1) -> List[np.ndarray]:
    """needle_hip_fingerprint_host: raw chromaprint items (every `step`-th) of each stream."""
    arrs = [np.ascontiguousarray(p, dtype=np.int16) for p in pcms]
    n = len(arrs)
    outs = [np.zeros(max(lib().needle_hip_fingerprint_num_kept(a.size // channels, step), 1), dtype=np.uint32)
            for a in arrs]
    ptrs = (C.c_void_p * max(n, 1))(*[a.ctypes.data for a in arrs])
    lens = (C.c_size_t * max(n, 1))(*[a.size for a in arrs])
    optrs = (C.c_void_p * max(n, 1))(*[o.ctypes.data for o in outs])
    check(lib().needle_hip_fingerprint_host(ptrs, lens, n, channels, step, optrs))
    return [o[:lib().needle_hip_fingerprint_num_kept(a.size // channels, step)] for o, a in zip(outs, arrs)]


def fingerprint_debug(pcm: np.ndarray, channels: int = 1):
    """(chroma [frames][12], features [frames-4][12]) of one stream, computed on the GPU."""
    a = np.ascontiguousarray(pcm, dtype=np.int16)
    samples = a.size // channels
    frames = 0 if samples < 4096 else (samples - 4096) // 1365 + 1
    chroma = np.zeros((max(frames, 1), 12))
    feats = np.zeros((max(frames - 4, 1), 12))
    check(lib().needle_hip_fingerprint_debug(a.ctypes.data, a.size, channels, chroma.ctypes.data, feats.ctypes.data))
    return chroma[:frames], feats[:max(frames - 4, 0)]


def resample(pcms: Sequence[np.ndarray], channels: int, sample_rate: int) -> List[np.ndarray]:
    """needle_hip_resample_host: interleaved s16 at `sample_rate` -> mono s16 at 11025 Hz."""
    arrs = [np.ascontiguousarray(p, dtype=np.int16) for p in pcms]
    n = len(arrs)
    lens_out = [lib().needle_hip_resample_out_len(a.size // channels, sample_rate) for a in arrs]
    outs = [np.zeros(max(k, 1), dtype=np.int16) for k in lens_out]
    ptrs = (C.c_void_p * max(n, 1))(*[a.ctypes.data for a in arrs])
    lens = (C.c_size_t * max(n, 1))(*[a.size for a in arrs])
    optrs = (C.c_void_p * max(n, 1))(*[o.ctypes.data for o in outs])
    check(lib().needle_hip_resample_host(ptrs, lens, n, channels, sample_rate, optrs))
    return [o[:k] for o, k in zip(outs, lens_out)]


def hamming_runs(seqs: Sequence[np.ndarray], problems: Sequence[Tuple[int, int, int]], threshold: int) -> np.ndarray:
    """needle_hip_hamming_runs_host.  problems: (src_seq, dst_seq, min_len); returns a structured array of
    (problem, src_end, dst_end, len), problem = index into `problems`."""
    arena = np.concatenate([np.ascontiguousarray(s, dtype=np.uint32) for s in seqs]) if seqs else np.zeros(0, np.uint32)
    arena = np.ascontiguousarray(arena, dtype=np.uint32)
    cs = (Seq * max(len(seqs), 1))()
    off = 0
    for i, s in enumerate(seqs):
        cs[i] = Seq(off, len(s))
        off += len(s)
    cp = (Problem * max(len(problems), 1))(*[Problem(a, b, m, i) for i, (a, b, m) in enumerate(problems)])
    runs = C.POINTER(Run)()
    n = C.c_size_t(0)
    check(lib().needle_hip_hamming_runs_host(arena.ctypes.data, arena.size, cs, len(seqs), cp, len(problems),
                                             threshold, C.byref(runs), C.byref(n)))
    out = np.zeros(n.value, dtype=RUN_DTYPE)
    if n.value:
        C.memmove(out.ctypes.data, runs, n.value * C.sizeof(Run))
    lib().needle_hip_host_free(runs)
    return out


# ---- communicator: one process per GPU (include/needle_hip.h, "multi-GPU") --------------------------------------
COMM_ID_BYTES = 128


def comm_create_id() -> bytes:
    buf = (C.c_uint8 * COMM_ID_BYTES)()
    check(lib().needle_hip_comm_create_id(buf))
    return bytes(buf)


def comm_init(comm_id: bytes, rank: int, world: int) -> None:
    assert len(comm_id) == COMM_ID_BYTES
    buf = (C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(comm_id)
    check(lib().needle_hip_comm_init(buf, rank, world))


def comm_finalize() -> None:
    lib().needle_hip_comm_finalize()


def comm_rank() -> int:
    return lib().needle_hip_comm_rank()


def comm_world_size() -> int:
    return lib().needle_hip_comm_world_size()


def comm_backend() -> str:
    return lib().needle_hip_comm_backend().decode()


def comm_barrier() -> None:
    check(lib().needle_hip_comm_barrier())


def comm_all_gather(local: np.ndarray) -> np.ndarray:
    """All-gather of equal-sized host arrays: returns [world, *local.shape]."""
    local = np.ascontiguousarray(local)
    out = np.zeros((comm_world_size(),) + local.shape, dtype=local.dtype)
    check(lib().needle_hip_comm_all_gather_host(local.ctypes.data, out.ctypes.data, local.nbytes))
    return out


def comm_shard(units: int, world: int, rank: int) -> Tuple[int, int]:
    first, count = C.c_size_t(0), C.c_size_t(0)
    lib().needle_hip_comm_shard(units, world, rank, C.byref(first), C.byref(count))
    return first.value, count.value


class CCertAudit(C.Structure):
    _fields_ = [("items", C.c_uint64), ("accepted", C.c_uint64), ("accepted_mismatches", C.c_uint64),
                ("mismatches", C.c_uint64), ("max_error_over_s", C.c_double), ("max_s", C.c_double)]

    def as_dict(self) -> dict:
        return {k: getattr(self, k) for k, _ in self._fields_}


def fingerprint_audit_device(d_pcm: int, pcm_offsets, num_values, channels: int, step: int, d_items: int, item_offsets) -> dict:
    """needle_hip_fingerprint_audit_device: f32 first pass vs f64 kernel over the same resident PCM."""
    n = len(num_values)
    a = CCertAudit()
    u64 = C.c_uint64 * n
    lib().needle_hip_fingerprint_audit_device.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_size_t,
                                                          C.c_int, C.c_uint32, C.c_void_p, C.POINTER(C.c_uint64),
                                                          C.POINTER(CCertAudit)]
    check(lib().needle_hip_fingerprint_audit_device(d_pcm, u64(*pcm_offsets), u64(*num_values), n, channels, step, d_items,
                                                    u64(*item_offsets), C.byref(a)))
    return a.as_dict()


# ---- HBM-resident library (bench / multi-GPU) -----------------------------------------------------------------
class Library:
    """NeedleHipLibrary: PCM resident in HBM, padded device hash arena, pair-sharded search."""

    def __init__(self, num_videos: int, opening_search_percentage: float = DEFAULT_OPENING_SEARCH_PERCENTAGE,
                 hash_duration: float = DEFAULT_HASH_DURATION):
        out = C.c_void_p()
        check(lib().needle_hip_library_new(num_videos, opening_search_percentage, hash_duration, C.byref(out)))
        self._h = out
        self.n = num_videos

    def include_endings(self, ending_search_percentage: float = DEFAULT_ENDING_SEARCH_PERCENTAGE) -> "Library":
        check(lib().needle_hip_library_include_endings(self._h, ending_search_percentage))
        return self

    def rows_per_video(self) -> int:
        return lib().needle_hip_library_rows_per_video(self._h)

    def set_pcm(self, pcm: Sequence[Optional[np.ndarray]], num_values: Sequence[int], channels: int = 1) -> None:
        arrs = [None if p is None else np.ascontiguousarray(p, dtype=np.int16) for p in pcm]
        ptrs = (C.c_void_p * self.n)(*[None if a is None else a.ctypes.data for a in arrs])
        lens = (C.c_size_t * self.n)(*list(num_values))
        check(lib().needle_hip_library_set_pcm(self._h, ptrs, lens, channels))

    def rank_videos(self, num_values: Sequence[int], world: int, rank: int, channels: int = 1) -> Tuple[int, int]:
        """(first, count) of the videos whose PCM `rank` of `world` must hold (needle_hip_library_rank_videos)."""
        lens = (C.c_size_t * self.n)(*list(num_values))
        first, count = C.c_size_t(0), C.c_size_t(0)
        lib().needle_hip_library_rank_videos.argtypes = [C.c_void_p, C.POINTER(C.c_size_t), C.c_int, C.c_int, C.c_int,
                                                         C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        check(lib().needle_hip_library_rank_videos(self._h, lens, channels, world, rank, C.byref(first), C.byref(count)))
        return first.value, count.value

    def set_pcm_device(self, d_ptrs: Sequence[Optional[int]], num_values: Sequence[int], channels: int = 1) -> None:
        """PCM already in HBM: device pointers (None for videos of other ranks), copied device to device."""
        ptrs = (C.c_void_p * self.n)(*[None if p is None else int(p) for p in d_ptrs])
        lens = (C.c_size_t * self.n)(*list(num_values))
        lib().needle_hip_library_set_pcm_device.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_int]
        check(lib().needle_hip_library_set_pcm_device(self._h, ptrs, lens, channels))

    def stream_pcm(self, pcm: Sequence[Optional[np.ndarray]], num_values: Sequence[int], channels: int = 1) -> None:
        """Upload + fingerprint overlapped, PCM not kept (needle_hip_library_stream_pcm)."""
        arrs = [None if p is None else np.ascontiguousarray(p, dtype=np.int16) for p in pcm]
        ptrs = (C.c_void_p * self.n)(*[None if a is None else a.ctypes.data for a in arrs])
        lens = (C.c_size_t * self.n)(*list(num_values))
        check(lib().needle_hip_library_stream_pcm(self._h, ptrs, lens, channels))

    def analyze(self, first: int = 0, count: Optional[int] = None, sync: bool = True) -> None:
        check(lib().needle_hip_library_analyze(self._h, first, self.n - first if count is None else count, sync))

    def hash_arena(self) -> Tuple[int, int]:
        ptr = C.c_void_p()
        stride = C.c_size_t(0)
        check(lib().needle_hip_library_hash_arena(self._h, C.byref(ptr), C.byref(stride)))
        return ptr.value, stride.value

    def use_hash_arena(self, d_ptr: int, rows: int, stride: int) -> None:
        check(lib().needle_hip_library_use_hash_arena(self._h, d_ptr, rows, stride))

    def num_pairs(self) -> int:
        return lib().needle_hip_library_num_pairs(self._h)

    def search(self, comparator: Comparator, first_pair: int, num_pairs: int, d_runs: int, capacity: int,
               d_count: int, sync: bool = True) -> None:
        check(lib().needle_hip_library_search(self._h, comparator._h or comparator.handle(), first_pair, num_pairs,
                                              d_runs, capacity, d_count, sync))

    def fetch_runs_begin(self, slot: int, d_runs: int, d_count: int, max_runs: int) -> None:
        check(lib().needle_hip_library_fetch_runs_begin(self._h, slot, d_runs, d_count, max_runs))

    def fetch_runs_end(self, slot: int, max_runs: int) -> Tuple[np.ndarray, int]:
        """(runs actually downloaded, total found).  The array is a copy, so the slot can be reused."""
        ptr = C.c_void_p()
        total = C.c_uint32(0)
        check(lib().needle_hip_library_fetch_runs_end(self._h, slot, C.byref(ptr), C.byref(total)))
        k = min(total.value, max_runs)
        out = np.zeros(k, dtype=RUN_DTYPE)
        if k:
            C.memmove(out.ctypes.data, ptr.value, k * RUN_DTYPE.itemsize)
        return out, total.value

    def finalize(self, comparator: Comparator, runs: np.ndarray) -> List[Optional[SearchResult]]:
        runs = np.ascontiguousarray(runs, dtype=RUN_DTYPE)
        res = (CSearchResult * self.n)()
        check(lib().needle_hip_library_finalize(self._h, comparator._h or comparator.handle(), runs.ctypes.data,
                                                runs.size, res))
        return _results(res, self.n)

    def frame_hashes(self, index: int) -> FrameHashes:
        out = C.c_void_p()
        check(lib().needle_hip_library_frame_hashes(self._h, index, C.byref(out)))
        return FrameHashes(out.value, True)

    def job_begin(self, comparator: Comparator, slot: int = 0) -> None:
        """Enqueues one analyze+search job of this rank's share (all of it without a communicator)."""
        check(lib().needle_hip_library_job_begin(self._h, comparator._h or comparator.handle(), slot))

    def job_end(self, comparator: Comparator, slot: int = 0) -> Tuple[List[Optional[SearchResult]], int]:
        """(results for all videos -- the same on every rank, runs found over all pairs)."""
        res = (CSearchResult * self.n)()
        found = C.c_size_t(0)
        check(lib().needle_hip_library_job_end(self._h, comparator._h or comparator.handle(), slot, res, C.byref(found)))
        return _results(res, self.n), found.value

    def audit(self) -> dict:
        """Both transforms over this rank's resident PCM, every kept item compared on the device (needle_hip_library_audit)."""
        a = CCertAudit()
        lib().needle_hip_library_audit.argtypes = [C.c_void_p, C.POINTER(CCertAudit)]
        check(lib().needle_hip_library_audit(self._h, C.byref(a)))
        return a.as_dict()

    def job_runs(self, slot: int = 0) -> np.ndarray:
        """The complete run list of the job that finished last in `slot` (a copy; needle_hip_library_job_runs)."""
        ptr, total = C.c_void_p(), C.c_size_t(0)
        lib().needle_hip_library_job_runs.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        check(lib().needle_hip_library_job_runs(self._h, slot, C.byref(ptr), C.byref(total)))
        out = np.zeros(total.value, dtype=RUN_DTYPE)
        if total.value:
            C.memmove(out.ctypes.data, ptr.value, out.nbytes)
        return out

    def job_form(self, slot: int = 0) -> dict:
        """Which forms the slot's last finished job took (needle_hip_library_job_form)."""
        f = (C.c_uint32 * 4)()
        lib().needle_hip_library_job_form.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint32)]
        check(lib().needle_hip_library_job_form(self._h, slot, f))
        return {"device_epilogue": bool(f[0]), "sharded_epilogue": bool(f[1]), "directed_runs": bool(f[2]), "scan_form": int(f[3])}

    def job_comm_bytes(self, slot: int = 0) -> dict:
        b = (C.c_uint64 * 4)()
        lib().needle_hip_library_job_comm_bytes.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint64)]
        check(lib().needle_hip_library_job_comm_bytes(self._h, slot, b))
        return {"hash_rows": int(b[0]), "run_heads": int(b[1]), "results": int(b[2]), "scans_repeated": int(b[3])}

    def __del__(self):
        if getattr(self, "_h", None):
            lib().needle_hip_library_free(self._h)
            self._h = None


class PinnedArray:
    """An int16 numpy array over page-locked host memory (needle_hip_host_alloc): PCM the copy engine reads in place."""

    def __init__(self, count: int):
        p = C.c_void_p()
        check(lib().needle_hip_host_alloc(C.byref(p), max(count, 1) * 2))
        self.ptr = p.value
        self.array = np.ctypeslib.as_array((C.c_int16 * max(count, 1)).from_address(self.ptr))[:count]

    def __del__(self):
        if getattr(self, "ptr", None):
            self.array = None
            lib().needle_hip_host_alloc_free(self.ptr)
            self.ptr = None


class DeviceBuffer:
    """hipMalloc'd scratch owned by Python (run lists, counters)."""

    def __init__(self, nbytes: int):
        p = C.c_void_p()
        check(lib().needle_hip_malloc(C.byref(p), nbytes))
        self.ptr = p.value
        self.nbytes = nbytes

    def to_host(self, dtype, count: int) -> np.ndarray:
        out = np.zeros(count, dtype=dtype)
        if count:
            check(lib().needle_hip_memcpy_d2h(out.ctypes.data, self.ptr, out.nbytes))
        return out

    def __del__(self):
        if getattr(self, "ptr", None):
            lib().needle_hip_free(self.ptr)
            self.ptr = None
