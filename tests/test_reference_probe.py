"""Optional pin of the oracle against the REAL third-party arithmetic (SURVEY.md §7 hard part 1).

The reference's analyze stage is libchromaprint 1.5.x (chromaprint-sys-next 1.5.3, needle/Cargo.lock:158-165), which is
not vendored and not installed in the build image.  If a system libchromaprint (or `fpcalc`) happens to exist where the
tests run, the oracle's raw fingerprints of the synthetic episodes are compared with it and the agreement is REPORTED
(exact-item rate, mean Hamming distance per item) in the test log and gpurun_out/chromaprint_probe.json: a build of
chromaprint on an f32 FFT (avfft / kissfft / vDSP) is expected to disagree in low-order bits of a few items even with
a perfect restatement, so only gross disagreement fails.  Absent library: skipped, never failed.
"""
import ctypes as C
import ctypes.util
import json
import os
import shutil
import subprocess
import tempfile

import numpy as np
import pytest

from needle_amd import synth
from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _real_chromaprint():
    """Returns f(pcm int16 mono @ 11025) -> np.uint32 raw items, or None."""
    name = os.environ.get("NEEDLE_REAL_CHROMAPRINT") or ctypes.util.find_library("chromaprint")
    if name:
        try:
            L = C.CDLL(name)
            L.chromaprint_new.restype = C.c_void_p
            L.chromaprint_new.argtypes = [C.c_int]
            L.chromaprint_start.argtypes = [C.c_void_p, C.c_int, C.c_int]
            L.chromaprint_feed.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
            L.chromaprint_finish.argtypes = [C.c_void_p]
            L.chromaprint_get_raw_fingerprint.argtypes = [C.c_void_p, C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(C.c_int)]
            L.chromaprint_dealloc.argtypes = [C.c_void_p]
            L.chromaprint_free.argtypes = [C.c_void_p]
            if b"needle" in (C.cast(L.chromaprint_get_version, C.CFUNCTYPE(C.c_char_p))() or b""):
                return None                      # that is our own libneedle_chromaprint.so, not the real thing

            def run(pcm):
                ctx = L.chromaprint_new(1)       # CHROMAPRINT_ALGORITHM_TEST2 = the default (analyzer.rs:176)
                assert L.chromaprint_start(ctx, 11025, 1) == 1
                pcm = np.ascontiguousarray(pcm, dtype=np.int16)
                assert L.chromaprint_feed(ctx, pcm.ctypes.data, pcm.size) == 1
                assert L.chromaprint_finish(ctx) == 1
                p, n = C.POINTER(C.c_uint32)(), C.c_int(0)
                assert L.chromaprint_get_raw_fingerprint(ctx, C.byref(p), C.byref(n)) == 1
                out = np.ctypeslib.as_array(p, shape=(n.value,)).copy() if n.value else np.zeros(0, np.uint32)
                L.chromaprint_dealloc(p)
                L.chromaprint_free(ctx)
                return out
            return run
        except (OSError, AttributeError):
            pass
    fpcalc = shutil.which("fpcalc")
    if fpcalc:
        def run(pcm):
            with tempfile.TemporaryDirectory() as d:
                path = os.path.join(d, "x.wav")
                synth.write_wav(path, np.ascontiguousarray(pcm, dtype=np.int16))
                out = subprocess.run([fpcalc, "-raw", "-length", "0", "-json", path], capture_output=True, text=True, check=True)
                return np.array(json.loads(out.stdout)["fingerprint"], dtype=np.int64).astype(np.uint32)
        return run
    return None


def test_oracle_against_a_real_libchromaprint_if_one_exists():
    real = _real_chromaprint()
    if real is None:
        pytest.skip("no system libchromaprint / fpcalc: the analyze stage stays pinned by chromaprint's own vectors only "
                    "(DESIGN.md §2, 'parity unpinned end to end on non-silent audio')")
    report = []
    for k, e in enumerate(synth.make_library(3, 90.0, 20.0)):
        want = real(e.pcm)
        got = O.fingerprint(e.pcm)
        assert len(got) == len(want), "item count (frame / latency arithmetic) differs from the real library"
        dist = np.array([bin(int(a) ^ int(b)).count("1") for a, b in zip(got.tolist(), want.tolist())])
        report.append({"episode": k, "items": len(want), "exact_items": int((dist == 0).sum()),
                       "exact_rate": float((dist == 0).mean()), "mean_hamming_bits": float(dist.mean()),
                       "max_hamming_bits": int(dist.max())})
    print("chromaprint probe:", json.dumps(report))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "chromaprint_probe.json"), "w") as f:
        json.dump(report, f)
    assert all(r["mean_hamming_bits"] < 1.0 for r in report), "the restatement is not chromaprint's algorithm"
