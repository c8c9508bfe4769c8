"""TEST INFRASTRUCTURE (like everything under oracle/): the one cheap chance of pinning the oracle's analyze stage to the
REAL third-party arithmetic.  The reference fingerprints with libchromaprint 1.5.x (chromaprint-sys-next 1.5.3,
needle/Cargo.lock:158-165; call sites needle/src/audio/analyzer.rs:176,218,275,286,300), which is neither vendored
nor installed in the build image.  Where a system libchromaprint (or `fpcalc`) exists -- probed at run time, never
required -- the oracle's raw fingerprints of three synthetic episodes are compared with it.  Used by
tests/test_reference_probe.py (CPU and `-m gpu` copies) and by bench.py's `oracle_pin` field (cpu_baseline leg)."""
import ctypes as C
import ctypes.util
import json
import os
import shutil
import subprocess
import tempfile

import numpy as np


def real_chromaprint():
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
                _synth().write_wav(path, np.ascontiguousarray(pcm, dtype=np.int16))
                out = subprocess.run([fpcalc, "-raw", "-length", "0", "-json", path], capture_output=True, text=True, check=True)
                return np.array(json.loads(out.stdout)["fingerprint"], dtype=np.int64).astype(np.uint32)
        return run
    return None


def _synth():
    from needle_amd import synth
    return synth


def probe_report():
    """{"libchromaprint": "found" | "absent", "exact_rate", "mean_hamming_bits", "episodes": [...]}: agreement of
    oracle.fingerprint with the real library on BASELINE.json configs[0]'s three 90 s episodes.  A chromaprint built on
    an f32 FFT (avfft / kissfft / vDSP) may differ from the f64 restatement in low-order bits of a few items."""
    real = real_chromaprint()
    if real is None:
        return {"libchromaprint": "absent", "exact_rate": None, "mean_hamming_bits": None,
                "what": "no system libchromaprint / fpcalc on this machine: the oracle's analyze stage stays pinned by "
                        "chromaprint's own recalled vectors only (parity unpinned)"}
    from oracle import oracle as O
    eps, items, exact, bits = [], 0, 0, 0
    for k, e in enumerate(_synth().make_library(3, 90.0, 20.0)):
        want, got = real(e.pcm), O.fingerprint(e.pcm)
        n = min(len(got), len(want))
        dist = np.array([bin(int(a) ^ int(b)).count("1") for a, b in zip(got[:n].tolist(), want[:n].tolist())])
        eps.append({"episode": k, "items": len(want), "oracle_items": len(got), "exact_items": int((dist == 0).sum()),
                    "exact_rate": float((dist == 0).mean()) if n else 0.0, "mean_hamming_bits": float(dist.mean()) if n else 32.0,
                    "max_hamming_bits": int(dist.max()) if n else 32})
        items, exact, bits = items + n, exact + int((dist == 0).sum()), bits + int(dist.sum())
    return {"libchromaprint": "found", "exact_rate": exact / max(items, 1), "mean_hamming_bits": bits / max(items, 1),
            "items": items, "episodes": eps}
