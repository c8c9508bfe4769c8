# The hostile corpus's headline job (28 x 24 min) under whichever library NEEDLE_CAPI_LIB names: the scan kernel alone, the
# job alone and the steady rate with two jobs in flight -- for A/B builds of search.hip (tools/build_variant.sh).
# usage: NEEDLE_CAPI_LIB=<path> python tools/hostile_scan_ab.py [tonal]
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from needle_amd import capi, synth
if "tonal" in sys.argv[1:]:
    import time
    n, samples = 28, int(12 * 60 * 11025)
    gen = synth.DeviceLibrary(n, samples, 90.0)
    lib = capi.Library(n, opening_search_percentage=1.0)
    lib.set_pcm_device(gen.pointers(), [samples] * n)
    cmp = capi.Comparator([f"e{k}.wav" for k in range(n)])
    for _ in range(60):
        lib.job_begin(cmp, 0); lib.job_end(cmp, 0)
    capi.set_kernel_timing("all,sum")
    best = 1e9
    for _ in range(10):
        lib.job_begin(cmp, 0); lib.job_end(cmp, 0)
        best = min(best, capi.last_kernel_ms("hamming_runs"))
    print("tonal 28 x 24 min: hamming_runs best of 10 alone", round(best, 4), "ms")
else:
    out = bench.corpus_hostile(capi, synth, 28, 24.0, jobs=40, check=1)
    print("hostile 28 x 24 min:", os.environ.get("NEEDLE_CAPI_LIB", "default lib"), "ms_per_step", out["ms_per_step"], "alone", out["latency_ms_one_job"],
          "kernels", out["kernel_ms_one_job_alone"], "runs", out["runs_per_step"], "issued", out["scan_issued_evaluations"])
