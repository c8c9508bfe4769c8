#!/usr/bin/env python3
"""PCIe-inclusive time of one analyze+search job on BASELINE.json configs[1] (28 x 24 min): PCM starts in host
memory (numpy arrays), is uploaded (Library.set_pcm), fingerprinted, searched and finalised.  Not the headline
metric (bench.py times the job with PCM resident in HBM); the number goes to DESIGN.md."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from needle_amd import capi, synth  # noqa: E402

n = 28
eps = synth.make_library(n, 24 * 60.0, 90.0)
cmp = capi.Comparator([f"episode-{k:04d}.wav" for k in range(n)])
cmp.handle()
cap = 1 << 16
d_runs, d_count = capi.DeviceBuffer(cap * capi.RUN_DTYPE.itemsize), capi.DeviceBuffer(4)
for rep in range(4):
    t0 = time.perf_counter()
    lib = capi.Library(n)
    lib.set_pcm([e.pcm for e in eps], [len(e.pcm) for e in eps])
    t1 = time.perf_counter()
    lib.analyze(0, n, sync=False)
    lib.search(cmp, 0, lib.num_pairs(), d_runs.ptr, cap, d_count.ptr, sync=True)
    found = int(d_count.to_host("uint32", 1)[0])
    runs = d_runs.to_host(capi.RUN_DTYPE, found)
    res = lib.finalize(cmp, runs)
    t2 = time.perf_counter()
    up = sum(len(e.pcm) // 2 for e in eps) * 2 / 1e6
    print(f"rep {rep}: upload {1e3 * (t1 - t0):.2f} ms ({up:.0f} MB, {up / (t1 - t0) / 1e3:.1f} GB/s), "
          f"analyze+search+epilogue {1e3 * (t2 - t1):.2f} ms, job {1e3 * (t2 - t0):.2f} ms = "
          f"{lib.num_pairs() / (t2 - t0):.0f} pairs/s, detected {sum(1 for r in res if r is not None and r.opening)}")
    del lib
