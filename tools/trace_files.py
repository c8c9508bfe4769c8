#!/usr/bin/env python3
"""Phase times (NEEDLE_HIP_TRACE) of file-based analyzer runs over a set made by tools/bench_files.py; the first
run includes the one-time costs (pinned ring, device arenas)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from needle_amd import capi  # noqa: E402

d = sys.argv[1]
paths = sorted(os.path.join(d, f) for f in os.listdir(d) if f.endswith(".wav"))
capi.device_count()
os.environ["NEEDLE_HIP_TRACE"] = "1"
for rep in range(3):
    print(f"--- run {rep}", file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    capi.Analyzer.from_files(paths, force=True).run(0.3)
    print(f"run {rep}: {1e3 * (time.perf_counter() - t0):.2f} ms", file=sys.stderr, flush=True)
