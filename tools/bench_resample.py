#!/usr/bin/env python3
"""Kernel time and achieved HBM bandwidth of the resampler front-end (not the headline metric)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from needle_amd import capi  # noqa: E402

capi.set_kernel_timing("all")

for rate, ch in ((48000, 2), (44100, 2), (48000, 1)):
    n = rate * 720                                   # one 12-minute opening window
    rng = np.random.default_rng(rate)
    pcms = [rng.integers(-20000, 20000, n * ch, dtype=np.int16) for _ in range(8)]
    for _ in range(3):
        out = capi.resample(pcms, ch, rate)
        ms = capi.last_kernel_ms("resample")
    in_bytes = sum(p.nbytes for p in pcms)
    out_bytes = sum(o.nbytes for o in out)
    print(f"{rate} Hz x{ch}: 8 streams x 720 s, {in_bytes / 1e9:.2f} GB in, {out_bytes / 1e6:.0f} MB out, "
          f"kernel {ms:.3f} ms = {(in_bytes + out_bytes) / ms / 1e6:.0f} GB/s "
          f"({(in_bytes + out_bytes) / ms / 1e6 / 8000 * 100:.1f} % of 8 TB/s)")
