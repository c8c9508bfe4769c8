import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from needle_amd import capi
capi.set_kernel_timing("all")
for rate, ch in ((96000, 2), (32000, 2), (16000, 2), (24000, 2), (8000, 2)):
    n = rate * 720
    rng = np.random.default_rng(rate)
    pcms = [rng.integers(-20000, 20000, n * ch, dtype=np.int16) for _ in range(4)]
    for _ in range(3):
        out = capi.resample(pcms, ch, rate)
        ms = capi.last_kernel_ms("resample")
    b = sum(p.nbytes for p in pcms) + sum(o.nbytes for o in out)
    print(f"{rate} Hz x{ch}: 4 streams x 720 s, kernel {ms:.3f} ms = {b / ms / 1e6:.0f} GB/s ({b / ms / 1e6 / 80:.1f} % of 8 TB/s)")
