#!/usr/bin/env python3
"""Wall-clock and peak host memory of the FILE-based analyzer (`needle_audio_analyzer_run`, what `needle analyze
<dir>` calls) over a library of WAV files: the search windows are read from the files, uploaded, resampled on the
device when needed and fingerprinted.  Not the headline metric (bench.py starts from PCM resident in HBM).

    python tools/bench_files.py [--lib needle_amd/lib/ab/files_old.so] [--dir /tmp/needle_files]

Each measurement runs in its own process so that ru_maxrss is the peak of that run alone."""
import argparse
import json
import os
import resource
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SETS = {  # name: (episodes, minutes, channels, rate)
    "28x24min_11025_mono": (28, 24, 1, 11025),
    "12x24min_44100_stereo": (12, 24, 2, 44100),
}


def make(dirname, name):
    import numpy as np
    from needle_amd import synth
    n, minutes, ch, rate = SETS[name]
    d = os.path.join(dirname, name)
    if os.path.isdir(d) and len(os.listdir(d)) == n:
        return d
    os.makedirs(d, exist_ok=True)
    for k, e in enumerate(synth.make_library(n, minutes * 60.0, 90.0)):
        x = e.pcm
        if rate != 11025:
            x = np.repeat(x, rate // 11025)
            x = ((x.astype(np.int32) + np.roll(x, 1)) // 2).astype(np.int16)
        synth.write_wav(os.path.join(d, f"episode-{k:03d}.wav"), x, channels=ch, rate=rate)
    return d


def child(d, threading, endings):
    from needle_amd import capi
    paths = sorted(os.path.join(d, f) for f in os.listdir(d) if f.endswith(".wav"))
    capi.Analyzer.from_files(paths[:1], force=True).run(0.3)  # device + tables up
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        fhs = capi.Analyzer.from_files(paths, force=True).with_include_endings(endings).run(0.3, threading=threading)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    size = sum(os.path.getsize(p) for p in paths)
    print(json.dumps({"files": len(paths), "file_GB": round(size / 1e9, 2), "seconds": round(best, 3),
                      "episodes_per_s": round(len(paths) / best, 1), "hashes": sum(len(f.opening_data()[0]) for f in fhs),
                      "peak_rss_GB": round(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6, 2)}))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--dir", default="/tmp/needle_files")
    ap.add_argument("--lib", action="append", default=None, help="library to time (repeatable); default: the product")
    ap.add_argument("--child", nargs=3)
    a = ap.parse_args()
    if a.child:
        child(a.child[0], a.child[1] == "1", a.child[2] == "1")
        sys.exit(0)
    for name in SETS:
        d = make(a.dir, name)
        for lib in a.lib or [""]:
            for threading, endings in ((True, False), (False, False), (True, True)):
                env = dict(os.environ)
                if lib:
                    env["NEEDLE_CAPI_LIB"] = os.path.abspath(lib)
                out = subprocess.run([sys.executable, __file__, "--child", d, "1" if threading else "0",
                                      "1" if endings else "0"], env=env, capture_output=True, text=True)
                line = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-400:]
                print(f"{name} lib={os.path.basename(lib) or 'product'} threading={threading} endings={endings}: {line}",
                      flush=True)
