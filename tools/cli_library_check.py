#!/usr/bin/env python3
"""End-to-end check of the `needle` command line on a generated library: E WAV episodes of M minutes with a shared
intro; `needle analyze` then `needle search --no-display --write-skip-files`; every episode must get a skip file
whose opening lies on the intro planted in that episode.  Usage: cli_library_check.py [episodes] [minutes] [dir]"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from needle_amd import synth  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 300
M = float(sys.argv[2]) if len(sys.argv) > 2 else 5.0
D = sys.argv[3] if len(sys.argv) > 3 else "/tmp/needle_cli_lib"
INTRO = 40.0
os.makedirs(D, exist_ok=True)
truth = {}
for k, e in enumerate(synth.make_library(E, M * 60.0, INTRO)):
    synth.write_wav(os.path.join(D, f"ep-{k:04d}.wav"), e.pcm)
    truth[f"ep-{k:04d}"] = (e.intro_off / synth.RATE, (e.intro_off + e.intro_len) / synth.RATE)
exe = os.path.join(ROOT, "needle_amd", "bin", "needle")
for cmd in (["analyze", "--force", D], ["search", "--no-display", "--write-skip-files", D]):
    t0 = time.perf_counter()
    out = subprocess.run([exe] + cmd, capture_output=True, text=True)
    print(f"needle {cmd[0]}: exit {out.returncode}, {time.perf_counter() - t0:.2f} s  {out.stderr[-200:]}")
    assert out.returncode == 0
skips = sorted(f for f in os.listdir(D) if f.endswith(".needle.skip.json"))
assert len(skips) == E, (len(skips), E)
bad = 0
for f in skips:
    op = json.load(open(os.path.join(D, f)))["opening"]
    # reported times carry chromaprint's 2.6 s delay; SURVEY.md's tolerance is +-0.25 s on top of that convention,
    # here a coarse +-4 s sanity bound against the planted position
    a, b = truth[f.split(".")[0]]
    if op is None or abs(op[0] - a) > 4.0 or abs(op[1] - b) > 4.0:
        bad += 1
print(f"{E} episodes: {len(skips)} skip files, {bad} with an opening off the planted intro; first: "
      f"{open(os.path.join(D, skips[0])).read()}")
assert bad == 0
