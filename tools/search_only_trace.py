#!/usr/bin/env python3
"""bench.py's search_only leg (BASELINE.json configs[2]: 280 x 24 min from .needle.dat files) on its own, with the
library's phase trace on stderr (NEEDLE_HIP_TRACE=1): where the wall time of needle_audio_comparator_run goes."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from needle_amd import capi, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 280
print(json.dumps(bench.search_only(capi, synth, n, 24.0, reps=3)))
