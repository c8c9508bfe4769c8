#!/usr/bin/env python3
"""Regenerates tests/golden/config1.json: expected outputs of the ORACLE (oracle/*.c) for seeded synthetic
inputs.  These vectors are regression pins of this repo's oracle, not outputs of the reference (which cannot
be built or run here).  Inputs are generated, not stored: needle_amd.synth is bit-reproducible.

    python tests/golden/make_golden.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from needle_amd import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402


def main():
    eps = synth.make_library(3, 90.0, 20.0)            # BASELINE.json configs[0]
    hd = O.duration_from_secs_f32(0.3)
    fhs = O.analyze_batch([e.pcm[: len(e.pcm) // 2] for e in eps], 1, hd)
    out = {"generator": "needle_amd.synth.make_library(3, 90.0, 20.0)", "hash_duration_ns": hd,
           "pcm_crc": [int(e.pcm.astype("int64").sum()) for e in eps],
           "raw_items_first_episode": O.fingerprint(eps[0].pcm[: len(eps[0].pcm) // 2]).tolist(),
           "opening": [[[h, t] for h, t in f.opening] for f in fhs], "results": {}}
    for min_s in (20, 10):
        res = O.run_with_frame_hashes(O.Comparator(min_opening_duration=min_s * O.NS), fhs)
        out["results"][str(min_s)] = [None if r is None else list(r.opening) if r.opening else [] for r in res]
    e = eps[0]
    out["skip_file_ep0_min10"] = O.skip_file_json(
        O.run_with_frame_hashes(O.Comparator(min_opening_duration=10 * O.NS), fhs)[0], "0" * 32)
    with open(os.path.join(HERE, "config1.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("wrote config1.json", len(out["opening"][0]), "hashes per episode")


if __name__ == "__main__":
    main()
