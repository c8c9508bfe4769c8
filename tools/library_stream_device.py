#!/usr/bin/env python3
"""BASELINE.json configs[4] as worded -- "analyze streamed from host-pinned PCM + full O(N^2) search" -- at its full
size on ONE GPU, without 59.5 GB of host memory and without 13 minutes of host synthesis: the episodes are generated in
HBM batch by batch (harness, needle_amd.synth.DeviceLibrary) and copied down into PINNED host buffers, one per episode
(the stand-in for a decoder's output; 59.5 GB at full size -- the GPU boxes have 3 TB), and only then handed to
needle_hip_library_stream_pcm batch by batch (upload on its own stream,
fingerprint kernels per landed group, nothing of the PCM kept in HBM); then the search job.  Timed: the stream_pcm calls
(PCIe-inclusive analyze) and the job (search + epilogue); generation and the copy DOWN are the harness, not the path.

usage: python tools/library_stream_device.py [episodes=2000] [batch=64] [minutes=45]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from needle_amd import capi, synth  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    minutes = float(sys.argv[3]) if len(sys.argv) > 3 else 45.0
    samples = int(round(minutes * 60.0 / 2 * 11025))
    lib = capi.Library(n, opening_search_percentage=1.0)
    cmp = capi.Comparator([f"episode-{k:05d}.wav" for k in range(n)])
    # Phase 1 (harness, untimed): every episode's PCM into its own pinned buffer -- generated in HBM batch by batch and
    # copied down.  All of it BEFORE the timed calls: generating between them (device allocations, kernels, copies down)
    # leaves the copy engine at half its rate for the calls that follow (profiles/NOTES.md, round 3: 31 GB/s instead of 53).
    t0 = time.perf_counter()
    pinned = [capi.PinnedArray(samples) for _ in range(n)]
    for first in range(0, n, batch):
        count = min(batch, n - first)
        gen = synth.DeviceLibrary(count, samples, 90.0, first_episode=first)
        for k in range(count):
            capi.check(capi.lib().needle_hip_memcpy_d2h(pinned[first + k].ptr, gen.pointers()[k], samples * 2))
        gen.free()
        print(f"[library_stream_device] {first + count}/{n} episodes generated", file=sys.stderr, flush=True)
    capi.synchronize()
    t_gen = time.perf_counter() - t0
    # Phase 2 (timed): needle_hip_library_stream_pcm batch by batch -- upload on its own stream, fingerprint kernels per
    # landed group, nothing of the PCM kept in HBM.
    lens = [samples] * n
    t_stream = 0.0
    bytes_up = 0
    per_call = []
    for first in range(0, n, batch):
        count = min(batch, n - first)
        arrays = [None] * n
        for k in range(count):
            arrays[first + k] = pinned[first + k].array
        t0 = time.perf_counter()
        lib.stream_pcm(arrays, lens)                             # H2D from pinned memory + fingerprint, overlapped
        per_call.append(time.perf_counter() - t0)
        t_stream += per_call[-1]
        bytes_up += 2 * samples * count
    capi.synchronize()
    t0 = time.perf_counter()
    lib.job_begin(cmp, 0)
    res, runs = lib.job_end(cmp, 0)
    t_first = time.perf_counter() - t0                           # grows the run slabs: scan repeated
    t0 = time.perf_counter()
    lib.job_begin(cmp, 1)
    res, runs = lib.job_end(cmp, 1)
    t_job = time.perf_counter() - t0
    pairs = n * (n - 1) // 2
    print(json.dumps({
        "episodes": n, "minutes": minutes, "pairs": pairs, "batch": batch,
        "analyze_streamed_from_pinned_s": round(t_stream, 3), "h2d_gbs": round(bytes_up / t_stream / 1e9, 2),
        "bytes_streamed": bytes_up, "h2d_gbs_per_call": [round(2 * samples * min(batch, n - k * batch) / t / 1e9, 1) for k, t in enumerate(per_call)], "search_and_epilogue_s": round(t_job, 3), "first_search_s": round(t_first, 3),
        "whole_job_s": round(t_stream + t_job, 3), "pairs_per_s": round(pairs / (t_stream + t_job), 1),
        "harness_generation_and_copy_down_s": round(t_gen, 2), "runs": int(runs),
        "detected": sum(1 for r in res if r is not None and r.opening is not None)}), flush=True)


if __name__ == "__main__":
    main()
