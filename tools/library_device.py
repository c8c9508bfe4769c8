#!/usr/bin/env python3
"""Library-scale job on ONE GPU with the PCM generated in HBM (needle_amd.synth.DeviceLibrary): E episodes x 45 min,
analyze + all-pairs search + epilogue through needle_hip_library_job_begin/_end, two jobs in flight.  Prints one JSON
line: ms per job, per-kernel times (HIP events), the scan's roofline (issued cell evaluations, counted in an untimed
launch, against the integer-VALU ceiling measured in the same run), the share of items the f32 first pass recomputed.
This is the program tools/profile_scan_library.sh puts behind rocprofv3.

usage: python tools/library_device.py [episodes=1000] [jobs=4] [warmup=4] [minutes=45]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from needle_amd import capi, synth  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    jobs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    warmup = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    minutes = float(sys.argv[4]) if len(sys.argv) > 4 else 45.0
    samples = int(round(minutes * 60.0 / 2 * 11025))
    t0 = time.perf_counter()
    gen = synth.DeviceLibrary(n, samples, 90.0)
    lib = capi.Library(n, opening_search_percentage=1.0)
    lib.set_pcm_device(gen.pointers(), [samples] * n)
    gen.free()
    prep_s = time.perf_counter() - t0
    cmp = capi.Comparator([f"episode-{k:05d}.wav" for k in range(n)])
    names = ["stft_chroma32", "features_cert", "stft_fallback", "fixup_items", "stft_chroma", "features_classify",
             "hamming_runs", "simhash_runs", "epilogue_buckets", "epilogue_entries", "epilogue_best_match"]
    state = {"res": None, "runs": 0}
    pending = []
    seq = [0]
    acc = {k: 0.0 for k in names}
    host = {"enqueue": 0.0, "end": 0.0}

    def step(collect):
        slot = seq[0] & 1
        seq[0] += 1
        t = time.perf_counter()
        lib.job_begin(cmp, slot)
        if collect:
            host["enqueue"] += time.perf_counter() - t
        if pending:
            t = time.perf_counter()
            state["res"], state["runs"] = lib.job_end(cmp, pending.pop())
            if collect:
                host["end"] += time.perf_counter() - t
        pending.append(slot)

    def flush():
        while pending:
            state["res"], state["runs"] = lib.job_end(cmp, pending.pop())
        capi.synchronize()

    for _ in range(warmup):
        step(False)
    flush()
    capi.cert_stats(reset=True)
    capi.set_kernel_timing("all")
    t0 = time.perf_counter()
    for _ in range(jobs):
        step(True)
        flush()                                                   # one job at a time here: its own kernels' events
        for k in names:
            acc[k] += max(capi.last_kernel_ms(k), 0.0)
    serial_ms = 1e3 * (time.perf_counter() - t0) / jobs
    capi.set_kernel_timing(None)
    t0 = time.perf_counter()
    for _ in range(jobs):
        step(True)
    flush()
    pipelined_ms = 1e3 * (time.perf_counter() - t0) / jobs
    cs = capi.cert_stats(reset=True)
    scan_form, scan_products = capi.scan_last_launch()           # of the timed jobs (the counting launches below are the vector form's)
    issued, survivors = 0, 0
    if not os.environ.get("NEEDLE_LIBRARY_DEVICE_NO_COUNT"):
        os.environ["NEEDLE_HIP_SCAN_COUNT"] = "1"
        step(False)
        flush()
        capi.scan_counts(reset=True)
        step(False)
        flush()
        issued, survivors = capi.scan_counts(reset=True)
        del os.environ["NEEDLE_HIP_SCAN_COUNT"]
    import hashlib
    runs_arr = lib.job_runs((seq[0] - 1) & 1)
    keys = np.stack([runs_arr[f].astype(np.uint32) for f in ("problem", "src_end", "dst_end", "len", "src_match_hash", "dst_match_hash")], axis=1)
    keys = keys[np.lexsort((keys[:, 2], keys[:, 1], keys[:, 0]))]
    digest = hashlib.sha256(np.ascontiguousarray(keys).tobytes()).hexdigest()[:16]
    ceiling = capi.int_valu_ceiling()
    pairs = n * (n - 1) // 2
    kept = capi.lib().needle_hip_fingerprint_num_kept(samples, 2)
    scan_ms = acc["hamming_runs"] / jobs
    out = {"episodes": n, "minutes": minutes, "pairs": pairs, "hashes_per_episode": int(kept), "prepare_s": round(prep_s, 2),
           "ms_per_job_two_in_flight": round(pipelined_ms, 3), "ms_per_job_one_at_a_time": round(serial_ms, 3),
           "pairs_per_s": round(pairs / (pipelined_ms * 1e-3), 1),
           "kernel_ms": {k: round(v / jobs, 4) for k, v in acc.items()},
           "host_ms_per_job": {k: round(1e3 * v / (2 * jobs), 3) for k, v in host.items()},
           "runs": int(state["runs"]), "run_list_digest": digest, "scan_shape": os.environ.get("NEEDLE_HIP_SCAN_SHAPE", "8,3"),
           "head_survivors": survivors, "scan_form": scan_form, "scan_matrix_products": scan_products,
           "scan_fp4_ops_per_s": round(scan_products * 131072.0 / max(scan_ms * 1e-3, 1e-12), 1),
           "detected": sum(1 for r in state["res"] if r is not None and r.opening is not None),
           "fallback": {"items": cs["items_recomputed"] / max(cs["items"], 1), "chunks": cs["chunks_recomputed"] / max(cs["chunks"], 1)},
           "scan_roofline": {"issued_cell_evaluations": issued, "lane_instructions_per_s": round(3.0 * issued / (scan_ms * 1e-3), 1),
                             "ceiling_lane_instructions_per_s": round(4.0 * ceiling, 1),
                             "frac": round(3.0 * issued / (scan_ms * 1e-3) / (4.0 * ceiling), 4),
                             "covered_table_cells": float(pairs) * kept * kept,
                             "pruning_factor": round(float(pairs) * kept * kept / max(issued, 1), 2)}}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
