#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on MI355X.

metric  : episode-pairs/sec (analyze+search) = N(N-1)/2 / wall(analyze N episodes + search all pairs +
          per-video best match), N = 28 synthetic 24-min episodes (BASELINE.json configs[1]).
step    : one complete job of the hot path over the library, through the C ABI (needle_hip_library_job_begin /
          _end): fingerprint every episode's opening window (f32 first pass stft_chroma32, features_cert, f64
          recomputation of the uncertified items: stft_fallback + fixup_items), scan every
          pair (hamming_runs kernel), hash the runs (simhash_runs), download the run list, run the order-sensitive
          host epilogue (duration validity, BinaryHeap order, find_best_match).  Two jobs are in flight (job k's
          epilogue overlaps job k+1's kernels); every job completes inside the timed region.
          `value` is measured with the PCM RESIDENT IN HBM before the timed region starts (the bench contract);
          the same job from caller-owned pinned host PCM, upload overlapped with compute, is `end_to_end`.
--gpus N: one process per GPU.  Launched by torchrun (RANK / LOCAL_RANK / WORLD_SIZE in the environment) this
          process is one rank; launched plainly (`python bench.py --gpus N`) it spawns the N rank processes itself,
          before touching any GPU.  The SAME 28-episode job is sharded inside libneedle_capi.so (BASELINE.json
          configs[3]): episodes in contiguous blocks, one RCCL all-gather of hash rows, pairs in contiguous ranges,
          one RCCL all-gather of run lists, epilogue on every rank ("strong" scaling).  No torch anywhere: the
          communicator id travels through a file rendezvous (needle_amd/rendezvous.py).

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel, timed live with HIP events on the library's
stream inside the timed region (the other kernels are timed in a few untimed steps after it); `roofline_search` is
the scan kernel against the integer-VALU ceiling measured in the same run; `cpu_baseline` times the oracle (the C
restatement of the reference's CPU path, full DP table per pair, one task per episode / pair over the usable host
cores) on rank 0 at N = 1; `search_only` is BASELINE.json configs[2] (280 x 24 min from .needle.dat files).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
FP4_DENSE_PEAK_TOPS = 10000.0   # MI355X_MICROARCH.md, matrix cores: FP4 (block-scaled f8f6f4 form) runs at 4 x the BF16 rate (~2.5 PF dense)
MATRIX_OPS = 131072.0           # one v_mfma_f32_32x32x64_f8f6f4: 32 x 32 x 64 multiply-adds
F64_VALU_PEAK_TFLOPS = 78.6    # vector f64 = half the 157.3 TFLOP/s f32 vector rate of MI355X_MICROARCH.md (no faster f64 MFMA)
F32_VALU_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: peak FP32 (vector)
RATE = 11025


def algorithmic_bytes(kernel: str, windows, kept, n_pairs: int, n_runs: int) -> float:
    """analyze 2*S + 4*H per episode; search 4*(n+m) + 12*R per pair (R runs of 3 u32)."""
    if kernel == "hamming_runs":
        n = sum(kept) / max(len(kept), 1)
        return n_pairs * 4.0 * 2.0 * n + 12.0 * n_runs
    return float(sum(2 * s + 4 * h for s, h in zip(windows, kept)))


def stft_flops(windows) -> float:
    """Floating-point work of stft_chroma_kernel that no implementation can skip much of: one 4096-point complex
    FFT (5 N log2 N) per PAIR of frames (two real frames ride in one complex transform).  Window, un-mixing of the
    two spectra and |X|^2 add about 15 % and are not counted."""
    pairs = sum(((max(s - 4096, -1365) // 1365 + 1) + 1) // 2 for s in windows)
    return pairs * 5.0 * 4096 * 12


def usable_cpus() -> int:
    """Host CPUs this process may actually use: its affinity mask, capped by the cgroup CPU quota (a container can
    see every core of the machine and still be throttled to a few CPUs' worth of time; more threads than that
    only add throttling stalls)."""
    n = max(1, len(os.sched_getaffinity(0)))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]           # cgroup v2
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())          # cgroup v1
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, quota // period))
        except (OSError, ValueError):
            pass
    return n


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


class DeviceTelemetry:
    """Engine clock, temperature and power of the bench's device, read from its sysfs node while the timed region
    runs (a thread, one read per ~2 ms; nothing is added to the GPU's queue).  The driver times a region of a few tens
    of milliseconds right after a short warm-up, so the clock the kernels actually ran at belongs in the line: a
    device that is still ramping up, or one that is power-capped under sustained f64 load, runs the same code slower."""

    def __init__(self, capi):
        import glob
        import threading
        self.node, self.samples, self._stop, self._thread = None, [], threading.Event(), None
        self.source = "unavailable"
        try:
            node = os.path.join("/sys/bus/pci/devices", capi.device_pci_bus_id().lower())
            if self._sclk(os.path.join(node, "pp_dpm_sclk")) is not None:
                self.node, self.source = node, "sysfs pp_dpm_sclk, sampled inside the timed region"
                hw = glob.glob(os.path.join(node, "hwmon", "hwmon*"))
                self.hwmon = hw[0] if hw else None
        except Exception:                                        # noqa: BLE001 -- diagnostics must never fail a run
            self.node = None
        self._threading = threading

    @staticmethod
    def _sclk(path):
        try:
            for line in open(path):
                if "*" in line:
                    return float(line.split(":")[1].lower().replace("mhz", "").replace("*", "").strip())
        except (OSError, ValueError, IndexError):
            pass
        return None

    def _read_int(self, name):
        try:
            return int(open(os.path.join(self.hwmon, name)).read())
        except (OSError, ValueError, TypeError):
            return None

    def start(self):
        if not self.node:
            return
        self.samples, path = [], os.path.join(self.node, "pp_dpm_sclk")
        self._stop.clear()

        def run():
            while not self._stop.is_set():
                v = self._sclk(path)
                if v is not None:
                    self.samples.append(v)
                time.sleep(0.002)
        self._thread = self._threading.Thread(target=run, daemon=True)
        self._thread.start()

    def stop(self):
        out = {"sclk_mhz": None, "sclk_mhz_min": None, "sclk_mhz_max": None, "temperature_c": None, "power_w": None,
               "samples": 0, "source": self.source}
        if self._thread is not None:
            self._stop.set()
            self._thread.join()
            self._thread = None
        if self.node and self.samples:
            s = self.samples
            out.update(sclk_mhz=round(sum(s) / len(s), 1), sclk_mhz_min=min(s), sclk_mhz_max=max(s), samples=len(s))
            t = max((v for v in (self._read_int(f"temp{i}_input") for i in (1, 2, 3)) if v is not None), default=None)
            p = self._read_int("power1_average") or self._read_int("power1_input")
            out["temperature_c"] = None if t is None else round(t / 1000.0, 1)
            out["power_w"] = None if p is None else round(p / 1e6, 1)
            return out
        try:                                                     # no readable sysfs node: one sample right after the region
            r = subprocess.run(["rocm-smi", "--showclocks", "--showtemp", "--showpower", "--json"], capture_output=True,
                               text=True, timeout=20)
            card = next(iter(json.loads(r.stdout).values()))
            for k, v in card.items():
                lk = k.lower()
                if "sclk clock speed" in lk:
                    out["sclk_mhz"] = float(str(v).strip("()").lower().replace("mhz", ""))
                elif "temperature" in lk and "junction" in lk:
                    out["temperature_c"] = float(v)
                elif "power" in lk and "socket" in lk:
                    out["power_w"] = float(v)
            out["source"] = "rocm-smi, ONE sample taken right after the timed region (the clock may already have dropped)"
        except Exception:                                        # noqa: BLE001
            pass
        return out


def cpu_baseline(windows, n_total, results_gpu, hashes_gpu, whole_job):
    """The oracle on the host cores (rank 0, N = 1 only): same episodes, same pairs, reference cost structure.
    `windows`: the opening-window PCM of a SAMPLE of the job's episodes (all of them when whole_job), n_total the job's
    episode count.  Bounded to ~30 s: analyze the sample, search as many of its pairs as the budget allows, scale both
    legs to the whole job.  Cross-checks the GPU's hashes of the sampled episodes (and, when the whole job was run,
    its results) against the oracle's."""
    import numpy as np
    from oracle import oracle as O
    threads = usable_cpus()
    hd = O.duration_from_secs_f32(0.3)
    t0 = time.perf_counter()
    fhs = O.analyze_batch(windows, 1, hd, threads=threads)
    t_analyze = (time.perf_counter() - t0) * n_total / len(windows)
    n = len(windows)
    pairs_total = n_total * (n_total - 1) // 2
    # ~0.1 core-seconds per 24-min pair: keep the search leg under ~25 s of wall
    est_pair_s = 0.1 * (len(fhs[0].opening) / 2897.0) ** 2
    budget_pairs = int(25.0 * threads / max(est_pair_s, 1e-6))
    k = n
    while k > 2 and k * (k - 1) // 2 > budget_pairs:
        k -= 1
    t0 = time.perf_counter()
    res = O.run_with_frame_hashes(O.Comparator(), fhs[:k], threads=threads)
    t_search = time.perf_counter() - t0
    sample_pairs = k * (k - 1) // 2
    t_search_full = t_search * pairs_total / max(sample_pairs, 1)
    value = pairs_total / (t_analyze + t_search_full)
    # second CPU number, so the ratio is not inflated by the reference's allocation pattern: the
    # same scan without the table, one pass per diagonal, all sampled pairs over all threads; same analyze stage
    seqs = [np.array([h for h, _ in f.opening], dtype=np.uint32) for f in fhs]
    t0 = time.perf_counter()
    opt_runs, _ = O.diagonal_runs_all_pairs(seqs, 10, 82, threads=threads)
    t_opt = (time.perf_counter() - t0) * pairs_total / max(n * (n - 1) // 2, 1)
    parity = all(hashes_gpu[v].tolist() == [h for h, _ in fhs[v].opening] for v in range(n))
    if whole_job and k == n:
        got = [None if r is None else (r.opening, r.ending) for r in results_gpu]
        want = [None if r is None else (r.opening, r.ending) for r in res]
        parity = parity and got == want
    return {
        "value": round(value, 3), "unit": "episode-pairs/s", "cores": threads, "kind": "port",
        "cpu_model": cpu_model(),
        "sample": f"oracle (C restatement of analyzer.rs/comparator.rs + chromaprint in f64, not the Rust binary): "
                  f"analyze {n} of {n_total} episodes, search {sample_pairs} of {pairs_total} pairs in "
                  f"{t_search:.2f} s (both legs scaled to the whole job), {threads} threads, PCM in host memory",
        "analyze_s_scaled": round(t_analyze, 3), "search_s_scaled": round(t_search_full, 3),
        "gpu_matches_oracle": bool(parity),
        "optimised_cpu_variant": {"value": round(pairs_total / (t_analyze + t_opt), 3), "unit": "episode-pairs/s",
                                  "search_s_scaled": round(t_opt, 4), "runs_in_sample": int(opt_runs),
                                  "what": "same analyze stage + table-free diagonal scan of the sampled pairs (min run 82), "
                                          f"{threads} threads, scaled, without the per-video epilogue"},
    }


def end_to_end(capi, eps, cmp, n_pairs, reps=6):
    """The same job from caller-owned host PCM (what a decoder hands over): needle_hip_library_stream_pcm uploads the
    opening windows on an upload stream and fingerprints them group by group as they land, then search + epilogue.
    Measured from pinned memory (read in place by the copy engine) and from pageable numpy arrays (ring of pinned
    slabs filled by host threads).  PCIe-inclusive; never `value`."""
    n = len(eps)
    lens = [len(e.pcm) for e in eps]
    window_bytes = sum(2 * (v // 2) for v in lens)
    out = {}
    pinned = [capi.PinnedArray(v) for v in lens]
    for p, e in zip(pinned, eps):
        p.array[:] = e.pcm
    for label, arrays in (("pinned", [p.array for p in pinned]), ("pageable", [e.pcm for e in eps])):
        lib = capi.Library(n)
        jobs, uploads = [], []
        res = None
        for rep in range(reps + 2):
            capi.synchronize()
            t0 = time.perf_counter()
            lib.stream_pcm(arrays, lens)
            t1 = time.perf_counter()
            lib.job_begin(cmp, 0)
            res, _ = lib.job_end(cmp, 0)
            t2 = time.perf_counter()
            if rep >= 2:
                jobs.append(t2 - t0)
                uploads.append(t1 - t0)
        ms = 1e3 * sum(jobs) / len(jobs)
        out[label] = {"ms_per_job": round(ms, 3), "best_ms": round(1e3 * min(jobs), 3),
                      "pairs_per_s": round(n_pairs / (ms * 1e-3), 1),
                      "h2d_gbs": round(window_bytes / (sum(uploads) / len(uploads)) / 1e9, 2),
                      "detected": sum(1 for r in res if r is not None and r.opening is not None)}
        del lib
    out["bytes_uploaded_per_job"] = window_bytes
    out["what"] = ("needle_hip_library_stream_pcm (upload on its own stream, fingerprint kernels per ~32 MiB group "
                   "behind events) + job_begin/_end (search, download, epilogue); one job at a time, wall clock")
    return out


def search_roofline(ceiling_cells, issued_evals, covered_cells, scan_ms):
    """The scan kernel against the integer-VALU roof.  ceiling_cells: needle_hip_int_valu_ceiling(), cells/s of a
    4-instruction cell (xor, popcount, compare, select) on registers, measured in this run.  issued_evals: cell
    evaluations the scan ISSUES per launch (3 instructions each; needle_hip_scan_issued_evaluations, counted in an
    untimed launch of the same work).  frac = issued lane-instructions/s over the ceiling's lane-instructions/s: <= 1
    by construction.  What the aligned-window scan saves by NOT evaluating cells is `pruning_factor`, kept apart."""
    if not issued_evals or scan_ms <= 0:
        return {"error": "no count from the counting launch"}
    sec = scan_ms * 1e-3
    achieved, peak = 3.0 * issued_evals / sec, 4.0 * ceiling_cells
    return {"bound": "int valu", "kernel": "hamming_runs", "unit": "lane-instructions/s",
            "achieved": round(achieved, 1), "peak": round(peak, 1), "frac": round(achieved / peak, 4),
            "issued_cell_evaluations_per_launch": int(issued_evals), "avg_launch_ms": round(scan_ms, 5),
            "ceiling_cells_per_s": round(ceiling_cells, 1),
            "covered_table_cells_per_launch": covered_cells,
            "covered_table_cells_per_s": round(covered_cells / sec, 1),
            "pruning_factor": round(covered_cells / issued_evals, 2),
            "note": "achieved = 3 (xor, popcount, compare) x the cell evaluations the scan issues, padding and repeated lanes "
                    "included, / kernel time; peak = 4 x the measured rate of the 4-instruction cell on registers.  "
                    "pruning_factor = cells of the reference's table covered per evaluation issued: algorithmic skipping, "
                    "not machine utilisation"}


def search_only(capi, synth, episodes, minutes, reps=10, hostile=False):
    """BASELINE.json configs[2]: `episodes` x 24-min episodes as real .needle.dat files (written once by this
    analyzer from synthetic audio), then needle_audio_comparator_run(analyze=false) timed from disk: file reads,
    upload of the hashes, scan + simhash + epilogue kernels, download of the results."""
    t_prep = time.perf_counter()
    half = minutes * 60.0 / 2
    tmp = tempfile.mkdtemp(prefix="needle_bench_search_")
    paths = [os.path.join(tmp, f"episode-{k:04d}.wav") for k in range(episodes)]
    # Only the opening half of each episode is ever hashed: that half is generated in HBM (synth.DeviceLibrary),
    # fingerprinted there, and every video's FrameHashes is written with the product's own writer
    # (needle_hip_library_frame_hashes + needle_hip_frame_hashes_write -> <video>.needle.dat, data.rs layout).
    samples = int(round(half * RATE))
    gen = synth.DeviceLibrary(episodes, samples, 90.0 if half > 400 else half / 4, hostile=hostile)
    src = capi.Library(episodes, opening_search_percentage=1.0)
    src.set_pcm_device(gen.pointers(), [samples] * episodes)
    gen.free()
    src.analyze(0, episodes, sync=True)
    for k, p in enumerate(paths):
        src.frame_hashes(k).write(os.path.splitext(p)[0] + ".needle.dat")
    del src, gen
    prep_s = time.perf_counter() - t_prep
    cmp = capi.Comparator(paths)
    capi.set_kernel_timing("hamming_runs,simhash_runs")
    capi.epilogue_host_fallbacks(reset=True)
    walls, scan, simh = [], [], []
    for rep in range(reps + 1):
        t0 = time.perf_counter()
        cmp.run(analyze=False, display=False)
        dt = time.perf_counter() - t0
        if rep:
            walls.append(dt)
            scan.append(capi.last_kernel_ms("hamming_runs"))
            simh.append(capi.last_kernel_ms("simhash_runs"))
    capi.set_kernel_timing(None)
    scan_form, scan_products = capi.scan_last_launch()           # of the timed calls (the counting launch is the vector form's)
    os.environ["NEEDLE_HIP_SCAN_COUNT"] = "1"                   # one more call through the counting scan (untimed)
    try:
        cmp.run(analyze=False, display=False)
        capi.scan_issued_evaluations(reset=True)
        cmp.run(analyze=False, display=False)
        issued = capi.scan_issued_evaluations(reset=True)
    finally:
        del os.environ["NEEDLE_HIP_SCAN_COUNT"]
    for p in paths:
        try:
            os.unlink(os.path.splitext(p)[0] + ".needle.dat")
        except OSError:
            pass
    try:
        os.rmdir(tmp)
    except OSError:
        pass
    pairs = episodes * (episodes - 1) // 2
    n_h = capi.lib().needle_hip_fingerprint_num_kept(int(round(half * RATE)), 2)
    wall = sorted(walls)[len(walls) // 2]                         # the median call (the first timed one still warms caches)
    return {"episodes": episodes, "pairs": pairs, "hashes_per_episode": int(n_h),
            "wall_ms": round(1e3 * wall, 3), "wall_ms_mean": round(1e3 * sum(walls) / len(walls), 3), "wall_ms_each": [round(1e3 * w, 3) for w in walls], "wall_ms_best": round(1e3 * min(walls), 3), "pairs_per_s": round(pairs / wall, 1),
            "scan_kernel_ms": round(sum(scan) / len(scan), 4), "simhash_kernel_ms": round(sum(simh) / len(simh), 4),
            "table_cells": float(pairs) * n_h * n_h, "prepare_s": round(prep_s, 2), "issued_evals": issued,
            "scan_form": scan_form, "matrix_instructions": scan_products,
            "corpus": "hostile" if hostile else "tonal", "epilogue_host_fallbacks": capi.epilogue_host_fallbacks(),
            "what": "needle_audio_comparator_run(analyze=false) over .needle.dat files in the page cache: read + parse, "
                    "H2D of hashes, scan, simhash, per-video epilogue on the device (from 16 384 sequence pairs up; the run "
                    "list stays in HBM), D2H of the results; wall clock per call"}


def library_scale(capi, synth, episodes, minutes, jobs=3, check=4):
    """BASELINE.json configs[4]'s SHAPE inside the default run: `episodes` x `minutes` (2000 x 45 is the configuration
    itself; the default 1000 x 45 is 29.8 GB of PCM in HBM and fits the driver's run), PCM generated in HBM, analyze + full
    O(N^2) search + per-video epilogue, two jobs in flight as in the headline.  Reports ms per job, pairs/s, the kernels of a
    job run alone, which form the scan took and its roofline on the pipe that bounds it; the GPU's hashes of `check` episodes
    are compared with the oracle's OUTSIDE the timed region."""
    import numpy as np
    from oracle import oracle as O
    t_prep = time.perf_counter()
    samples = int(round(minutes * 60.0 / 2 * RATE))
    gen = synth.DeviceLibrary(episodes, samples, 90.0)
    ids = sorted({0, episodes // 3, (2 * episodes) // 3, episodes - 1})[:check]
    sample_pcm = {k: gen.episode(k) for k in ids}
    lib = capi.Library(episodes, opening_search_percentage=1.0)
    lib.set_pcm_device(gen.pointers(), [samples] * episodes)
    gen.free()
    del gen
    cmp = capi.Comparator([f"episode-{k:05d}.wav" for k in range(episodes)])
    prep_s = time.perf_counter() - t_prep
    names = ["stft_chroma32", "features_cert", "stft_fallback", "fixup_items", "hamming_runs", "simhash_runs",
             "epilogue_buckets", "epilogue_entries", "epilogue_best_match"]
    state = {"res": None, "runs": 0}
    pending, seq = [], [0]

    def step():
        slot = seq[0] & 1
        seq[0] += 1
        lib.job_begin(cmp, slot)
        if pending:
            state["res"], state["runs"] = lib.job_end(cmp, pending.pop())
        pending.append(slot)

    def flush():
        while pending:
            state["res"], state["runs"] = lib.job_end(cmp, pending.pop())
        capi.synchronize()

    step()
    flush()                                                      # first job: slabs grow, tables are built
    step()
    flush()
    capi.set_kernel_timing("all,sum")                            # a kernel's time = the SUM over the job's launches of it
    t0 = time.perf_counter()
    step()
    flush()                                                      # one job alone: its own kernels' events, and the latency
    alone_ms = 1e3 * (time.perf_counter() - t0)
    kernel_ms = {k: round(max(capi.last_kernel_ms(k), 0.0), 4) for k in names}
    capi.set_kernel_timing(None)
    t0 = time.perf_counter()
    for _ in range(jobs):
        step()
    flush()
    ms = 1e3 * (time.perf_counter() - t0) / jobs
    form, products = capi.scan_last_launch()
    pairs = episodes * (episodes - 1) // 2
    hd = O.duration_from_secs_f32(0.3)
    want = O.analyze_batch([sample_pcm[k] for k in ids], 1, hd, threads=usable_cpus())
    hashes_ok = all(lib.frame_hashes(k).opening_data()[0].tolist() == [h for h, _ in w.opening] for k, w in zip(ids, want))
    scan_s = kernel_ms["hamming_runs"] * 1e-3
    out = {"episodes": episodes, "minutes": minutes, "pairs": pairs,
           "hashes_per_episode": int(capi.lib().needle_hip_fingerprint_num_kept(samples, 2)),
           "pcm_bytes_in_hbm": 2 * samples * episodes, "prepare_s": round(prep_s, 2), "jobs_timed": jobs,
           "ms_per_job": round(ms, 3), "pairs_per_s": round(pairs / (ms * 1e-3), 1), "latency_ms_one_job": round(alone_ms, 3),
           "kernel_ms_one_job_alone": kernel_ms, "runs_per_job": int(state["runs"]),
           "detected": sum(1 for r in state["res"] if r is not None and r.opening is not None),
           "scan_form": {1: "generic", 2: "band", 3: "aligned windows, vector ALU", 4: "aligned windows, matrix pipe"}.get(form, str(form)),
           "gpu_hashes_match_oracle": {"episodes_checked": ids, "ok": bool(hashes_ok)},
           "what": "BASELINE.json configs[4]'s shape (2000 x 45 min is the configuration itself: --episodes 2000 --minutes 45 "
                   "--device-synth): PCM generated in HBM, analyze + all-pairs search + per-video epilogue through "
                   "needle_hip_library_job_begin/_end, two jobs in flight; kernel times from one job run alone"}
    if form == 4 and scan_s > 0:
        tops = products * MATRIX_OPS / scan_s / 1e12
        out["roofline"] = {"bound": "mfma", "kernel": "hamming_runs (aligned windows, first stage on the matrix pipe)", "unit": "TOP/s",
                           "achieved": round(tops, 1), "peak": FP4_DENSE_PEAK_TOPS, "frac": round(tops / FP4_DENSE_PEAK_TOPS, 4),
                           "matrix_instructions_per_launch": int(products), "avg_launch_ms": kernel_ms["hamming_runs"],
                           "note": "v_mfma_f32_32x32x64_f8f6f4 (FP4 operands) instructions the launch issues x 131 072 operations / kernel time "
                                   "against the dense FP4 peak of MI355X_MICROARCH.md"}
        try:                                                     # SQ_VALU_MFMA_BUSY_CYCLES of an earlier rocprofv3 --pmc run of this kernel
            cj = json.load(open(os.path.join(ROOT, "profiles", "scan_mfma_counters.json")))
            out["roofline"].update(pipe_busy=cj.get("pipe_busy"), valu_issue_frac=cj.get("valu_issue_frac"),
                                   vector_per_matrix_instruction=cj.get("vector_per_matrix_instruction"), pipe_busy_source=cj.get("source"))
        except (OSError, ValueError):
            pass
    return out


def corpus_hostile(capi, synth, episodes, minutes, jobs=20, check=2):
    """The headline job on the HOSTILE corpus (needle_amd/csrc/synth_hip.hip: broadband speech-like bodies, noise 20 dB under
    the programme, 25 - 60 s of digital silence in every second episode and of one sustained chord in every third, a tonal
    shared intro): the reference's cost does not depend on content (comparator.rs:176-187 visits every cell), this build's
    does -- the first pass certifies fewer items, constant hashes defeat the scan's aligned-window filter and fill a pair's
    bucket with hundreds of runs.  Same shape and call sequence as the headline (two jobs in flight); the GPU's hashes of
    `check` episodes are compared with the oracle's outside the timed region (the whole job against the oracle:
    tests/test_gpu_hostile.py and `bench.py --corpus hostile`)."""
    from oracle import oracle as O
    samples = int(round(minutes * 60.0 / 2 * RATE))
    gen = synth.DeviceLibrary(episodes, samples, 90.0 if minutes >= 10 else minutes * 60.0 / 8, hostile=True)
    ids = sorted({0, episodes // 2, episodes - 1})[:check]
    sample_pcm = {k: gen.episode(k) for k in ids}
    segments = gen.segments.tolist()
    lib = capi.Library(episodes, opening_search_percentage=1.0)
    lib.set_pcm_device(gen.pointers(), [samples] * episodes)
    gen.free()
    del gen
    cmp = capi.Comparator([f"episode-{k:05d}.wav" for k in range(episodes)])
    names = ["stft_chroma32", "features_cert", "stft_fallback", "fixup_items", "hamming_runs", "simhash_runs",
             "epilogue_buckets", "epilogue_entries", "epilogue_best_match"]
    state = {"res": None, "runs": 0}
    pending, seq = [], [0]

    def step():
        slot = seq[0] & 1
        seq[0] += 1
        lib.job_begin(cmp, slot)
        if pending:
            state["res"], state["runs"] = lib.job_end(cmp, pending.pop())
        pending.append(slot)

    def flush():
        while pending:
            state["res"], state["runs"] = lib.job_end(cmp, pending.pop())
        capi.synchronize()

    for _ in range(3):
        step()
    flush()
    capi.cert_stats(reset=True)
    capi.epilogue_host_fallbacks(reset=True)
    capi.set_kernel_timing("all,sum")
    t0 = time.perf_counter()
    step()
    flush()
    alone_ms = 1e3 * (time.perf_counter() - t0)
    kernel_ms = {k: round(max(capi.last_kernel_ms(k), 0.0), 4) for k in names}
    capi.set_kernel_timing(None)
    cs = capi.cert_stats(reset=True)
    for _ in range(10):                                          # back to the steady state of two jobs in flight
        step()
    flush()
    t0 = time.perf_counter()
    for _ in range(jobs):
        step()
    flush()
    ms = 1e3 * (time.perf_counter() - t0) / jobs
    form, products = capi.scan_last_launch()
    os.environ["NEEDLE_HIP_SCAN_COUNT"] = "1"                   # what the vector form ISSUES on this content (untimed)
    try:
        step()
        flush()
        capi.scan_counts(reset=True)
        step()
        flush()
        issued, survivors = capi.scan_counts(reset=True)
    finally:
        del os.environ["NEEDLE_HIP_SCAN_COUNT"]
    pairs = episodes * (episodes - 1) // 2
    n_h = int(capi.lib().needle_hip_fingerprint_num_kept(samples, 2))
    hd = O.duration_from_secs_f32(0.3)
    want = O.analyze_batch([sample_pcm[k] for k in ids], 1, hd, threads=usable_cpus())
    hashes_ok = all(lib.frame_hashes(k).opening_data()[0].tolist() == [h for h, _ in w.opening] for k, w in zip(ids, want))
    return {"episodes": episodes, "minutes": minutes, "pairs": pairs, "hashes_per_episode": n_h, "jobs_timed": jobs,
            "ms_per_step": round(ms, 4), "pairs_per_s": round(pairs / (ms * 1e-3), 1), "latency_ms_one_job": round(alone_ms, 3),
            "kernel_ms_one_job_alone": kernel_ms, "runs_per_step": int(state["runs"]),
            "detected": sum(1 for r in state["res"] if r is not None and r.opening is not None),
            "fallback_frac": {"items": round(cs["items_recomputed"] / max(cs["items"], 1), 6),
                              "frame_pair_chunks": round(cs["chunks_recomputed"] / max(cs["chunks"], 1), 6)},
            "scan_form": {1: "generic", 2: "band", 3: "aligned windows, vector ALU", 4: "aligned windows, matrix pipe"}.get(form, str(form)),
            "scan_issued_evaluations": int(issued), "scan_head_survivors": int(survivors),
            "scan_pruning_factor": round(float(pairs) * n_h * n_h / max(issued, 1), 2),
            "epilogue_host_fallbacks": capi.epilogue_host_fallbacks(), "job_form": lib.job_form(0),
            "episodes_with_silence": sum(1 for s in segments if s[1] > 0), "episodes_with_chord": sum(1 for s in segments if s[3] > 0),
            "gpu_hashes_match_oracle": {"episodes_checked": ids, "ok": bool(hashes_ok)},
            "what": "the headline job (28 x 24 min, analyze + search, two jobs in flight, PCM resident) on the hostile corpus; "
                    "kernel times from one job run alone; results against the oracle: tests/test_gpu_hostile.py, bench.py --corpus hostile"}


# ---- launching N ranks -----------------------------------------------------------------------------------------------
# Every rank of an N > 1 run is TWO processes: a supervisor that never touches a GPU and the worker it starts.  A
# collective that never completes (the first multi-rank RCCL bring-up happens on the driver's clock) cannot be
# cancelled from inside the process that is stuck in it, so the supervisors do it from outside: rank 0's supervisor
# gives the workers `--launch-timeout` seconds to finish; if they do not, or one dies, it tells every supervisor
# (through a file rendezvous of their own) to kill its worker and start it once more over the host-staged transport
# (NEEDLE_HIP_COMM=host: shared memory, nothing from the interconnect).  The result line is written by worker 0 into
# the supervisors' directory and printed by supervisor 0 only when an attempt has succeeded, so exactly one JSON line
# comes out whatever happened on the way.  Under torchrun (the driver's form) each launched process is a supervisor;
# `python bench.py --gpus N` without a launcher's environment starts the N supervisors itself.
def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _kill(proc) -> None:
    if proc is not None and proc.poll() is None:
        proc.kill()                                              # by PID: this process started it
    if proc is not None:
        proc.wait()


def supervise_rank(rank: int, world: int, timeout_s: float) -> int:
    from needle_amd import rendezvous                            # file exchange only: nothing here touches a GPU
    sup = rendezvous.FileRendezvous(rank, world, timeout_s=max(120.0, timeout_s), role="sup")
    backend0 = os.environ.get("NEEDLE_HIP_COMM", "rccl")
    proc = None
    try:
        for attempt in range(2):
            env = dict(os.environ, NEEDLE_BENCH_WORKER="1", NEEDLE_BENCH_SUP_DIR=sup.dir, NEEDLE_BENCH_ATTEMPT=str(attempt),
                       NEEDLE_RDZV_NONCE=f"{os.path.basename(sup.dir)}-a{attempt}")
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            if attempt:
                env["NEEDLE_HIP_COMM"] = "host"
            proc = subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                    stdout=sys.stderr)           # the JSON line travels through the directory
            t_start = time.monotonic()
            decision = None
            reported = False
            while decision is None:
                code = proc.poll()
                if code is not None and not reported:
                    sup.set(f"exit.{attempt}.{rank}", str(code).encode())
                    reported = True
                if rank == 0:
                    codes = [sup.try_get(f"exit.{attempt}.{r}") for r in range(world)]
                    finished = [sup.try_get(f"finished.{attempt}.{r}") is not None for r in range(world)]
                    if all(finished):
                        decision = b"done"
                    elif any(c is not None and c != b"0" for c in codes):
                        decision = b"retry:a rank failed"
                    elif time.monotonic() - t_start > timeout_s:
                        stuck = [r for r in range(world) if not finished[r]]
                        decision = f"retry:ranks {stuck} did not finish within {timeout_s:g} s".encode()
                    if decision is not None:
                        sup.set(f"decision.{attempt}", decision)
                else:
                    decision = sup.try_get(f"decision.{attempt}")
                    if decision is None and time.monotonic() - t_start > timeout_s + 60.0:
                        print(f"[bench] supervisor {rank}: no decision from supervisor 0; giving up", file=sys.stderr)
                        return 1
                if decision is None:
                    time.sleep(0.02)
            if decision == b"done":
                try:
                    proc.wait(timeout=20.0)                      # communicator teardown; nothing depends on it
                except subprocess.TimeoutExpired:
                    pass
                _kill(proc)
                if rank == 0:
                    sys.stdout.write(sup.get(f"result.{attempt}", timeout_s=5.0).decode())
                    sys.stdout.flush()
                sup.barrier("bye")
                return 0
            _kill(proc)
            why = decision.decode().split(":", 1)[-1]
            last = attempt == 1 or backend0 == "host"
            if rank == 0:
                print(f"[bench] attempt {attempt} over {'host' if attempt else backend0}: {why}; "
                      f"{'giving up' if last else 'every rank restarts over the host-staged transport (NEEDLE_HIP_COMM=host)'}",
                      file=sys.stderr, flush=True)
            if last:
                return 1
            sup.barrier(f"killed.{attempt}")                     # nobody of attempt 0 is alive when attempt 1 starts
        return 1
    finally:
        _kill(proc)
        if rank == 0:
            time.sleep(0.2)
            sup.remove()


def spawn_ranks(n: int, timeout_s: float) -> int:
    """`python bench.py --gpus N` without a launcher's environment: starts the N per-rank supervisors (this process
    never touches a GPU) and waits for them; they carry the timeout and the retry."""
    env = dict(os.environ, WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
                              env=dict(env, RANK=str(rank), LOCAL_RANK=str(rank)),
                              stdout=None if rank == 0 else sys.stderr) for rank in range(n)]
    deadline = time.monotonic() + 2.0 * timeout_s + 180.0
    try:
        while any(p.poll() is None for p in procs) and time.monotonic() < deadline:
            time.sleep(0.05)
    finally:
        for p in procs:
            _kill(p)
    return 0 if all(p.returncode == 0 for p in procs) else 1


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--episodes", type=int, default=28)
    ap.add_argument("--minutes", type=float, default=24.0)
    ap.add_argument("--intro-seconds", type=float, default=90.0)
    ap.add_argument("--preheat", type=int, default=80,
                    help="untimed jobs run BEFORE the W warm-up steps: a step is 0.6 ms, so 5 warm-up steps are 3 ms, and the "
                         "device needs ~30 ms of load before its kernels run at their steady rate (DESIGN.md section 5)")
    ap.add_argument("--device-synth", action="store_true",
                    help="generate every rank's episodes in HBM (needle_amd.synth.DeviceLibrary) instead of on the host: what "
                         "makes BASELINE.json configs[4] (--episodes 2000 --minutes 45) fit a bench run; the opening half of "
                         "each episode is generated and is the whole search window")
    ap.add_argument("--corpus", choices=["tonal", "hostile"], default="tonal",
                    help="hostile: the whole run (value, roofline, cpu_baseline, parity check) on the hostile corpus of "
                         "needle_amd/csrc/synth_hip.hip (generated in HBM: implies --device-synth)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip end_to_end / search_only / roofline_search")
    ap.add_argument("--search-only-episodes", type=int, default=280)
    ap.add_argument("--library-scale-episodes", type=int, default=1000,
                    help="the library_scale leg of the default run: this many x 45 min generated in HBM (0 = skip)")
    ap.add_argument("--no-live-events", action="store_true",
                    help="diagnostic: no HIP events around the dominant kernel inside the timed region (what they cost)")
    ap.add_argument("--force-comm", action="store_true", help="create a 1-rank communicator even at N=1")
    ap.add_argument("--launch-timeout", type=float, default=120.0,
                    help="N > 1: seconds the ranks of one attempt get before they are killed and restarted over the "
                         "host-staged transport (an 8-rank run takes ~15 s)")
    ap.add_argument("--selftest-worker", choices=["ok", "hang", "fail"], help=argparse.SUPPRESS)
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(spawn_ranks(args.gpus, args.launch_timeout))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and os.environ.get("NEEDLE_BENCH_WORKER") != "1":
        raise SystemExit(supervise_rank(rank, world, args.launch_timeout))
    sup_dir, attempt = os.environ.get("NEEDLE_BENCH_SUP_DIR"), os.environ.get("NEEDLE_BENCH_ATTEMPT", "0")
    if args.selftest_worker:                                     # tests/test_bench_launcher_cpu.py: the protocol, no GPU
        if args.selftest_worker != "ok" and os.environ.get("NEEDLE_HIP_COMM") != "host":
            if args.selftest_worker == "fail" and rank == world - 1:
                raise SystemExit(3)
            time.sleep(3600)                                     # "a collective that never completes"
        if rank == 0:
            with open(os.path.join(sup_dir, f"result.{attempt}"), "w") as f:
                f.write(json.dumps({"selftest": args.selftest_worker, "n_gpus": world,
                                    "comm": os.environ.get("NEEDLE_HIP_COMM", "rccl"), "attempt": int(attempt)}) + "\n")
        open(os.path.join(sup_dir, f"finished.{attempt}.{rank}"), "w").close()
        return
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    import numpy as np
    from needle_amd import capi, rendezvous, synth

    if capi.device_count() < 1:
        raise SystemExit("bench.py needs a HIP device: the needle path has no CPU fallback")
    rdzv = None
    if world > 1 or args.force_comm:
        # workers of one attempt share the nonce their supervisors exported (their parents differ)
        rdzv = rendezvous.init_comm(capi, rank, world, local_rank % capi.device_count(),
                                    key=os.environ.get("NEEDLE_RDZV_NONCE") if sup_dir else None)
        print(f"[bench] rank {rank}/{world}: communicator up over {capi.comm_backend()} "
              f"(device {local_rank % capi.device_count()})", file=sys.stderr, flush=True)
    else:
        capi.set_device(0)

    n = args.episodes
    total_samples = int(round(args.minutes * 60.0 * RATE))
    # every rank needs every length (metadata), but only the PCM its share of the fingerprinting depends on: the hashes
    # of all episodes are cut into `world` equal blocks (3.5 episodes' worth each for 28 on 8), and rank_videos names
    # the episodes a rank's block meets
    t_setup = time.perf_counter()
    eps = None
    if args.corpus == "hostile":
        args.device_synth = True
    if args.device_synth:
        window_samples = int(round(args.minutes * 60.0 / 2 * RATE))
        lib = capi.Library(n, opening_search_percentage=1.0)
        first, count = lib.rank_videos([window_samples] * n, world, rank)
        gen = synth.DeviceLibrary(count, window_samples, args.intro_seconds, first_episode=first, hostile=args.corpus == "hostile")
        ptrs = gen.pointers()
        # a bounded sample for the CPU baseline / cross-check, read back before the generator's buffer goes
        sample_ids = list(range(min(n, 40))) if (world == 1 and not args.no_cpu_baseline) else []
        sample_pcm = [gen.episode(k) for k in sample_ids]
        lib.set_pcm_device([ptrs[k - first] if first <= k < first + count else None for k in range(n)], [window_samples] * n)
        gen.free()
        del gen
        windows = [window_samples] * n
    else:
        lib = capi.Library(n)
        first, count = lib.rank_videos([total_samples] * n, world, rank)
        mine = {k: synth.make_episode(k, args.minutes * 60.0, args.intro_seconds) for k in range(first, first + count)}
        if world == 1:
            eps = [mine[k] for k in range(n)]
        lib.set_pcm([mine[k].pcm if k in mine else None for k in range(n)], [total_samples] * n)
        windows = [total_samples // 2] * n
    setup_s = time.perf_counter() - t_setup
    cmp = capi.Comparator([f"episode-{k:04d}.wav" for k in range(n)])
    cmp.handle()
    n_pairs = lib.num_pairs()
    kept = [capi.lib().needle_hip_fingerprint_num_kept(w, 2) for w in windows]

    # default arithmetic: f32 first pass + certification + f64 recomputation of what could not be certified (include/
    # needle_hip.h); NEEDLE_HIP_STFT=f64 runs stft_chroma + features_classify instead.  Kernels that did not run read 0.
    kernel_names = ["stft_chroma32", "features_cert", "stft_fallback", "fixup_items", "stft_chroma", "features_classify",
                    "hamming_runs", "simhash_runs", "epilogue_buckets", "epilogue_entries", "epilogue_best_match"]
    kernel_ms = {k: 0.0 for k in kernel_names}      # timed region: the dominant kernel only (see below)
    warm_ms = {k: 0.0 for k in kernel_names}        # warm-up: all kernels, to find the dominant one
    extra_ms = {k: 0.0 for k in kernel_names}       # untimed steps after the timed region: all kernels (breakdown)
    timed, acc = [list(kernel_names)], [warm_ms]
    state = {"runs": 0, "results": None}
    host_ms = {"enqueue": 0.0, "wait_and_epilogue": 0.0}
    finished = [0]
    pending = []
    seq = [0]

    # Jobs are pipelined two deep: job k's run list is downloaded asynchronously into pinned memory and its host
    # epilogue runs while job k+1's kernels execute.  Every job (analyze, [gather], search, [gather], download,
    # epilogue) completes inside the timed region; flush() finishes the one still in flight.
    def finish(slot, collect):
        t0 = time.perf_counter()
        state["results"], state["runs"] = lib.job_end(cmp, slot)
        if collect:
            host_ms["wait_and_epilogue"] += 1e3 * (time.perf_counter() - t0)
            finished[0] += 1

    def step(collect):
        slot = seq[0] & 1
        seq[0] += 1
        t0 = time.perf_counter()
        lib.job_begin(cmp, slot)
        if collect:
            host_ms["enqueue"] += 1e3 * (time.perf_counter() - t0)
        if pending:
            finish(pending.pop(), collect)                   # previous job's epilogue overlaps this job's kernels
        pending.append(slot)
        for k in timed[0]:                                   # events of the job before: already complete
            acc[0][k] += max(capi.last_kernel_ms(k), 0.0)

    def barrier():
        while pending:
            finish(pending.pop(), True)
        capi.synchronize()
        capi.comm_barrier()

    # HIP events around every kernel cost 3 % of a step (one more packet between dependent dispatches each), so
    # the timed region carries them for the dominant kernel only -- the one `roofline` is about, found in the
    # warm-up where all kernels are timed; the other kernels' times come from a few untimed steps afterwards.
    def rank0_says(x: float) -> float:                       # a decision every rank must take alike (a job has collectives)
        return float(capi.comm_all_gather(np.array([x], dtype=np.float64))[0, 0]) if world > 1 else x

    # First the COLD figure: W + K steps straight after set-up with nothing in front -- what `--preheat 0` measures and
    # what a driver that trusts its own --warmup sees.  It doubles as the start of the preheat.  Skipped when one job is
    # long enough (> 20 ms) that the device's ramp-up (~30 ms under load) cannot matter.
    step(False)
    barrier()                                                # first job: slabs grow, tables are built, scan repeated
    t0 = time.perf_counter()
    step(False)
    barrier()
    probe_ms = rank0_says(1e3 * (time.perf_counter() - t0))
    cold = None
    preheat_jobs = 0
    if args.preheat > 0 and probe_ms <= 20.0:
        for i in range(args.warmup):
            step(False)
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(False)
        barrier()
        cold_s = time.perf_counter() - t0
        if world > 1:
            cold_s = float(capi.comm_all_gather(np.array([cold_s], dtype=np.float64)).max())
        cold = {"value": round(n_pairs / (cold_s / args.steps), 2), "ms_per_step": round(1e3 * cold_s / args.steps, 4),
                "what": f"the same {args.warmup} + {args.steps} steps run first, two untimed jobs after set-up and nothing else "
                        "in front (the --preheat 0 figure); `value` is the steady state reached after the preheat jobs"}
        preheat_jobs = max(0, args.preheat - args.warmup - args.steps)
    telemetry = DeviceTelemetry(capi) if rank == 0 else None      # (sysfs probing takes milliseconds: not between warm-up and t0)
    for i in range(preheat_jobs):
        step(False)
    barrier()
    for k in warm_ms:
        warm_ms[k] = 0.0
    capi.set_kernel_timing("all")
    for i in range(args.warmup):
        step(False)
    barrier()
    dominant = max(warm_ms, key=warm_ms.get) if any(v > 0 for v in warm_ms.values()) else "stft_chroma32"
    capi.cert_stats(reset=True)
    if world > 1:                                            # every rank times the same kernel: rank 0 decides
        dominant = kernel_names[int(capi.comm_all_gather(np.array([kernel_names.index(dominant)], dtype=np.int32))[0, 0])]
    timed[0], acc[0] = [dominant], kernel_ms
    capi.set_kernel_timing(None if args.no_live_events else dominant)
    barrier()
    if telemetry:
        telemetry.start()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(True)
    barrier()
    elapsed = time.perf_counter() - t0
    device_state = telemetry.stop() if telemetry else None
    scan_form, scan_products = capi.scan_last_launch()           # of the timed jobs (a counting launch below is the vector form's)
    extra_steps = min(args.steps, 10)
    timed[0], acc[0] = list(kernel_names), extra_ms
    capi.set_kernel_timing("all")
    for i in range(extra_steps + 1):                 # a step reads the events of the job before it
        step(False)
    barrier()
    # the dominant kernel ALONE: with two jobs in flight a job's first pass runs beside the previous job's tail kernels
    # (second stream, libneedle_capi's default) and its live duration includes what they take from it; a few jobs run one
    # at a time give the kernel's own duration
    alone_ms, alone_n = 0.0, 0
    latency_resident_ms = 0.0
    capi.set_kernel_timing(dominant)
    for i in range(6):
        t_job = time.perf_counter()
        lib.job_begin(cmp, 0)
        state["results"], state["runs"] = lib.job_end(cmp, 0)
        t_job = time.perf_counter() - t_job
        capi.synchronize()
        if i:
            alone_ms += max(capi.last_kernel_ms(dominant), 0.0)
            alone_n += 1
            latency_resident_ms += 1e3 * t_job / 5
    capi.comm_barrier()
    capi.set_kernel_timing(None)
    # what the scan ISSUES, for roofline_search: one more (untimed) job through the counting instantiation of the kernel
    issued_evals = None
    if not args.no_extras:
        os.environ["NEEDLE_HIP_SCAN_COUNT"] = "1"
        try:
            step(False)
            barrier()
            capi.scan_issued_evaluations(reset=True)
            step(False)
            barrier()
            issued_evals = capi.scan_issued_evaluations(reset=True)
        finally:
            del os.environ["NEEDLE_HIP_SCAN_COUNT"]
    if world > 1:
        elapsed = float(capi.comm_all_gather(np.array([elapsed], dtype=np.float64)).max())

    if rank == 0:
        ms_per_step = 1000.0 * elapsed / args.steps
        value = n_pairs / (elapsed / args.steps)
        avg = {k: extra_ms[k] / max(extra_steps, 1) for k in kernel_names}
        avg[dominant] = kernel_ms[dominant] / args.steps         # live, inside the timed region
        # per-launch work of THIS rank's launch of the dominant kernel
        _, pcount = capi.comm_shard(n_pairs, world, 0)
        if dominant in ("hamming_runs", "simhash_runs"):
            abytes = algorithmic_bytes("hamming_runs", windows, kept, pcount, state["runs"] // world)
        else:                                                    # a rank fingerprints 1 / world of the hashes
            abytes = algorithmic_bytes(dominant, windows, kept, 0, 0) / world
        cs = capi.cert_stats()
        certified = cs["items"] > 0
        achieved = abytes / (avg[dominant] * 1e-3) / 1e9 if avg[dominant] > 0 else 0.0
        traffic = traffic_source = valu_issue = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")   # HBM bytes/launch from rocprofv3 PMC passes (N = 1 launch shape)
        if os.path.exists(tpath) and world == 1 and n == 28 and args.minutes == 24.0:
            try:
                tj = json.load(open(tpath))
                traffic = tj.get(dominant)
                valu_issue = (tj.get("valu_issue_frac") or {}).get(dominant)
                if traffic is not None:
                    traffic_source = (f"{tj.get('source', 'profiles/traffic.json')}: rocprofv3 --pmc pass of an earlier run of "
                                      "this command, NOT measured in this run (counters cannot be read from inside it)")
            except Exception:
                traffic = None
        compute = None
        if dominant in ("stft_chroma", "stft_chroma32") and avg[dominant] > 0:   # what actually bounds it: VALU issue + LDS exchange
            fl = stft_flops(windows) / world
            tf = fl / (avg[dominant] * 1e-3) / 1e12
            peak = F64_VALU_PEAK_TFLOPS if dominant == "stft_chroma" else F32_VALU_PEAK_TFLOPS
            compute = {"bound": "f64 valu" if dominant == "stft_chroma" else "f32 valu", "achieved": round(tf, 2), "peak": peak,
                       "unit": "TFLOP/s", "frac": round(tf / peak, 4), "flops_per_launch": int(fl),
                       "valu_issue_frac": valu_issue,
                       "valu_issue_note": "share of the SIMDs' cycles in which a vector instruction issues (SQ_ACTIVE_INST_VALU x 4 / SIMD "
                                          "cycles, profiles/traffic.json: the same earlier counter pass as `traffic`): the kernel's real bound -- "
                                          "of ~716 vector instructions per frame pair 432 are the butterflies' arithmetic"}
        out = {
            "metric": "episode-pairs/sec (analyze+search)", "value": round(value, 2), "unit": "episode-pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64-certified (f32 first pass)" if certified else "f64",
            "data": "synthetic",
            "config": {"workload": (f"{n} episodes x {args.minutes:g} min synthetic mono s16 PCM @ 11025 Hz, PCM RESIDENT IN HBM "
                                    f"before the timed region (host->device copy excluded: see end_to_end), "
                                    f"{args.intro_seconds:g} s shared intro, opening window 50 %, hash 0.3 s, "
                                    f"threshold 10, min opening 20 s; analyze+search, {n_pairs} pairs "
                                    + ("(BASELINE.json configs[4]'s library: 2000 x 45 min, full O(N^2) search; PCM generated in HBM, "
                                       "the streamed-from-pinned form is tools/library_stream_device.py)"
                                       if (n, args.minutes) == (2000, 45.0) else
                                       "(BASELINE.json configs[1]; configs[3] sharding when n_gpus > 1)"
                                       if (n, args.minutes) == (28, 24.0) else "(not one of BASELINE.json's configurations)")),
                       "episodes": n, "pairs": n_pairs, "hashes_per_episode": kept[0],
                       "parallelism": "1 gpu" if world == 1 else
                       f"{world} ranks, one process per GPU: equal blocks of hashes + pair ranges, 2 all-gathers per job "
                       f"inside libneedle_capi.so ({capi.comm_backend()})",
                       "comm": capi.comm_backend(),
                       "preheat_jobs": (2 + args.warmup + args.steps + preheat_jobs) if cold else 2,
                       "preheat_note": "untimed jobs in front of the W warm-up steps, the cold measurement's own steps "
                                       "included (a step is ~0.6 ms; the device needs ~30 ms under load before kernels run "
                                       "at their steady rate); `cold` is the figure without them.  Jobs longer than 20 ms "
                                       "get two untimed jobs and no cold figure",
                       "setup_s": round(setup_s, 2), "synth": "device (HBM)" if args.device_synth else "host",
                       "corpus": args.corpus},
            "cold": cold,
            "roofline": {"bound": "hbm", "kernel": dominant, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "traffic_source": traffic_source,
                         "algorithmic_bytes_per_launch": int(abytes),
                         "avg_launch_ms": round(avg[dominant], 5), "compute": compute,
                         "alone": None if not alone_n or dominant not in ("stft_chroma", "stft_chroma32") else {
                             "avg_launch_ms": round(alone_ms / alone_n, 5),
                             "achieved": round(abytes / (alone_ms / alone_n * 1e-3) / 1e9, 2),
                             "frac": round(abytes / (alone_ms / alone_n * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                             "what": "the same kernel in jobs run one at a time (5 launches after the timed region): in the "
                                     "timed region, two jobs in flight, a first pass shares the chip with the previous job's "
                                     "tail kernels -- its live duration above is longer, the job shorter"}},
            "kernel_ms_per_step": {k: round(v, 5) for k, v in avg.items()},
            "kernel_ms_note": f"rank 0's launches; {dominant}: HIP events inside the timed region; the others: {extra_steps} "
                              "untimed steps after it (events around every kernel slow a step by 3 %).  With two jobs in flight the "
                              "tail kernels of job k run BESIDE job k+1's first pass: these durations overlap and do not add up "
                              "to ms_per_step",
            "host_ms_per_step": {"enqueue": round(host_ms["enqueue"] / args.steps, 4),
                                 "wait_and_epilogue": round(host_ms["wait_and_epilogue"] / max(finished[0], 1), 4)},
            "runs_per_step": state["runs"],
            "fallback_frac": None if not certified else {
                "items": round(cs["items_recomputed"] / max(cs["items"], 1), 6),
                "frame_pair_chunks": round(cs["chunks_recomputed"] / max(cs["chunks"], 1), 6),
                "what": "share of the kept items the f32 first pass could not certify (recomputed from f64 chroma) and of "
                        "the 2-pair chunks of frame pairs the f64 kernel therefore ran over again; every emitted u32 is the "
                        "f64 pipeline's either way", "counts": cs},
            "device_state": device_state,
            "detected": sum(1 for r in state["results"] if r is not None and r.opening is not None),
        }
        mfma_roofline = None
        if scan_form == 4 and avg["hamming_runs"] > 0:          # this rank's scan took the matrix-pipe form (large launches)
            tops = scan_products * MATRIX_OPS / (avg["hamming_runs"] * 1e-3) / 1e12
            mfma_roofline = {"bound": "mfma", "achieved": round(tops, 1), "peak": FP4_DENSE_PEAK_TOPS, "unit": "TOP/s",
                             "frac": round(tops / FP4_DENSE_PEAK_TOPS, 4)}
            if dominant in ("hamming_runs", "simhash_runs"):     # (whatever --no-extras says, and for every world size)
                out["roofline"].update(mfma_roofline, matrix_instructions_per_launch=int(scan_products),
                                       hbm={"achieved_gbs": round(achieved, 2), "peak_gbs": HBM_PEAK_GBS,
                                            "frac": round(achieved / HBM_PEAK_GBS, 5),
                                            "note": "compute-bound by construction (~1450 ops per byte): the HBM fraction says nothing"})
        if not args.no_extras:
            try:
                out["roofline_search"] = search_roofline(capi.int_valu_ceiling(), issued_evals, float(pcount) * kept[0] * kept[0],
                                                         avg["hamming_runs"])
                form, products = scan_form, scan_products
                if form == 4:                                    # the job's scan took the matrix-pipe form (large launches)
                    sec = avg["hamming_runs"] * 1e-3
                    tops = products * MATRIX_OPS / sec / 1e12 if sec > 0 else 0.0
                    out["roofline_search"] = {
                        "bound": "mfma", "kernel": "hamming_runs (aligned windows, head rows on the matrix pipe)", "unit": "TOP/s",
                        "achieved": round(tops, 1), "peak": FP4_DENSE_PEAK_TOPS, "frac": round(tops / FP4_DENSE_PEAK_TOPS, 4),
                        "matrix_instructions_per_launch": int(products), "avg_launch_ms": round(avg["hamming_runs"], 5),
                        "note": "achieved = v_mfma_f32_32x32x64_f8f6f4 (FP4 operands) instructions of the launch x 131 072 operations / "
                                "kernel time; peak = the dense FP4 figure of MI355X_MICROARCH.md (4 x BF16).  Round 4 / early "
                                "round 5 multiplied the same bits as int8 (one head row per instruction, 5 POP/s peak): in those "
                                "units this is achieved / 5000"}
                if dominant in ("hamming_runs", "simhash_runs") and form == 4:
                    pass                                         # (done above)
                elif dominant in ("hamming_runs", "simhash_runs"):   # the scan dominates (library scale): it is not HBM-bound
                    rs = out["roofline_search"]
                    out["roofline"].update(bound="int valu", achieved=rs["achieved"], peak=rs["peak"], unit=rs["unit"], frac=rs["frac"],
                                           hbm={"achieved_gbs": round(achieved, 2), "peak_gbs": HBM_PEAK_GBS,
                                                "frac": round(achieved / HBM_PEAK_GBS, 5),
                                                "note": "integer-VALU-bound by construction (~1450 ops per byte): the HBM fraction says nothing"})
            except capi.NeedleError as e:
                out["roofline_search"] = {"error": str(e)}
        out["comm_bytes_per_job"] = dict(lib.job_comm_bytes(0), what="received per rank in one steady-state job: all-gather of "
                                         "hash rows (the arena's equal blocks); run_heads = the run lists -- with the sharded device "
                                         "epilogue (library scale) the OWNER-DIRECTED exchange: a count matrix + an all-to-all of blocks "
                                         "sized by the last job's counts, a run of pair (i, j) going to the owners of videos i and j only; "
                                         "otherwise the all-gather of every rank's head (count + the last job's largest list + margin) --; "
                                         "all-gather of per-video results when the epilogue is sharded; scans_repeated = overflows met "
                                         "(0 in the steady state)")
        out["host_threads_per_rank"] = capi.host_threads()
        out["job_form"] = lib.job_form(0)
        if world == 1 and not args.no_extras and eps is not None:
            out["end_to_end"] = end_to_end(capi, eps, cmp, n_pairs)
        if world == 1 and not args.no_extras:
            so = search_only(capi, synth, args.search_only_episodes, 24.0)
            vec = out.get("roofline_search", {})
            issued = so.pop("issued_evals", None)
            if so.get("scan_form") == 4 and so["scan_kernel_ms"] > 0:  # the call's scan took the matrix-pipe form
                tops = so["matrix_instructions"] * MATRIX_OPS / (so["scan_kernel_ms"] * 1e-3) / 1e12
                so["roofline"] = {"bound": "mfma", "kernel": "hamming_runs (aligned windows, first stage on the matrix pipe)",
                                  "unit": "TOP/s", "achieved": round(tops, 1), "peak": FP4_DENSE_PEAK_TOPS,
                                  "frac": round(tops / FP4_DENSE_PEAK_TOPS, 4)}
            elif "ceiling_cells_per_s" in vec:
                so["roofline"] = search_roofline(vec["ceiling_cells_per_s"], issued, so["table_cells"], so["scan_kernel_ms"])
            out["search_only"] = so
            if args.corpus == "tonal" and (n, args.minutes) == (28, 24.0):
                try:                                             # the same two shapes on the hostile corpus
                    out["corpus_hostile"] = corpus_hostile(capi, synth, 28, 24.0)
                    soh = search_only(capi, synth, args.search_only_episodes, 24.0, reps=6, hostile=True)
                    soh.pop("issued_evals", None)
                    out["corpus_hostile"]["search_only"] = soh
                    out["corpus_hostile"]["vs_tonal"] = {
                        "ms_per_step": round(out["corpus_hostile"]["ms_per_step"] / ms_per_step, 3),
                        "search_only_wall_ms": round(soh["wall_ms"] / so["wall_ms"], 3)}
                except Exception as e:                           # noqa: BLE001
                    out["corpus_hostile"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and not args.no_cpu_baseline:
            if eps is not None:
                sample_ids, sample_pcm = list(range(n)), [e.pcm[: len(e.pcm) // 2] for e in eps]
            hashes = [lib.frame_hashes(v).opening_data()[0] for v in sample_ids]
            out["cpu_baseline"] = cpu_baseline(sample_pcm, n, state["results"], hashes, whole_job=len(sample_ids) == n)
            try:                                                 # the checker's own pin (oracle/pin.py): costs nothing when absent
                from oracle.pin import probe_report
                out["oracle_pin"] = probe_report()
            except Exception as e:                               # noqa: BLE001
                out["oracle_pin"] = {"libchromaprint": "error", "error": f"{type(e).__name__}: {e}"}
        # one job at a time: resident (5 jobs after the timed region, wall clock around job_begin .. job_end) and from pinned
        # host PCM (end_to_end: upload + analyze + search + epilogue)
        out["latency_ms"] = {"resident": round(latency_resident_ms, 4),
                             "from_pinned_host_pcm": out.get("end_to_end", {}).get("pinned", {}).get("ms_per_job"),
                             "what": "one job at a time, wall clock: job_begin .. job_end with the PCM resident in HBM; and "
                                     "needle_hip_library_stream_pcm + job_begin .. job_end from pinned host memory.  `value` is "
                                     "steady-state throughput with two jobs in flight"}
        # the like-for-like ratio: BASELINE.md publishes no number for this metric on this hardware, so the baseline is the
        # reference's CPU path timed beside it (cpu_baseline, same box, PCM in host memory) against the GPU path that also
        # starts from host memory (end_to_end.pinned) -- NOT the resident figure `value` holds
        e2e = out.get("end_to_end", {}).get("pinned", {}).get("pairs_per_s")
        cpu = out.get("cpu_baseline", {}).get("value")
        if e2e and cpu:
            out["vs_baseline"] = round(e2e / cpu, 2)
            out["vs_baseline_what"] = ("end_to_end.pinned.pairs_per_s / cpu_baseline.value: both start from PCM in host memory, same "
                                       "machine, same job; the resident figure `value` over the same baseline is "
                                       f"{round(value / cpu, 1)}")
        # LAST: 30 GB of PCM generated in HBM and several all-pairs jobs -- every other figure of the line has
        # been measured by now, so a failure here (a device without 30 GB to spare) cannot colour them; its buffers are
        # released whatever happens
        if world == 1 and not args.no_extras and args.library_scale_episodes >= 2 and (n, args.minutes) == (28, 24.0):
            try:
                out["library_scale"] = library_scale(capi, synth, args.library_scale_episodes, 45.0)
            except Exception as e:                               # noqa: BLE001
                import gc
                gc.collect()                                     # the leg's Library / DeviceLibrary objects free their HBM in __del__
                try:
                    capi.synchronize()
                except Exception:                                # noqa: BLE001
                    pass
                out["library_scale"] = {"error": f"{type(e).__name__}: {e}"}
        if sup_dir:                                              # supervisor 0 prints it once the attempt has succeeded
            tmp = os.path.join(sup_dir, f".result.{attempt}.tmp")
            with open(tmp, "w") as f:
                f.write(json.dumps(out) + "\n")
            os.replace(tmp, os.path.join(sup_dir, f"result.{attempt}"))
        else:
            print(json.dumps(out), flush=True)
    if rdzv is not None:
        capi.comm_barrier()
        if sup_dir:                                              # every collective of this rank has completed
            open(os.path.join(sup_dir, f"finished.{attempt}.{rank}"), "w").close()
        capi.comm_finalize()
        rdzv.close()


if __name__ == "__main__":
    main()
