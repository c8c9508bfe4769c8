#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on MI355X.

metric  : episode-pairs/sec (analyze+search) = N(N-1)/2 / wall(analyze N episodes + search all pairs +
          per-video best match), N = 28 synthetic 24-min episodes (BASELINE.json configs[1]).
step    : one complete pass of the hot path over the library: fingerprint every episode's opening window
          (stft_chroma and features_classify kernels), scan every pair (hamming_runs kernel), copy the run list
          and the hash arena back, run the order-sensitive host epilogue (duration validity, simhash32,
          BinaryHeap order, find_best_match).  PCM is resident in HBM before the timed region starts.
N > 1   : one process per GPU (torch.distributed, backend nccl = RCCL).  The SAME 28-episode job is sharded:
          episodes in contiguous blocks, one all-gather of hash rows, pairs in contiguous ranges, one
          all-gather of run lists, epilogue on rank 0 ("strong" scaling, BASELINE.json configs[3]).

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel, timed live with HIP events on the
library's stream inside the timed region (the other kernels are timed in a few untimed steps after it); `cpu_baseline` times the oracle (the C restatement of the reference's CPU path, full
DP table per pair, one task per episode / pair over all host cores) on rank 0 at N = 1.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
RATE = 11025


def algorithmic_bytes(kernel: str, windows, kept, n_pairs: int, n_runs: int) -> float:
    """analyze 2*S + 4*H per episode; search 4*(n+m) + 12*R per pair (R runs of 3 u32)."""
    if kernel == "hamming_runs":
        n = sum(kept) / max(len(kept), 1)
        return n_pairs * 4.0 * 2.0 * n + 12.0 * n_runs
    return float(sum(2 * s + 4 * h for s, h in zip(windows, kept)))


F64_VALU_PEAK_TFLOPS = 78.6    # vector f64 = half the 157.3 TFLOP/s f32 vector rate of MI355X_MICROARCH.md (no faster f64 MFMA)


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


def cpu_baseline(eps, results_gpu, hashes_gpu):
    """The oracle on the host cores (rank 0, N = 1 only): same episodes, same pairs, reference cost
    structure.  Bounded: the whole 28-episode job when the core count makes it ~<= 30 s, else a prefix of
    the pair list, scaled.  Also cross-checks the GPU's hashes and results against it."""
    from oracle import oracle as O
    threads = usable_cpus()
    hd = O.duration_from_secs_f32(0.3)
    windows = [e.pcm[: len(e.pcm) // 2] for e in eps]
    t0 = time.perf_counter()
    fhs = O.analyze_batch(windows, 1, hd, threads=threads)
    t_analyze = time.perf_counter() - t0
    n = len(eps)
    pairs_total = n * (n - 1) // 2
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
    # same scan without the table, one pass per diagonal, all pairs over all threads; same analyze stage
    seqs = [np.array([h for h, _ in f.opening], dtype=np.uint32) for f in fhs]
    t0 = time.perf_counter()
    opt_runs, _ = O.diagonal_runs_all_pairs(seqs, 10, 82, threads=threads)
    t_opt = time.perf_counter() - t0
    parity = all(hashes_gpu[v].tolist() == [h for h, _ in fhs[v].opening] for v in range(n))
    if k == n:
        got = [None if r is None else (r.opening, r.ending) for r in results_gpu]
        want = [None if r is None else (r.opening, r.ending) for r in res]
        parity = parity and got == want
    return {
        "value": round(value, 3), "unit": "episode-pairs/s", "cores": threads, "kind": "port",
        "sample": f"oracle (C restatement of analyzer.rs/comparator.rs + chromaprint, not the Rust binary): "
                  f"analyze all {n} episodes in {t_analyze:.2f} s, search {sample_pairs}/{pairs_total} pairs in "
                  f"{t_search:.2f} s (scaled to all pairs), {threads} threads",
        "analyze_s": round(t_analyze, 3), "search_s_scaled": round(t_search_full, 3),
        "gpu_matches_oracle": bool(parity),
        "optimised_cpu_variant": {"value": round(pairs_total / (t_analyze + t_opt), 3), "unit": "episode-pairs/s",
                                  "search_s": round(t_opt, 4), "runs": int(opt_runs),
                                  "what": "same analyze stage + table-free diagonal scan of all pairs (min run 82), "
                                          f"{threads} threads, without the per-video epilogue"},
    }


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--episodes", type=int, default=28)
    ap.add_argument("--minutes", type=float, default=24.0)
    ap.add_argument("--intro-seconds", type=float, default=90.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-dist", action="store_true", help="use the torch.distributed path even at N=1")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1 or args.force_dist

    torch = dist = None
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29513")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import torch  # noqa: F811  (loads its HIP runtime first; libneedle_capi.so binds to the same one)
        import torch.distributed as dist  # noqa: F811
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from needle_amd import capi, synth
    from needle_amd import dist as ndist

    if capi.device_count() < 1:
        raise SystemExit("bench.py needs a HIP device: the needle path has no CPU fallback")
    capi.set_device(local_rank if distributed else 0)

    n = args.episodes
    eps = synth.make_library(n, args.minutes * 60.0, args.intro_seconds)
    first, count = ndist.shard(n, world, rank)
    lib = capi.Library(n)
    lib.set_pcm([e.pcm if first <= k < first + count else None for k, e in enumerate(eps)],
                [len(e.pcm) for e in eps])
    cmp = capi.Comparator([f"episode-{k:04d}.wav" for k in range(n)])
    cmp.handle()
    n_pairs = lib.num_pairs()
    cap = max(1 << 16, 4 * n_pairs)          # run-list capacity: every pair of a library with one shared intro matches
    windows = [len(e.pcm) // 2 for e in eps]
    kept = [capi.lib().needle_hip_fingerprint_num_kept(w, 2) for w in windows]

    kernel_names = ["stft_chroma", "features_classify", "hamming_runs", "simhash_runs"]
    kernel_ms = {k: 0.0 for k in kernel_names}      # timed region: the dominant kernel only (see below)
    warm_ms = {k: 0.0 for k in kernel_names}        # warm-up: all kernels, to find the dominant one
    extra_ms = {k: 0.0 for k in kernel_names}       # untimed steps after the timed region: all kernels (breakdown)
    timed, acc = [list(kernel_names)], [warm_ms]
    state = {"runs": 0, "results": None}

    if not distributed:
        d_runs, d_count = capi.DeviceBuffer(cap * capi.RUN_DTYPE.itemsize), capi.DeviceBuffer(4)

        host_ms = {"enqueue": 0.0, "enqueue_analyze": 0.0, "enqueue_search": 0.0, "wait_runs": 0.0, "epilogue": 0.0}
        # Jobs are pipelined two deep: job k's run list is downloaded asynchronously into pinned memory and its
        # host epilogue runs while job k+1's kernels execute.  Every job (analyze, search, download, epilogue)
        # completes inside the timed region; flush() finishes the one still in flight.
        max_runs = max(4096, cap if n_pairs > 2048 else 0)   # what the asynchronous download fetches: at library scale the
                                                             # whole buffer (a second, synchronous trip would wait for the NEXT job too)
        bufs = [(capi.DeviceBuffer(cap * capi.RUN_DTYPE.itemsize), capi.DeviceBuffer(4)) for _ in range(2)]
        pending = []
        seq = [0]
        trace_host = os.environ.get("NEEDLE_BENCH_TRACE") is not None   # host-side timeline of the job pipeline on stderr
        t_origin = time.perf_counter()
        finished = [0]                           # epilogues counted in host_ms (the flushes add a few to the K steps)

        def finish(slot, collect):
            t0 = time.perf_counter()
            runs, found = lib.fetch_runs_end(slot, max_runs)
            if found > max_runs:                                 # rare: fetch the whole list synchronously
                if found > cap:
                    raise SystemExit("run list overflow")
                runs = bufs[slot][0].to_host(capi.RUN_DTYPE, found)
            t1 = time.perf_counter()
            state["results"] = lib.finalize(cmp, runs)
            state["runs"] = found
            if trace_host:
                print(f"[bench] finish slot {slot}: wait {1e3 * (t0 - t_origin):9.2f} -> {1e3 * (t1 - t_origin):9.2f}, "
                      f"epilogue -> {1e3 * (time.perf_counter() - t_origin):9.2f} ms", file=sys.stderr)
            if collect:
                host_ms["wait_runs"] += 1e3 * (t1 - t0)
                host_ms["epilogue"] += 1e3 * (time.perf_counter() - t1)
                finished[0] += 1

        def step(collect):
            slot = seq[0] & 1
            seq[0] += 1
            t0 = time.perf_counter()
            d_runs, d_count = bufs[slot]
            lib.analyze(0, n, sync=False)
            ta = time.perf_counter()
            lib.search(cmp, 0, n_pairs, d_runs.ptr, cap, d_count.ptr, sync=False)
            ts = time.perf_counter()
            lib.fetch_runs_begin(slot, d_runs.ptr, d_count.ptr, max_runs)
            if trace_host:
                print(f"[bench] enqueue slot {slot}: {1e3 * (t0 - t_origin):9.2f} -> {1e3 * (time.perf_counter() - t_origin):9.2f} ms",
                      file=sys.stderr)
            if collect:
                host_ms["enqueue"] += 1e3 * (time.perf_counter() - t0)
                host_ms["enqueue_analyze"] += 1e3 * (ta - t0)
                host_ms["enqueue_search"] += 1e3 * (ts - ta)
            if pending:
                finish(pending.pop(), collect)                   # previous job's epilogue overlaps this job's kernels
            pending.append(slot)
            for k in timed[0]:                                   # events of the job before: already complete
                acc[0][k] += max(capi.last_kernel_ms(k), 0.0)

        def flush():
            while pending:
                finish(pending.pop(), True)

        def barrier():
            flush()
            capi.synchronize()
    else:
        b = ndist.block(n, world)
        _, stride = lib.hash_arena()
        arena = torch.zeros((b * world, stride), dtype=torch.int32, device="cuda")
        lib.use_hash_arena(arena.data_ptr(), b * world, stride)
        t_runs = torch.zeros((cap, capi.RUN_WORDS), dtype=torch.int32, device="cuda")
        t_count = torch.zeros(1, dtype=torch.int32, device="cuda")

        # torch's "current stream" becomes the library's own stream: the staging copies and the collectives'
        # stream dependencies are then ordered against the library's kernels by the stream itself, and the only
        # host wait of a job is the download of the gathered run list
        torch.cuda.set_stream(torch.cuda.ExternalStream(capi.stream_ptr(), device=torch.device("cuda", local_rank)))

        def sync():
            pass

        def full_sync():
            capi.synchronize()
            torch.cuda.synchronize()

        def search_pairs(pfirst, pcount):
            lib.search(cmp, pfirst, pcount, t_runs.data_ptr(), cap, t_count.data_ptr(), sync=False)
            return t_runs, t_count          # gathered with one fixed-size collective (ndist.gather_runs_slab)

        def finalize(runs_np):
            runs = np.ascontiguousarray(runs_np.astype(np.int32)).view(capi.RUN_DTYPE).reshape(-1)
            state["runs"] = len(runs)
            return lib.finalize(cmp, runs)

        # communicator set-up (RCCL creates its channels lazily on first use): two throw-away collectives of the
        # job's own shapes, so that it never lands in a timed step however short the warm-up is
        warm = torch.zeros((b * world, stride), dtype=torch.int32, device="cuda")
        for _ in range(2):
            ndist.gather_rows(warm, world, rank)
        dist.barrier()
        torch.cuda.synchronize()
        del warm
        gather = ndist.SlabGather(t_runs, world)
        row_block = torch.zeros((b, stride), dtype=torch.int32, device="cuda")
        pipe = ndist.JobPipeline(n, world, rank, arena, lambda f, c: lib.analyze(f, c, sync=False), search_pairs,
                                 finalize, gather, row_block, side_stream=torch.cuda.Stream(priority=-1))

        def step(collect, prefetch=True):
            res = pipe.step(prefetch)
            if rank == 0:
                state["results"] = res
            for k in timed[0]:
                acc[0][k] += max(capi.last_kernel_ms(k), 0.0)

        def barrier():
            full_sync()
            dist.barrier()
            torch.cuda.synchronize()

    # N > 1: a step enqueues the NEXT job's fingerprinting before it waits for its own run list, except on the
    # last step of a phase, so the timed region holds exactly `steps` analyses and `steps` searches
    ahead = (lambda i, total: {"prefetch": i + 1 < total}) if distributed else (lambda i, total: {})
    # HIP events around every kernel cost 3 % of a step (one more packet between dependent dispatches each), so
    # the timed region carries them for the dominant kernel only -- the one `roofline` is about, found in the
    # warm-up where all kernels are timed; the other kernels' times come from a few untimed steps afterwards.
    capi.set_kernel_timing("all")
    for i in range(args.warmup):
        step(False, **ahead(i, args.warmup))
    barrier()
    dominant = max(warm_ms, key=warm_ms.get) if any(v > 0 for v in warm_ms.values()) else "stft_chroma"
    timed[0], acc[0] = [dominant], kernel_ms
    capi.set_kernel_timing(dominant)
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(True, **ahead(i, args.steps))
    barrier()
    elapsed = time.perf_counter() - t0
    extra_steps = min(args.steps, 10)
    timed[0], acc[0] = list(kernel_names), extra_ms
    capi.set_kernel_timing("all")
    for i in range(extra_steps + 1):                 # a step reads the events of the job before it
        step(False, **ahead(i, extra_steps + 1))
    barrier()
    capi.set_kernel_timing(None)
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        ms_per_step = 1000.0 * elapsed / args.steps
        value = n_pairs / (elapsed / args.steps)
        avg = {k: extra_ms[k] / max(extra_steps, 1) for k in kernel_names}
        avg[dominant] = kernel_ms[dominant] / args.steps         # live, inside the timed region
        # per-launch work of THIS rank's launch of the dominant kernel
        if dominant == "hamming_runs":
            _, pcount = ndist.shard(n_pairs, world, 0)
            abytes = algorithmic_bytes(dominant, windows, kept, pcount, state["runs"])
        else:
            f0, c0 = ndist.shard(n, world, 0)
            abytes = algorithmic_bytes(dominant, windows[f0:f0 + c0], kept[f0:f0 + c0], 0, 0)
        achieved = abytes / (avg[dominant] * 1e-3) / 1e9 if avg[dominant] > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")   # HBM bytes/launch from rocprofv3 PMC passes
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(dominant)
            except Exception:
                traffic = None
        compute = None
        if dominant == "stft_chroma" and avg[dominant] > 0:   # what actually bounds it: f64 issue + LDS exchange latency
            fl = stft_flops(windows[f0:f0 + c0])
            tf = fl / (avg[dominant] * 1e-3) / 1e12
            compute = {"bound": "f64 valu", "achieved": round(tf, 2), "peak": F64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                       "frac": round(tf / F64_VALU_PEAK_TFLOPS, 4), "flops_per_launch": int(fl)}
        out = {
            "metric": "episode-pairs/sec (analyze+search)", "value": round(value, 2), "unit": "episode-pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"{n} episodes x {args.minutes:g} min synthetic mono s16 PCM @ 11025 Hz, "
                                   f"{args.intro_seconds:g} s shared intro, opening window 50 %, hash 0.3 s, "
                                   f"threshold 10, min opening 20 s; analyze+search, {n_pairs} pairs "
                                   f"(BASELINE.json configs[1]; configs[3] sharding when n_gpus > 1)",
                       "episodes": n, "pairs": n_pairs, "hashes_per_episode": kept[0],
                       "parallelism": "1 gpu" if world == 1 else f"{world} ranks: episode blocks + pair ranges, "
                                                                "2 all-gathers (RCCL)"},
            "roofline": {"bound": "hbm", "kernel": dominant, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "algorithmic_bytes_per_launch": int(abytes),
                         "avg_launch_ms": round(avg[dominant], 5), "compute": compute},
            "kernel_ms_per_step": {k: round(v, 5) for k, v in avg.items()},
            "kernel_ms_note": f"{dominant}: HIP events inside the timed region; the others: {extra_steps} untimed "
                              "steps after it (events around every kernel slow a step by 3 %)",
            "host_ms_per_step": ({k: round(v / (max(finished[0], 1) if k in ("wait_runs", "epilogue") else args.steps), 4)
                                  for k, v in host_ms.items()} if not distributed else None),
            "runs_per_step": state["runs"],
            "detected": sum(1 for r in state["results"] if r is not None and r.opening is not None),
        }
        if world == 1 and not args.no_cpu_baseline:
            hashes = [lib.frame_hashes(v).opening_data()[0] for v in range(n)]
            out["cpu_baseline"] = cpu_baseline(eps, state["results"], hashes)
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
