"""-m gpu: BASELINE.json configs[4]'s SHAPE (45-min episodes: 5 441 hashes per opening window, several bands per
scan workgroup, run lists of tens of thousands of runs, threaded host epilogue) checked against the oracle at a size the
oracle finishes in about a minute: 200 episodes x 45 min = 19 900 pairs, 5.9e11 table cells."""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from needle_amd import capi, synth
from oracle import oracle as O

pytestmark = pytest.mark.gpu
NS = O.NS

N_EPISODES = int(os.environ.get("NEEDLE_TEST_LIBRARY_EPISODES", "200"))
MINUTES = 45.0


def _cpus():
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            return max(1, min(len(os.sched_getaffinity(0)), int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, len(os.sched_getaffinity(0)))


def test_library_scale_hashes_runs_and_results_match_oracle(monkeypatch):
    assert capi.device_count() > 0
    n, threads = N_EPISODES, _cpus()
    half = MINUTES * 60.0 / 2                                     # only the opening half is ever hashed
    with ThreadPoolExecutor(max_workers=min(threads, 16)) as pool:
        eps = list(pool.map(lambda k: synth.make_episode(k, half, 90.0), range(n)))
    lens = [len(e.pcm) for e in eps]
    monkeypatch.setenv("NEEDLE_HIP_SLAB_RUNS", "2048")            # first job overflows its slab: grow + rescan
    lib = capi.Library(n, opening_search_percentage=1.0)
    lib.stream_pcm([e.pcm for e in eps], lens)                    # streamed: 3 GB of PCM through the 2 GiB staging arena
    cmp = capi.Comparator([f"episode-{k:04d}.wav" for k in range(n)])
    lib.job_begin(cmp, 0)
    results, found = lib.job_end(cmp, 0)
    lib.job_begin(cmp, 1)                                         # steady state: one-trip download
    results2, found2 = lib.job_end(cmp, 1)
    assert found == found2 and found >= n * (n - 1) // 2
    assert [None if r is None else (r.opening, r.ending) for r in results] == \
           [None if r is None else (r.opening, r.ending) for r in results2]
    assert all(r is not None and r.opening is not None for r in results)

    # (1) hashes of a sample of episodes vs the oracle's f64 pipeline
    hd = O.duration_from_secs_f32(0.3)
    gpu_hashes = [lib.frame_hashes(v).opening_data()[0] for v in range(n)]
    assert len(gpu_hashes[0]) == 5441
    sample = sorted(set([0, 1, n // 3, n // 2, n - 2, n - 1]))
    ref = O.analyze_batch([eps[v].pcm for v in sample], 1, hd, threads=threads)
    for v, fh in zip(sample, ref):
        assert gpu_hashes[v].tolist() == [h for h, _ in fh.opening], f"episode {v}"

    # (2) the complete run list vs the oracle's table-free scan of ALL pairs (same hashes in, min run 82)
    cap = max(4 * found, 1 << 16)
    d_runs, d_count = capi.DeviceBuffer(cap * capi.RUN_DTYPE.itemsize), capi.DeviceBuffer(4)
    lib.search(cmp, 0, lib.num_pairs(), d_runs.ptr, cap, d_count.ptr, sync=True)
    k = int(d_count.to_host(np.uint32, 1)[0])
    assert k == found
    runs = d_runs.to_host(capi.RUN_DTYPE, k)
    total, want = O.diagonal_runs_all_pairs(gpu_hashes, 10, 82, threads=threads, capacity=cap)
    assert total == k
    got = np.stack([runs["problem"], runs["src_end"], runs["dst_end"], runs["len"]], axis=1).astype(np.uint32)
    order_g = np.lexsort((got[:, 2], got[:, 1], got[:, 0]))
    order_w = np.lexsort((want[:, 2], want[:, 1], want[:, 0]))
    assert np.array_equal(got[order_g], want[order_w])
    # simhashes of a sample of runs (comparator.rs:226-229: L + 1 hashes)
    pairs = [(i, j) for i in range(n) for j in range(i + 1, n)]
    for q in np.linspace(0, k - 1, 200).astype(int):
        r = runs[q]
        i, j = pairs[int(r["problem"])]
        a, b, ln = int(r["src_end"]), int(r["dst_end"]), int(r["len"])
        assert int(r["src_match_hash"]) == O.simhash32(gpu_hashes[i][a - ln: a + 1].tolist())
        assert int(r["dst_match_hash"]) == O.simhash32(gpu_hashes[j][b - ln: b + 1].tolist())

    # (3) final results on a 40-episode sub-library vs the reference path (full tables, heap order, best match)
    m = min(40, n)
    sub = capi.Library(m, opening_search_percentage=1.0)
    sub.stream_pcm([e.pcm for e in eps[:m]], lens[:m])
    cmp_m = capi.Comparator([f"episode-{k:04d}.wav" for k in range(m)])
    sub.job_begin(cmp_m, 0)
    res_m, _ = sub.job_end(cmp_m, 0)
    ofh = []
    for v in range(m):
        h = gpu_hashes[v]
        ofh.append(O.FrameHashes(O.step_and_timestamp(np.repeat(h, 2)[: 2 * len(h) - 1], hd), [], hd, ""))
        assert [x for x, _ in ofh[-1].opening] == h.tolist()
    want_m = O.run_with_frame_hashes(O.Comparator(), ofh, threads=threads)
    assert [None if r is None else (r.opening, r.ending) for r in res_m] == \
           [None if r is None else (r.opening, r.ending) for r in want_m]


CONFIG4_EPISODES = int(os.environ.get("NEEDLE_TEST_CONFIG4_EPISODES", "2000"))


def test_config4_full_size_on_one_gpu():
    """BASELINE.json configs[4] at its FULL size on one MI355X: 2000 episodes x 45 min, 1 999 000 pairs, 59.5 GB of
    opening-window PCM -- generated in HBM (13 minutes of host synthesis otherwise), so the PCM is resident and the job
    is analyze + search + epilogue.  The oracle cannot run this job; it checks, with the GPU's own hashes as input where
    the stage under test starts from hashes:
      (1) the u32 hashes of sampled episodes against its f64 pipeline on the PCM read back from the device;
      (2) the run list of all 2016 pairs among 64 sampled episodes against its table-free scan;
      (3) the final results of 24 sampled videos -- each depends on its 1999 pairs -- against comparator.rs:524-629
          through the table-free pair function (oracle/ora_needle.h ora_run_selected_videos);
    and the planted ground truth: every episode's opening is detected on the shared intro."""
    assert capi.device_count() > 0
    n, threads = CONFIG4_EPISODES, _cpus()
    samples = int(round(MINUTES * 60.0 / 2 * 11025))            # 14 883 750: the opening half of 45 minutes
    gen = synth.DeviceLibrary(n, samples, 90.0)
    rng = np.random.default_rng(4)
    sample_eps = sorted(set([0, 1, n // 3, n // 2, n - 2, n - 1]))
    sample_pcm = {v: gen.episode(v) for v in sample_eps}
    lib = capi.Library(n, opening_search_percentage=1.0)
    lib.set_pcm_device(gen.pointers(), [samples] * n)
    gen.free()
    cmp = capi.Comparator([f"episode-{k:04d}.wav" for k in range(n)])
    lib.job_begin(cmp, 0)
    results, found = lib.job_end(cmp, 0)                         # first job: run slabs grow, scan repeated
    lib.job_begin(cmp, 1)
    results2, found2 = lib.job_end(cmp, 1)
    assert found == found2 and found >= n * (n - 1) // 2
    if "NEEDLE_HIP_SCAN_MFMA" not in os.environ:                 # a launch of this size takes the scan's matrix-pipe form by itself
        form, products = capi.scan_last_launch()
        assert form == 4 and products > 0
    got_all = [None if r is None else (r.opening, r.ending) for r in results]
    assert got_all == [None if r is None else (r.opening, r.ending) for r in results2]
    assert sum(1 for r in results if r is not None and r.opening is not None) == n      # detected: 2000
    # planted truth: the opening found covers most of the 90 s intro at the episode's offset (the reference's clock
    # runs 0.65 % slow and a hash covers 2.7 s: DESIGN.md §2 -- +-2.5 s is that, not a tolerance of this build)
    for v in range(0, n, max(1, n // 50)):
        start, end = results[v].opening
        off = gen.intro_off[v] / 11025.0
        clock = 0.123 / (1365 / 11025.0)
        assert abs(start / 1e9 - off * clock) < 4.0 and abs(end / 1e9 - (off + 90.0) * clock) < 4.0, (v, start, end, off)

    # the certified first pass audited over ALL 2000 episodes: the f64 kernel over the same resident PCM, every one of the
    # 10.9 M kept items compared on the device (the oracle checks below can only sample)
    audit = lib.audit()
    print("audit 2000 x 45 min:", audit)
    assert audit["items"] == n * 5441 and audit["mismatches"] == 0 and audit["accepted_mismatches"] == 0
    assert audit["accepted"] > 0.99 * audit["items"] and audit["max_error_over_s"] <= 8.0

    # every video's kept hashes in one copy of the arena
    d_arena, stride = lib.hash_arena()
    kept = 5441
    arena = np.zeros(n * stride, dtype=np.uint32)
    capi.check(capi.lib().needle_hip_memcpy_d2h(arena.ctypes.data, d_arena, arena.nbytes))
    gpu_hashes = [arena[v * stride: v * stride + kept] for v in range(n)]
    hd = O.duration_from_secs_f32(0.3)
    fh0 = lib.frame_hashes(0)
    assert len(fh0.opening_data()[0]) == kept and fh0.opening_data()[0].tolist() == gpu_hashes[0].tolist()
    ts = fh0.opening_data()[1].astype(np.uint64)                 # the same for every video: equal lengths

    # (1) hashes
    ref = O.analyze_batch([sample_pcm[v] for v in sample_eps], 1, hd, threads=threads)
    for v, fh in zip(sample_eps, ref):
        assert gpu_hashes[v].tolist() == [h for h, _ in fh.opening], f"episode {v}"
        assert ts.tolist() == [t for _, t in fh.opening]

    # (2) runs of the pairs among 64 sampled episodes
    cap = max(2 * found, 1 << 16)
    d_runs, d_count = capi.DeviceBuffer(cap * capi.RUN_DTYPE.itemsize), capi.DeviceBuffer(4)
    lib.search(cmp, 0, lib.num_pairs(), d_runs.ptr, cap, d_count.ptr, sync=True)
    k = int(d_count.to_host(np.uint32, 1)[0])
    assert k == found
    runs = d_runs.to_host(capi.RUN_DTYPE, k)
    sub = np.sort(rng.choice(n, size=min(64, n), replace=False))
    pos = -np.ones(n, dtype=np.int64)
    pos[sub] = np.arange(len(sub))
    # global pair index -> (i, j): row i of the upper triangle starts at i n - i (i + 1) / 2
    starts = np.array([i * n - i * (i + 1) // 2 for i in range(n)], dtype=np.int64)
    pi = np.searchsorted(starts, runs["problem"].astype(np.int64), side="right") - 1
    pj = runs["problem"].astype(np.int64) - starts[pi] + pi + 1
    inside = (pos[pi] >= 0) & (pos[pj] >= 0)
    m = len(sub)
    local = pos[pi[inside]] * m - pos[pi[inside]] * (pos[pi[inside]] + 1) // 2 + (pos[pj[inside]] - pos[pi[inside]] - 1)
    got = np.stack([local, runs["src_end"][inside], runs["dst_end"][inside], runs["len"][inside]], axis=1).astype(np.uint32)
    total, want = O.diagonal_runs_all_pairs([gpu_hashes[v] for v in sub], 10, 82, threads=threads, capacity=4 * len(got) + 1024)
    assert total == len(got) and total >= m * (m - 1) // 2
    assert np.array_equal(got[np.lexsort((got[:, 2], got[:, 1], got[:, 0]))], want[np.lexsort((want[:, 2], want[:, 1], want[:, 0]))])

    # (3) final results of sampled videos: each is a function of all its n - 1 pairs
    sel = sorted(set([0, n - 1] + rng.choice(n, size=min(22, n), replace=False).tolist()))
    want_sel = O.run_selected_videos(O.Comparator(), gpu_hashes, [ts] * n, hd, sel, threads=threads)
    assert [None if r is None else (r.opening, r.ending) for r in want_sel] == [got_all[v] for v in sel]


RANKS_EPISODES = int(os.environ.get("NEEDLE_TEST_RANKS_EPISODES", "400"))


@pytest.mark.parametrize("world,env", [
    (4, {"NEEDLE_HIP_SLAB_RUNS": "2048"}),                       # slab overflow on the first job: grow + rescan + regather
    (5, {"NEEDLE_HIP_HEAD_RUNS": "64", "NEEDLE_HIP_SHARD_EPILOGUE": "0"}),   # head overflow; unsharded epilogue on every rank
])
def test_config4_shape_between_ranks_on_one_gpu(tmp_path, monkeypatch, world, env):
    """BASELINE.json configs[4]'s shape with world > 1: 400 episodes x 45 min, 79 800 pairs, 5 441
    hashes per episode, every rank a real process on device 0 over the host-staged transport, PCM generated in HBM per
    rank for the episodes needle_hip_library_rank_videos names.  Exercised at the size that triggers them: hash-block
    sharding that cuts rows mid-episode, the run-slab overflow (grow + rescan + regather) or the head overflow, the
    sharded epilogue (> 2^17 runs: its extra gather) and the unsharded one, two jobs in flight.  Every rank's results,
    run count, complete run list (order-free digest) and hash arena equal the single-rank job's.  (Five ranks is what a
    one-GPU box allows beside the test process: its guard stops a run with more than 6 processes on the GPU.)"""
    from tests.comm_worker import run_digest
    from tests.test_comm_cpu import launch
    import hashlib
    assert capi.device_count() > 0
    n = RANKS_EPISODES
    samples = int(round(MINUTES * 60.0 / 2 * 11025))
    gen = synth.DeviceLibrary(n, samples, 90.0)
    lib = capi.Library(n, opening_search_percentage=1.0)
    lib.set_pcm_device(gen.pointers(), [samples] * n)
    gen.free()
    cmp = capi.Comparator([f"episode-{k:04d}.wav" for k in range(n)])
    lib.job_begin(cmp, 0)
    res, found = lib.job_end(cmp, 0)
    want_results = [None if r is None else [None if x is None else list(x) for x in (r.opening, r.ending)] for r in res]
    want_runs = lib.job_runs(0).copy()
    want_digest = run_digest(want_runs)
    # what a rank holds after an OWNER-DIRECTED exchange (round 6): the runs of the pairs with a video of its block
    pi, pj = np.triu_indices(n, 1)                                # pair p = (pi[p], pj[p]), i-major (comparator.rs:534-545)
    held_digest, held_count = {}, {}
    for r in range(world):
        f, c = capi.comm_shard(n, world, r)
        i, j = pi[want_runs["problem"]], pj[want_runs["problem"]]
        mine = ((i >= f) & (i < f + c)) | ((j >= f) & (j < f + c))
        held_digest[r], held_count[r] = run_digest(want_runs[mine]), int(mine.sum())
    assert lib.job_comm_bytes(0) == {"hash_rows": 0, "run_heads": 0, "results": 0, "scans_repeated": 0}
    # 79 800 pairs: the job above ran the per-video epilogue on the device (epilogue.hip); the host form must agree
    monkeypatch.setenv("NEEDLE_HIP_DEVICE_EPILOGUE", "0")
    lib.job_begin(cmp, 1)
    res_host, found_host = lib.job_end(cmp, 1)
    monkeypatch.delenv("NEEDLE_HIP_DEVICE_EPILOGUE")
    assert found_host == found
    assert [None if r is None else [None if x is None else list(x) for x in (r.opening, r.ending)] for r in res_host] == want_results
    d_arena, stride = lib.hash_arena()
    arena = np.zeros(n * stride, dtype=np.uint32)
    capi.check(capi.lib().needle_hip_memcpy_d2h(arena.ctypes.data, d_arena, arena.nbytes))
    kept = int(capi.lib().needle_hip_fingerprint_num_kept(samples, 2))
    want_arena = hashlib.sha256(np.ascontiguousarray(arena.reshape(n, stride)[:, :kept]).tobytes()).hexdigest()
    assert kept == 5441 and found >= n * (n - 1) // 2
    assert n < 300 or found >= (1 << 17)                          # the size at which the epilogue is sharded by default
    assert sum(1 for r in res if r is not None and r.opening is not None) == n
    del lib, gen                                                  # the ranks need the device's memory and process slots

    got = launch("lib", world, str(tmp_path / f"w{world}"), [n, MINUTES], local_ranks=[0] * world,
                 extra_env=dict(env, NEEDLE_HIP_COMM="host"), timeout=900)
    usable = _cpus()
    for g in got:
        assert g["backend"] == "host" and g["world"] == world and g["videos_held"][1] >= n // world
        assert g["host_threads"] == max(1, usable // world)       # the node's CPUs are divided between the ranks
        assert g["arena_digest"] == want_arena, g["rank"]         # rows computed by other ranks included
        sharded = env.get("NEEDLE_HIP_SHARD_EPILOGUE") != "0" and found >= (1 << 17)
        for k, job in enumerate(g["jobs"]):
            assert job["runs"] == found and job["results"] == want_results, (g["rank"], k)
            c = job["comm"]
            assert c["hash_rows"] == n * g["stride"] * 4          # one all-gather of the arena's equal blocks
            assert (c["results"] > 0) == sharded
            # the first two jobs are in flight before any count matrix exists and travel as heads; with the sharded device
            # epilogue the third goes owner-directed: a rank receives the runs of its own videos' pairs and nothing else
            directed = sharded and k == 2
            if directed:
                assert job["held"] == held_count[g["rank"]] < found and job["digest"] == held_digest[g["rank"]], (g["rank"], k)
                assert c["run_heads"] <= 0.6 * 24 * found         # (every run to every rank: >= 24 * found)
            else:
                assert job["held"] == found and job["digest"] == want_digest, (g["rank"], k)
                assert c["run_heads"] >= 24 * found               # every rank receives every run once (+ margins)
        # the first job met the overflow that was forced and repeated its scan; the steady state repeats nothing
        assert g["jobs"][0]["comm"]["scans_repeated"] >= 1 and g["jobs"][2]["comm"]["scans_repeated"] == 0
        assert g["jobs"][2]["comm"]["run_heads"] < 3 * 24 * found + world * 4096
    if env.get("NEEDLE_HIP_SHARD_EPILOGUE") != "0" and found >= (1 << 17):
        print("run bytes received per rank, job 2 (owner-directed):", [g["jobs"][2]["comm"]["run_heads"] for g in got],
              "as heads (job 1):", [g["jobs"][1]["comm"]["run_heads"] for g in got], "24 x runs:", 24 * found)
        a = g["audit"]                                            # this rank's block of hashes, f32 first pass vs f64 kernel
        assert a["mismatches"] == 0 and a["accepted_mismatches"] == 0 and a["max_error_over_s"] <= 8.0
    assert sum(g["host_threads"] for g in got) <= max(usable, world)
    assert sum(g["audit"]["items"] for g in got) == n * kept      # the blocks tile the library: every hash audited once


def test_config3_at_its_stated_count_280_files(tmp_path, capfd):
    """BASELINE.json configs[2] at its stated count: 280 episodes x 24 min as real .needle.dat files (written by the
    product from audio generated in HBM), then needle_audio_comparator_run(analyze=false, display) -- the reference's
    `needle search` -- over the files.  Checked: what it prints for every video against comparator.rs:524-629 through the
    oracle (table-free pair function; each of the sampled videos depends on its 279 pairs), and the COMPLETE run list of
    the 39 060 pairs (same files, read back) against the oracle's table-free scan of all pairs."""
    assert capi.device_count() > 0
    n, threads = int(os.environ.get("NEEDLE_TEST_CONFIG3_EPISODES", "280")), _cpus()
    samples = int(round(24 * 60.0 / 2 * 11025))
    gen = synth.DeviceLibrary(n, samples, 90.0)
    src = capi.Library(n, opening_search_percentage=1.0)
    src.set_pcm_device(gen.pointers(), [samples] * n)
    gen.free()
    src.analyze(0, n, sync=True)
    paths = [str(tmp_path / f"episode-{k:04d}.wav") for k in range(n)]
    for k, p in enumerate(paths):
        src.frame_hashes(k).write(p[:-4] + ".needle.dat")
    del src, gen
    capfd.readouterr()
    capi.Comparator(paths).run(analyze=False, display=True)
    out = capfd.readouterr().out
    blocks = out.strip("\n").split("\n\n")                      # per video: its path, then what was found
    assert len(blocks) == 2 * n and blocks[0::2] == paths
    found_lines = blocks[1::2]
    assert sum(b.startswith("* Opening - \"") for b in found_lines) == n   # planted truth: every episode carries the shared intro

    fhs = [capi.FrameHashes.from_path(p[:-4] + ".needle.dat") for p in paths]
    hashes = [f.opening_data()[0] for f in fhs]
    ts = [f.opening_data()[1].astype(np.uint64) for f in fhs]
    assert all(len(h) == 2897 for h in hashes)
    hd = O.duration_from_secs_f32(0.3)
    sel = sorted(set([0, 1, n // 2, n - 1] + np.random.default_rng(3).choice(n, size=min(36, n), replace=False).tolist()))
    want = O.run_selected_videos(O.Comparator(), hashes, ts, hd, sel, threads=threads)
    for v, w in zip(sel, want):
        o = f'* Opening - "{O.format_time(w.opening[0])}"-"{O.format_time(w.opening[1])}"' if w and w.opening else "* Opening - N/A"
        assert found_lines[v].splitlines()[0] == (o if w else "No opening found."), (v, found_lines[v], o)

    # the complete run list of all pairs, GPU scan vs the oracle's table-free scan (min run 82 = 20 s at 0.246 s per hash)
    pairs = [(i, j, 82) for i in range(n) for j in range(i + 1, n)]
    runs = capi.hamming_runs(hashes, pairs, 10)
    total, ref = O.diagonal_runs_all_pairs(hashes, 10, 82, threads=threads, capacity=4 * len(runs) + 1024)
    got = np.stack([runs["problem"], runs["src_end"], runs["dst_end"], runs["len"]], axis=1).astype(np.uint32)
    assert total == len(got) >= len(pairs)
    assert np.array_equal(got[np.lexsort((got[:, 2], got[:, 1], got[:, 0]))], ref[np.lexsort((ref[:, 2], ref[:, 1], ref[:, 0]))])


def test_config4_streamed_from_pinned_host_pcm(monkeypatch):
    """BASELINE.json configs[4] as worded -- "analyze streamed from host-pinned PCM" -- at 1000 episodes x 45 min:
    29.8 GB of opening-window PCM in pinned host memory (filled from the device generator, so the same audio as the
    resident form), needle_hip_library_stream_pcm in calls of 250 episodes each through the 2 GiB staging arena, then
    the full O(N^2) search.  The hash arena and the results must equal the resident job's over the same audio."""
    assert capi.device_count() > 0
    n = int(os.environ.get("NEEDLE_TEST_STREAMED_EPISODES", "1000"))
    samples = int(round(MINUTES * 60.0 / 2 * 11025))
    gen = synth.DeviceLibrary(n, samples, 90.0)
    lib = capi.Library(n, opening_search_percentage=1.0)
    lib.set_pcm_device(gen.pointers(), [samples] * n)
    cmp = capi.Comparator([f"episode-{k:04d}.wav" for k in range(n)])
    lib.job_begin(cmp, 0)
    want, found = lib.job_end(cmp, 0)
    d_arena, stride = lib.hash_arena()
    want_arena = np.zeros(n * stride, dtype=np.uint32)
    capi.check(capi.lib().needle_hip_memcpy_d2h(want_arena.ctypes.data, d_arena, want_arena.nbytes))
    del lib
    pinned = [capi.PinnedArray(samples) for _ in range(n)]
    for k, p in enumerate(pinned):                              # device -> pinned host: the caller's decoded PCM
        capi.check(capi.lib().needle_hip_memcpy_d2h(p.ptr, gen.pointers()[k], 2 * samples))
    gen.free()
    streamed = capi.Library(n, opening_search_percentage=1.0)
    for first in range(0, n, 250):                               # a decoder hands over a batch at a time
        batch = [pinned[k].array if first <= k < first + 250 else None for k in range(n)]
        streamed.stream_pcm(batch, [samples] * n)
    streamed.job_begin(cmp, 0)
    got, found2 = streamed.job_end(cmp, 0)
    d_arena, stride2 = streamed.hash_arena()
    got_arena = np.zeros(n * stride2, dtype=np.uint32)
    capi.check(capi.lib().needle_hip_memcpy_d2h(got_arena.ctypes.data, d_arena, got_arena.nbytes))
    assert stride2 == stride and np.array_equal(got_arena, want_arena)
    assert found2 == found >= n * (n - 1) // 2
    assert [None if r is None else (r.opening, r.ending) for r in got] == [None if r is None else (r.opening, r.ending) for r in want]
    assert sum(1 for r in got if r is not None and r.opening is not None) == n
