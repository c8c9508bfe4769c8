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
