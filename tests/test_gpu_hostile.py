"""-m gpu: the HOSTILE corpus (round 6) against the oracle -- broadband speech-like bodies, noise 20 dB under
the programme, stretches of digital silence and of one sustained chord inside the search window (needle_amd/csrc/synth_hip.hip,
synth.hostile_segments), a tonal shared intro.  The reference visits every cell whatever the content (comparator.rs:176-200);
here the content decides how much the f32 first pass certifies, how many diagonals survive the scan's head rows and whether a
pair's bucket of runs still fits the device epilogue -- none of which may change a result."""
import os

import numpy as np
import pytest

from needle_amd import capi, synth
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _cpus():
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            return max(1, min(len(os.sched_getaffinity(0)), int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, len(os.sched_getaffinity(0)))


def _library(n, samples, intro_s):
    gen = synth.DeviceLibrary(n, samples, intro_s, hostile=True)
    pcm = [gen.episode(k) for k in range(n)]
    lib = capi.Library(n, opening_search_percentage=1.0)
    lib.set_pcm_device(gen.pointers(), [samples] * n)
    return gen, pcm, lib


def _pairs(res):
    return [None if r is None else (r.opening, r.ending) for r in res]


def test_hostile_corpus_hashes_runs_and_results_match_oracle(monkeypatch):
    assert capi.device_count() > 0
    n, samples, threads = 12, int(8 * 60 * 11025), _cpus()       # 12 windows of 8 minutes: 1937 hashes each, 66 pairs
    gen, pcm, lib = _library(n, samples, 45.0)
    seg = gen.segments
    assert (seg[:, 1] > 20 * 11025).sum() >= 5 and (seg[:, 3] > 20 * 11025).sum() >= 3   # the stretches are really there
    for k in range(n):                                            # ... and really silent / really apart from the intro
        so, sl, co, cl = (int(x) for x in seg[k])
        if sl:
            assert not pcm[k][so:so + sl].any()
        io = int(gen.intro_off[k])
        assert so + sl <= io or so >= io + gen.intro_len
        assert cl == 0 or co + cl <= io or co >= io + gen.intro_len
    cmp = capi.Comparator([f"hostile-{k:02d}.wav" for k in range(n)])
    capi.cert_stats(reset=True)
    capi.epilogue_host_fallbacks(reset=True)
    lib.job_begin(cmp, 0)
    res, found = lib.job_end(cmp, 0)
    assert not lib.job_form(0)["device_epilogue"]                 # 66 pairs: the host form ... until the run count is known:
    lib.job_begin(cmp, 1)
    res2, found2 = lib.job_end(cmp, 1)
    assert found == found2 and _pairs(res) == _pairs(res2)
    assert lib.job_form(1)["device_epilogue"] == (found >= 16384)  # (the switch on a measured density, needle_hip.h)
    cs = capi.cert_stats()
    gen.free()

    # (1) every u32 of every episode == the oracle's f64 pipeline (certified or recomputed: more of the latter than on notes)
    hd = O.duration_from_secs_f32(0.3)
    ref = O.analyze_batch(pcm, 1, hd, threads=threads)
    hashes = [lib.frame_hashes(v).opening_data()[0] for v in range(n)]
    for v in range(n):
        assert hashes[v].tolist() == [h for h, _ in ref[v].opening], f"episode {v}"
    silent = [v for v in range(n) if seg[v, 1] > 20 * 11025]
    h0 = hashes[silent[0]]
    first = int(seg[silent[0], 0]) // 1365 // 2 + 12               # a kept hash well inside the silence
    assert len(set(h0[first:first + 40].tolist())) == 1            # digital silence: one constant hash

    # (2) the complete run list == the oracle's table walk over the same hashes (hundreds of runs per silent pair)
    cap = max(4 * found, 1 << 16)
    d_runs, d_count = capi.DeviceBuffer(cap * capi.RUN_DTYPE.itemsize), capi.DeviceBuffer(4)
    lib.search(cmp, 0, lib.num_pairs(), d_runs.ptr, cap, d_count.ptr, sync=True)
    k = int(d_count.to_host(np.uint32, 1)[0])
    assert k == found
    runs = d_runs.to_host(capi.RUN_DTYPE, k)
    total, want = O.diagonal_runs_all_pairs(hashes, 10, 82, threads=threads, capacity=cap)
    assert total == k
    got = np.stack([runs["problem"], runs["src_end"], runs["dst_end"], runs["len"]], axis=1).astype(np.uint32)
    assert np.array_equal(got[np.lexsort((got[:, 2], got[:, 1], got[:, 0]))], want[np.lexsort((want[:, 2], want[:, 1], want[:, 0]))])
    per_pair = np.bincount(runs["problem"].astype(np.int64), minlength=lib.num_pairs())
    assert per_pair.max() > 256, "no pair of silent stretches filled a bucket beyond the device epilogue's limit: not hostile enough"
    # ... and the same list from the MATRIX-PIPE form of the scan (taken by itself from 2048 pairs up; forced here): blocks of equal
    # hashes are where its workgroups hand chains round (a crowd: scan_mfma_kernel.h drain()) and its waves' run buffers overflow
    monkeypatch.setenv("NEEDLE_HIP_SCAN_MFMA", "1")
    lib.search(cmp, 0, lib.num_pairs(), d_runs.ptr, cap, d_count.ptr, sync=True)
    assert capi.scan_last_launch()[0] == 4
    k2 = int(d_count.to_host(np.uint32, 1)[0])
    assert k2 == k
    runs2 = d_runs.to_host(capi.RUN_DTYPE, k2)
    got2 = np.stack([runs2["problem"], runs2["src_end"], runs2["dst_end"], runs2["len"]], axis=1).astype(np.uint32)
    assert np.array_equal(got2[np.lexsort((got2[:, 2], got2[:, 1], got2[:, 0]))], want[np.lexsort((want[:, 2], want[:, 1], want[:, 0]))])
    monkeypatch.delenv("NEEDLE_HIP_SCAN_MFMA")

    # (3) final results == comparator.rs:524-629 on the oracle (full tables, heap order, clustering), through both entry points
    want_res = O.run_with_frame_hashes(O.Comparator(), ref, threads=threads)
    assert _pairs(res) == _pairs(want_res)
    fhs = [lib.frame_hashes(v) for v in range(n)]
    assert _pairs(cmp.run_with_frame_hashes(fhs)) == _pairs(want_res)
    print("hostile corpus: runs", found, "largest bucket", int(per_pair.max()), "recomputed items",
          cs["items_recomputed"], "of", cs["items"], "chunks", cs["chunks_recomputed"], "of", cs["chunks"],
          "epilogue host fallbacks", capi.epilogue_host_fallbacks())


def test_hostile_corpus_on_the_device_epilogue(monkeypatch):
    """A library large enough for the DEVICE epilogue (16 384 sequence pairs up) whose silent pairs overflow a lane's bucket:
    those buckets are a workgroup's (pair_entries_large_kernel) and nothing falls back; the results equal the host form's,
    and with the workgroup kernel switched off the job IS handed to the host form and counted."""
    assert capi.device_count() > 0
    n, samples = 182, int(3 * 60 * 11025)                         # 16 471 pairs of 3-minute windows (727 hashes)
    gen, _, lib = _library(n, samples, 25.0)
    gen.free()
    cmp = capi.Comparator([f"hostile-{k:03d}.wav" for k in range(n)])
    capi.epilogue_host_fallbacks(reset=True)
    lib.job_begin(cmp, 0)
    res, found = lib.job_end(cmp, 0)
    assert capi.epilogue_host_fallbacks() == 0
    monkeypatch.setenv("NEEDLE_HIP_EPILOGUE_NO_LARGE", "1")
    lib.job_begin(cmp, 1)
    res_flagged, found_flagged = lib.job_end(cmp, 1)
    assert capi.epilogue_host_fallbacks(reset=True) >= 1, "no bucket beyond a lane's limit: not hostile enough"
    monkeypatch.delenv("NEEDLE_HIP_EPILOGUE_NO_LARGE")
    monkeypatch.setenv("NEEDLE_HIP_DEVICE_EPILOGUE", "0")
    lib.job_begin(cmp, 0)
    res_host, found_host = lib.job_end(cmp, 0)
    assert found == found_host == found_flagged
    assert _pairs(res) == _pairs(res_host) == _pairs(res_flagged)
    assert sum(1 for r in res if r is not None and r.opening is not None) >= n // 2


def test_run_density_moves_a_small_librarys_epilogue_to_the_device(monkeypatch):
    """The headline shape on the hostile corpus: 378 pairs -- far below the 16 384 from which the epilogue goes to the device by
    itself -- with 41 000 runs (852 on the tonal corpus).  A job enqueued knowing that the library's last scan found 16 384
    runs or more takes the device form (needle_hip_library_job_form) -- the first job already when its slab overflowed and the
    scan was repeated --; the results are the same as with the device form switched off; the tonal library of the same
    shape stays with the host form."""
    assert capi.device_count() > 0
    n, samples = 28, int(12 * 60 * 11025)
    gen, _, lib = _library(n, samples, 90.0)
    gen.free()
    cmp = capi.Comparator([f"hostile-{k:02d}.wav" for k in range(n)])
    capi.epilogue_host_fallbacks(reset=True)
    lib.job_begin(cmp, 0)
    res0, found0 = lib.job_end(cmp, 0)
    assert found0 >= 16384
    lib.job_begin(cmp, 1)
    res1, found1 = lib.job_end(cmp, 1)
    assert found1 == found0 and lib.job_form(1)["device_epilogue"] and capi.epilogue_host_fallbacks() == 0
    assert _pairs(res1) == _pairs(res0)
    monkeypatch.setenv("NEEDLE_HIP_DEVICE_EPILOGUE", "0")
    lib.job_begin(cmp, 0)
    res2, _ = lib.job_end(cmp, 0)
    assert not lib.job_form(0)["device_epilogue"] and _pairs(res2) == _pairs(res0)
    assert sum(1 for r in res0 if r is not None and r.opening is not None) == n
    monkeypatch.delenv("NEEDLE_HIP_DEVICE_EPILOGUE")
    tonal = synth.DeviceLibrary(n, samples, 90.0)
    lib_t = capi.Library(n, opening_search_percentage=1.0)
    lib_t.set_pcm_device(tonal.pointers(), [samples] * n)
    tonal.free()
    for slot in (0, 1, 0):
        lib_t.job_begin(cmp, slot)
        _, found_t = lib_t.job_end(cmp, slot)
        assert found_t < 16384 and not lib_t.job_form(slot)["device_epilogue"]
