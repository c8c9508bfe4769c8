"""-m gpu: the per-video epilogue ON THE DEVICE (needle_amd/csrc/epilogue.hip) against the host form (comparator.cpp)
and the oracle (comparator.rs:191-249,405-515,583-626 restated in oracle/ora_needle.c).  The device form is what a
library-scale job or search call uses (>= 16 384 sequence pairs, or NEEDLE_HIP_DEVICE_EPILOGUE=1); both forms see the same run list.
What can go wrong is ORDER: the reverse table walk, BinaryHeap's backing-array order, the candidate numbering that
breaks ties between equal (f32) scores -- so the inputs here are built to tie: the same segment planted bit-identically
in many videos, competing segments of equal length, several runs per pair, endings, padding."""
import os

import numpy as np
import pytest

from needle_amd import capi, synth
from oracle import oracle as O

pytestmark = pytest.mark.gpu
NS = O.NS


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert capi.device_count() > 0, "GPU tests need a HIP device (the product has no CPU fallback)"


def _as(rs):
    return [None if r is None else (r.opening, r.ending) for r in rs]


def _job(lib, cmp, monkeypatch, device):
    monkeypatch.setenv("NEEDLE_HIP_DEVICE_EPILOGUE", "1" if device else "0")
    lib.job_begin(cmp, 0)
    res, runs = lib.job_end(cmp, 0)
    return _as(res), runs


def _planted_library(rng, n, kept, endings, exact):
    """Hash rows with shared segments: `exact` plants bit-identical copies (equal simhashes, equal lengths: ties)."""
    rows = []
    seg_a, seg_b, seg_c = (rng.integers(0, 2 ** 32, L, dtype=np.uint64).astype(np.uint32) for L in (110, 110, 95))
    for v in range(n):
        regions = []
        for r in range(2 if endings else 1):
            h = rng.integers(0, 2 ** 32, kept[v], dtype=np.uint64).astype(np.uint32)
            for seg, every, base in ((seg_a, 1, 7), (seg_b, 2, 260), (seg_c, 3, 420)):
                if v % every == 0 and base + 13 * (v % 5) + len(seg) < kept[v]:
                    a = base + 13 * (v % 5) + 31 * r
                    if a + len(seg) >= kept[v]:
                        continue
                    flips = np.zeros(len(seg), dtype=np.uint32) if exact else \
                        (np.uint32(1) << rng.integers(0, 32, len(seg)).astype(np.uint32)) * (rng.random(len(seg)) < 0.5)
                    h[a:a + len(seg)] = seg ^ flips
            regions.append(h)
        rows.append(regions)
    return rows


@pytest.mark.parametrize("n,endings,exact,min_s,padding,threshold", [
    (24, False, True, 20, 0.0, 10), (24, True, True, 15, 0.0, 10), (17, False, False, 10, 0.0, 10),
    (9, True, False, 20, 1.5, 6), (31, False, True, 25, 0.25, 12), (12, True, True, 5, 0.0, 10)])
def test_device_epilogue_on_planted_hashes_equals_host_and_oracle(monkeypatch, n, endings, exact, min_s, padding, threshold):
    """Hashes written straight into the library's arena (the search and the epilogue start from hashes): identical
    segments in many videos give every pair several runs and many candidates exactly equal scores."""
    rng = np.random.default_rng(1000 + n)
    seconds = [170.0 + 11.0 * (v % 4) for v in range(n)]
    lens = [int(round(s * synth.RATE)) for s in seconds]
    lib = capi.Library(n)
    if endings:
        lib.include_endings()
    lib.stream_pcm([np.zeros(v, dtype=np.int16) for v in lens], lens)      # geometry only: the rows are overwritten below
    fhs0 = [lib.frame_hashes(v) for v in range(n)]
    kept = [(len(f.opening_data()[0]), len(f.ending_data()[0]) if endings else 0) for f in fhs0]
    rows = _planted_library(rng, n, [k[0] for k in kept], endings, exact)
    d_arena, stride = lib.hash_arena()
    R = lib.rows_per_video()
    for v in range(n):
        for r in range(R):
            h = rows[v][r][: kept[v][r]] if r == 0 else rows[v][r][: kept[v][1]]
            h = np.ascontiguousarray(h, dtype=np.uint32)
            capi.check(capi.lib().needle_hip_memcpy_h2d(d_arena + 4 * (v * R + r) * stride, h.ctypes.data, h.nbytes))
    cmp = capi.Comparator([f"v{v}.wav" for v in range(n)], include_endings=endings, min_opening_duration=min_s,
                          min_ending_duration=min_s, hash_match_threshold=threshold, time_padding=padding)
    host, runs_h = _job(lib, cmp, monkeypatch, device=False)
    dev, runs_d = _job(lib, cmp, monkeypatch, device=True)
    assert runs_h == runs_d > 0
    assert dev == host
    hd = O.duration_from_secs_f32(0.3)
    ofh = []
    for v in range(n):
        f = lib.frame_hashes(v)
        op = list(zip(f.opening_data()[0].tolist(), f.opening_data()[1].tolist()))
        en = list(zip(f.ending_data()[0].tolist(), f.ending_data()[1].tolist())) if endings else []
        assert [h for h, _ in op] == rows[v][0][: kept[v][0]].tolist()
        ofh.append(O.FrameHashes(op, en, hd, ""))
    want = O.run_with_frame_hashes(O.Comparator(include_endings=endings, hash_match_threshold=threshold,
                                                min_opening_duration=min_s * NS, min_ending_duration=min_s * NS,
                                                time_padding=O.duration_from_secs_f32(padding)), ofh, threads=8)
    assert dev == _as(want)
    assert sum(1 for r in dev if r is not None and r[0] is not None) >= n // 2


def test_pairs_with_hundreds_of_runs_are_a_workgroups_or_the_hosts(monkeypatch):
    """two stretches of one repeated hash (silence, a sustained tone) give a pair ~2 S runs -- every diagonal of
    an S x S block is one.  The device form orders a pair's runs in ONE lane, quadratically; a bucket beyond a lane's 24 runs goes to a
    WORKGROUP (round 6: pair_entries_large_kernel, sort + heap on packed keys in LDS), and with that kernel switched off the
    job's results come from the host form, counted.  Same results every way, and equal to the oracle's."""
    rng = np.random.default_rng(77)
    n, S = 5, 230
    lens = [int(round(400.0 * synth.RATE))] * n                         # ~800 kept hashes in the opening half
    lib = capi.Library(n)
    lib.stream_pcm([np.zeros(v, dtype=np.int16) for v in lens], lens)
    kept = len(lib.frame_hashes(0).opening_data()[0])
    assert kept > 500
    d_arena, stride = lib.hash_arena()
    rows = []
    for v in range(n):
        h = rng.integers(0, 2 ** 32, kept, dtype=np.uint64).astype(np.uint32)
        if v < 3:
            h[40 + 5 * v: 40 + 5 * v + S] = np.uint32(0x5A5A1234)          # one hash, S times: "silence"
        seg = np.arange(90, dtype=np.uint32) * np.uint32(2654435761)
        h[350: 350 + len(seg)] = seg                                        # and an ordinary shared segment behind it
        rows.append(h)
        capi.check(capi.lib().needle_hip_memcpy_h2d(d_arena + 4 * v * stride, h.ctypes.data, h.nbytes))
    cmp = capi.Comparator([f"v{v}.wav" for v in range(n)], min_opening_duration=15)
    host, runs_h = _job(lib, cmp, monkeypatch, device=False)
    capi.epilogue_host_fallbacks(reset=True)
    dev, runs_d = _job(lib, cmp, monkeypatch, device=True)
    assert capi.epilogue_host_fallbacks() == 0                              # the workgroup kernel took the silent pairs
    assert runs_h == runs_d > 3 * 2 * (S - 70)                              # the silent pairs alone: > 256 runs each
    assert dev == host
    monkeypatch.setenv("NEEDLE_HIP_EPILOGUE_NO_LARGE", "1")                 # ... and without it: flagged, host form, counted
    dev2, runs_d2 = _job(lib, cmp, monkeypatch, device=True)
    monkeypatch.delenv("NEEDLE_HIP_EPILOGUE_NO_LARGE")
    assert capi.epilogue_host_fallbacks(reset=True) >= 1 and dev2 == host and runs_d2 == runs_h
    hd = O.duration_from_secs_f32(0.3)
    ofh = []
    for v in range(n):
        f = lib.frame_hashes(v)
        ofh.append(O.FrameHashes(list(zip(f.opening_data()[0].tolist(), f.opening_data()[1].tolist())), [], hd, ""))
    want = O.run_with_frame_hashes(O.Comparator(min_opening_duration=15 * NS), ofh, threads=8)
    assert dev == _as(want)


def test_a_bucket_beyond_the_workgroup_kernels_limit_is_still_the_hosts(monkeypatch):
    """Two windows that are silent for 17 minutes each: ~8 250 runs in one pair's bucket, more than pair_entries_large_kernel holds
    in LDS (8192).  The job is handed to the host form and counted (needle_hip_epilogue_host_fallbacks); same results."""
    rng = np.random.default_rng(78)
    n, S = 3, 4210
    lens = [int(round(2230.0 * synth.RATE))] * n                        # ~4 500 kept hashes in the opening half
    lib = capi.Library(n)
    lib.stream_pcm([np.zeros(v, dtype=np.int16) for v in lens], lens)
    kept = len(lib.frame_hashes(0).opening_data()[0])
    assert kept > S + 100
    d_arena, stride = lib.hash_arena()
    for v in range(n):
        h = rng.integers(0, 2 ** 32, kept, dtype=np.uint64).astype(np.uint32)
        if v < 2:
            h[30 + 7 * v: 30 + 7 * v + S] = np.uint32(0x0F1E2D3C)
        capi.check(capi.lib().needle_hip_memcpy_h2d(d_arena + 4 * v * stride, h.ctypes.data, h.nbytes))
    cmp = capi.Comparator([f"v{v}.wav" for v in range(n)])
    host, runs_h = _job(lib, cmp, monkeypatch, device=False)
    capi.epilogue_host_fallbacks(reset=True)
    dev, runs_d = _job(lib, cmp, monkeypatch, device=True)
    assert runs_h == runs_d > 8192 and capi.epilogue_host_fallbacks(reset=True) >= 1
    assert dev == host and dev[0] is not None and dev[0][0] is not None


def test_device_epilogue_on_audio_with_two_jobs_in_flight(monkeypatch):
    eps = [synth.make_episode(k, 100.0 + 7.0 * k, 22.0, 21.0) for k in range(6)]
    lens = [len(e.pcm) for e in eps]
    lib = capi.Library(6).include_endings()
    lib.set_pcm([e.pcm for e in eps], lens)
    cmp = capi.Comparator([f"ep{k}.wav" for k in range(6)], min_opening_duration=10, min_ending_duration=10, include_endings=True)
    host, _ = _job(lib, cmp, monkeypatch, device=False)
    monkeypatch.setenv("NEEDLE_HIP_DEVICE_EPILOGUE", "1")
    monkeypatch.setenv("NEEDLE_HIP_SLAB_RUNS", "4")               # (a fresh library: its first job overflows and is redone)
    lib2 = capi.Library(6).include_endings()
    lib2.set_pcm([e.pcm for e in eps], lens)
    lib2.job_begin(cmp, 0)
    lib2.job_begin(cmp, 1)
    a = _as(lib2.job_end(cmp, 0)[0])
    lib2.job_begin(cmp, 0)
    b = _as(lib2.job_end(cmp, 1)[0])
    c = _as(lib2.job_end(cmp, 0)[0])
    assert a == b == c == host
    assert all(r is not None and r[0] is not None and r[1] is not None for r in a)


def test_padding_larger_than_the_match_fails_like_the_host_form(monkeypatch):
    eps = synth.make_library(4, 90.0, 20.0)
    lib = capi.Library(4)
    lib.set_pcm([e.pcm for e in eps], [len(e.pcm) for e in eps])
    cmp = capi.Comparator([f"ep{k}.wav" for k in range(4)], min_opening_duration=10, time_padding=4000.0)
    for device in (False, True):
        monkeypatch.setenv("NEEDLE_HIP_DEVICE_EPILOGUE", "1" if device else "0")
        lib.job_begin(cmp, 0)
        with pytest.raises(capi.NeedleError, match="overflow when subtracting durations"):
            lib.job_end(cmp, 0)


def test_comparator_objects_use_the_device_epilogue_with_display_and_skip_files(tmp_path, capfd, monkeypatch):
    """needle_audio_comparator_run (analyze=false) through the device epilogue: what it prints and the skip files it writes
    equal the host form's, video by video in the reference's order (comparator.rs:593-626), endings and padding included."""
    eps = [synth.make_episode(k, 95.0 + 5.0 * (k % 3), 21.0, 19.0) for k in range(9)]
    paths = [str(tmp_path / f"ep{k}.wav") for k in range(9)]
    for p, e in zip(paths, eps):
        synth.write_wav(p, e.pcm)
    capi.Analyzer.from_files(paths).with_include_endings(True).run(0.3, persist=True)
    outs = {}
    for device in ("0", "1"):
        monkeypatch.setenv("NEEDLE_HIP_DEVICE_EPILOGUE", device)
        for p in paths:
            skip = p[:-4] + ".needle.skip.json"
            if os.path.exists(skip):
                os.unlink(skip)
        capfd.readouterr()
        cmp = capi.Comparator(paths, include_endings=True, min_opening_duration=10, min_ending_duration=10, time_padding=0.5)
        cmp.run(analyze=False, display=True, write_skip_files=True)
        text = capfd.readouterr().out
        skips = {p: open(p[:-4] + ".needle.skip.json").read() for p in paths if os.path.exists(p[:-4] + ".needle.skip.json")}
        outs[device] = (text, skips)
    assert outs["0"] == outs["1"]
    assert outs["1"][0].count('* Opening - "') == 9 and len(outs["1"][1]) == 9
