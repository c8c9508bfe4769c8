"""GPU tests (-m gpu) of the job pipeline, the streaming analyzer and the N-rank path of libneedle_capi.so
(include/needle_hip.h: needle_hip_library_job_begin/_end, needle_hip_library_stream_pcm, needle_hip_comm_*).

The oracle is the checker throughout.  Multi-rank runs are real processes (tests/comm_worker.py gpu): over the
host-staged transport with every rank on device 0 (any box), over RCCL with one rank (any box: the calls go through
ncclAllGather) and over RCCL with 2 / 4 / 8 ranks where that many devices exist (skipped below)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from needle_amd import capi, synth
from oracle import oracle as O
from tests.test_comm_cpu import launch

pytestmark = pytest.mark.gpu
NS = O.NS
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert capi.device_count() > 0, "GPU tests need a HIP device (the product has no CPU fallback)"


def _oracle(eps, min_opening=10, endings=False):
    hd = O.duration_from_secs_f32(0.3)
    if not endings:
        ref = O.analyze_batch([e.pcm[: len(e.pcm) // 2] for e in eps], 1, hd)
    else:
        ref = []
        for e in eps:
            dur = O.duration_from_secs_f64(len(e.pcm) * (1.0 / 11025.0))
            n_open = O.duration_mul_f32(dur, 0.5) * 11025 // NS
            seek = O.duration_mul_f32(dur, float(np.float32(1.0) - np.float32(0.25)))
            first = seek * 11025 // NS
            op = O.step_and_timestamp(O.fingerprint(e.pcm[:n_open]), hd)
            en = O.step_and_timestamp(O.fingerprint(e.pcm[first:]), hd, seek_to_ns=seek)
            ref.append(O.FrameHashes(op, en, hd, ""))
    want = O.run_with_frame_hashes(O.Comparator(include_endings=endings, min_opening_duration=min_opening * NS,
                                                min_ending_duration=min_opening * NS), ref)
    return ref, [None if r is None else [None if r.opening is None else list(r.opening),
                                         None if r.ending is None else list(r.ending)] for r in want]


def _as_json(rs):
    return [None if r is None else [None if r.opening is None else list(r.opening),
                                    None if r.ending is None else list(r.ending)] for r in rs]


@pytest.fixture(scope="module")
def lib7():
    return synth.make_library(7, 90.0, 20.0)


@pytest.mark.parametrize("slab_runs", [None, 4])
def test_job_api_single_rank_equals_oracle(lib7, monkeypatch, slab_runs):
    """job_begin / job_end without a communicator: two jobs in flight, results = the oracle's; a 4-run slab forces
    the overflow -> grow -> rescan path on the first job."""
    if slab_runs:
        monkeypatch.setenv("NEEDLE_HIP_SLAB_RUNS", str(slab_runs))
    n = len(lib7)
    lib = capi.Library(n)
    lib.set_pcm([e.pcm for e in lib7], [len(e.pcm) for e in lib7])
    cmp = capi.Comparator([f"ep{k}.wav" for k in range(n)], min_opening_duration=10)
    ref, want = _oracle(lib7)
    lib.job_begin(cmp, 0)
    lib.job_begin(cmp, 1)
    with pytest.raises(capi.NeedleError):
        lib.job_begin(cmp, 1)                                  # slot still pending
    r0, k0 = lib.job_end(cmp, 0)
    lib.job_begin(cmp, 0)
    r1, k1 = lib.job_end(cmp, 1)
    r2, k2 = lib.job_end(cmp, 0)
    assert _as_json(r0) == _as_json(r1) == _as_json(r2) == want
    assert k0 == k1 == k2 >= n * (n - 1) // 2
    for v in range(n):
        assert lib.frame_hashes(v).opening_data()[0].tolist() == [h for h, _ in ref[v].opening]


@pytest.mark.parametrize("source", ["pinned", "pageable-ring", "pageable-small"])
def test_stream_pcm_equals_resident_analyze_and_oracle(lib7, monkeypatch, source):
    """needle_hip_library_stream_pcm: uploads overlapped with per-group fingerprint launches, several launch groups
    and several device batches, from pinned memory (read in place), through the slab ring, and as plain copies."""
    monkeypatch.setenv("NEEDLE_HIP_LAUNCH_GROUP_BYTES", str(1_500_000))       # ~1.5 streams per group
    monkeypatch.setenv("NEEDLE_HIP_MAX_BATCH_VALUES", str(2_000_000))         # ~4 streams per device batch
    if source == "pageable-ring":
        monkeypatch.setenv("NEEDLE_HIP_RING_UPLOAD_MIN_BYTES", "0")
        monkeypatch.setenv("NEEDLE_HIP_UPLOAD_SLAB_BYTES", str(300_000))
    n = len(lib7)
    lens = [len(e.pcm) for e in lib7]
    keep = [capi.PinnedArray(v) for v in lens] if source == "pinned" else None
    if keep:
        for p, e in zip(keep, lib7):
            p.array[:] = e.pcm
    arrays = [p.array for p in keep] if keep else [e.pcm for e in lib7]
    ref, want = _oracle(lib7)
    cmp = capi.Comparator([f"ep{k}.wav" for k in range(n)], min_opening_duration=10)
    lib = capi.Library(n)
    for rep in range(2):                                       # a second pass reuses arena and staging buffers
        lib.stream_pcm(arrays, lens)
        lib.job_begin(cmp, 0)
        res, _ = lib.job_end(cmp, 0)
        assert _as_json(res) == want
        for v in range(n):
            assert lib.frame_hashes(v).opening_data()[0].tolist() == [h for h, _ in ref[v].opening], (source, rep, v)
    with pytest.raises(capi.NeedleError):
        lib.analyze(0, n)                                      # nothing resident to analyze again


def test_stream_pcm_with_endings_and_ragged_lengths():
    eps = [synth.make_episode(k, 100.0 + 7.0 * k, 22.0, 21.0) for k in range(4)]
    lens = [len(e.pcm) for e in eps]
    ref, want = _oracle(eps, min_opening=10, endings=True)
    lib = capi.Library(4).include_endings()
    lib.stream_pcm([e.pcm for e in eps], lens)
    cmp = capi.Comparator([f"ep{k}.wav" for k in range(4)], min_opening_duration=10, min_ending_duration=10,
                          include_endings=True)
    lib.job_begin(cmp, 0)
    res, _ = lib.job_end(cmp, 0)
    assert _as_json(res) == want
    for v in range(4):
        fh = lib.frame_hashes(v)
        assert fh.opening_data()[0].tolist() == [h for h, _ in ref[v].opening]
        assert fh.ending_data()[0].tolist() == [h for h, _ in ref[v].ending]
        assert fh.ending_data()[1].tolist() == [t for _, t in ref[v].ending]


def _check_ranks(got, lib, want, ref, world, backend):
    n = len(lib)
    for g in got:
        assert g["backend"] == backend and g["world"] == world
        # hash-sharded fingerprinting: every rank has a block of hashes to compute, whatever n and world are
        assert g["stft_ms"] > 0 and g["videos_held"][1] >= 1, g["rank"]
        for job in g["jobs"]:
            assert job["results"] == want
            assert job["runs"] == got[0]["jobs"][0]["runs"] >= n * (n - 1) // 2
        for v in range(n):
            assert g["hashes"][v] == [h for h, _ in ref[v].opening]     # rows fingerprinted by other ranks included


@pytest.mark.parametrize("world,shard_epilogue,slab_runs", [(2, "0", None), (2, "1", None), (3, "1", 4), (2, "0", 4),
                                                            (5, "1", None), (3, "directed", None), (4, "directed", 4),
                                                            (3, "directed-overflow", None)])
def test_ranks_over_host_transport_on_one_gpu(lib7, tmp_path, world, shard_epilogue, slab_runs):
    """The complete N-rank path of the library -- own-block analyze, all-gather of rows, own pair range, all-gather of
    run slabs, (sharded) epilogue, all-gather of results, two jobs in flight -- between real processes that share
    device 0, over the host-staged transport.  The fingerprinting is cut by hashes, not by videos: 7 episodes on 5 ranks
    are 1.4 episodes' worth of frames each (whole videos would be 2, 2, 2, 1, 0), and every rank's STFT kernel runs."""
    ref, want = _oracle(lib7)
    env = {"NEEDLE_HIP_COMM": "host", "NEEDLE_HIP_SHARD_EPILOGUE": shard_epilogue}
    if shard_epilogue.startswith("directed"):   # sharded DEVICE epilogue at this small size: the third job's runs travel owner-directed
        env.update(NEEDLE_HIP_SHARD_EPILOGUE="1", NEEDLE_HIP_DEVICE_EPILOGUE="1")
    if shard_epilogue == "directed-overflow":   # ... into blocks of 4 runs the first time: every rank sees the counts, sizes grow, once more
        env.update(NEEDLE_HIP_TEST_DIRECTED_CAP="4")
    if slab_runs:
        env["NEEDLE_HIP_SLAB_RUNS"] = str(slab_runs)
    got = launch("gpu", world, str(tmp_path / "r"), [len(lib7), 90.0], extra_env=env, local_ranks=[0] * world)
    _check_ranks(got, lib7, want, ref, world, "host")
    if shard_epilogue.startswith("directed"):
        # jobs 0 and 1 were in flight before any count matrix existed (heads: a rank holds every run); job 2 went
        # owner-directed: a rank holds the runs of its own videos' pairs -- all ranks together every run at least once
        for g in got:
            assert g["jobs"][0]["comm"]["held"] == g["jobs"][0]["runs"] == g["jobs"][1]["comm"]["held"]
            assert 0 < g["jobs"][2]["comm"]["held"] <= g["jobs"][2]["runs"]
        assert sum(g["jobs"][2]["comm"]["held"] for g in got) >= got[0]["jobs"][2]["runs"]
        assert any(g["jobs"][2]["comm"]["held"] < g["jobs"][2]["runs"] for g in got)
        repeated = [g["jobs"][2]["comm"]["scans_repeated"] for g in got]
        assert repeated == [1 if shard_epilogue == "directed-overflow" else 0] * world


def test_device_epilogue_failing_on_one_rank_keeps_the_collective_shape(lib7, tmp_path):
    """the device epilogue is requested, the pair-count rule says "sharded" (threshold lowered for the test), and
    the enqueue fails on rank 1 alone.  That rank falls back to the host form -- for ITS block of videos, and it still enters
    the all-gather of results the other ranks enter (it used to re-decide by the run count, skip it, and hang the job)."""
    ref, want = _oracle(lib7)
    env = {"NEEDLE_HIP_COMM": "host", "NEEDLE_HIP_DEVICE_EPILOGUE": "1", "NEEDLE_HIP_SHARD_EPILOGUE_PAIRS": "1",
           "NEEDLE_HIP_TEST_EPILOGUE_FAIL_RANK": "1"}
    got = launch("gpu", 3, str(tmp_path / "f"), [len(lib7), 90.0], extra_env=env, local_ranks=[0] * 3)
    _check_ranks(got, lib7, want, ref, 3, "host")


def test_ranks_with_endings_over_host_transport(tmp_path):
    eps = [synth.make_episode(k, 90.0, 20.0) for k in range(5)]          # what comm_worker.py synthesises
    ref, want = _oracle(eps, endings=True)
    got = launch("gpu", 2, str(tmp_path / "e"), [5, 90.0], local_ranks=[0, 0],
                 extra_env={"NEEDLE_HIP_COMM": "host", "NEEDLE_HIP_SHARD_EPILOGUE": "1", "NEEDLE_TEST_ENDINGS": "1"})
    for g in got:
        for job in g["jobs"]:
            assert job["results"] == want
    # the same with the device epilogue: two comparator regions per pair, the third job's runs owner-directed
    got = launch("gpu", 2, str(tmp_path / "d"), [5, 90.0], local_ranks=[0, 0],
                 extra_env={"NEEDLE_HIP_COMM": "host", "NEEDLE_HIP_SHARD_EPILOGUE": "1", "NEEDLE_TEST_ENDINGS": "1",
                            "NEEDLE_HIP_DEVICE_EPILOGUE": "1"})
    for g in got:
        for job in g["jobs"]:
            assert job["results"] == want
        assert g["jobs"][2]["comm"]["held"] <= g["jobs"][2]["runs"] == g["jobs"][0]["comm"]["held"]


@pytest.mark.parametrize("world,hash_duration,step", [(3, 0.3, 2), (4, 0.15, 1), (3, 0.4, 3)])
def test_hash_sharding_with_ragged_lengths_and_other_steps(tmp_path, world, hash_duration, step):
    """The ranks' blocks of hashes cut rows of DIFFERENT lengths at arbitrary columns, and the sub-window a block needs
    depends on the step between kept hashes (hash k <- frames k step .. k step + 19): episodes of 60 .. 82 s, steps 1,
    2 and 3, three and four ranks on one GPU; every rank ends up with the hashes one rank computes alone = the oracle's."""
    n = 7
    eps = [synth.make_episode(k, 60.0 + 3.7 * k, 20.0) for k in range(n)]
    hd = O.duration_from_secs_f32(hash_duration)
    ref = [O.FrameHashes(O.step_and_timestamp(O.fingerprint(e.pcm[: len(e.pcm) // 2]), hd), [], hd, "") for e in eps]
    assert len(ref[0].opening) == -(-O.num_items(len(eps[0].pcm) // 2) // step)
    want = O.run_with_frame_hashes(O.Comparator(min_opening_duration=10 * NS), ref)
    got = launch("gpu", world, str(tmp_path / "h"), [n, 60.0], local_ranks=[0] * world,
                 extra_env={"NEEDLE_HIP_COMM": "host", "NEEDLE_TEST_RAGGED": "1", "NEEDLE_TEST_HASH_DURATION": str(hash_duration)})
    for g in got:
        assert g["stft_ms"] > 0
        for v in range(n):
            assert g["hashes"][v] == [h for h, _ in ref[v].opening], (g["rank"], v)
        for job in g["jobs"]:
            assert job["results"] == _as_json(want)


def test_rccl_communicator_with_one_rank(lib7, tmp_path):
    """librccl loaded on demand, ncclCommInitRank, and -- forced -- every collective of a job through ncclAllGather
    with a single rank: the RCCL call path on a one-GPU box."""
    ref, want = _oracle(lib7)
    got = launch("gpu", 1, str(tmp_path / "c"), [len(lib7), 90.0],
                 extra_env={"NEEDLE_HIP_COMM_FORCE_COLLECTIVES": "1", "NEEDLE_HIP_SHARD_EPILOGUE": "1"})
    _check_ranks(got, lib7, want, ref, 1, "rccl")


@pytest.mark.parametrize("world", [2, 4, 8])
def test_rccl_ranks_bit_identical_results(lib7, tmp_path, world):
    """G in {2, 4, 8} over RCCL / xGMI, one process per GPU: results identical to one rank's and to the oracle's."""
    if capi.device_count() < world:
        pytest.skip(f"needs {world} GPUs, this box has {capi.device_count()}")
    ref, want = _oracle(lib7)
    for shard in ("0", "1"):
        got = launch("gpu", world, str(tmp_path / f"g{world}_{shard}"), [len(lib7), 90.0],
                     extra_env={"NEEDLE_HIP_SHARD_EPILOGUE": shard})
        _check_ranks(got, lib7, want, ref, world, "rccl")


def test_bench_gpus_flag_spawns_the_ranks_itself(tmp_path):
    """`python bench.py --gpus 2` with no launcher environment starts two rank processes and reports n_gpus = 2
    (here both on device 0 over the host-staged transport; on a multi-GPU box the default transport is RCCL)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["NEEDLE_HIP_COMM"] = "host"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--episodes", "5", "--minutes", "2",
                          "--intro-seconds", "30", "--steps", "3", "--warmup", "1", "--no-extras", "--no-cpu-baseline"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["config"]["comm"] == "host" and line["detected"] == 5
    assert line["steps"] == 3 and line["value"] > 0


def _bench(args, env_extra, launcher=None, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(env_extra)
    cmd = (launcher or [sys.executable]) + [os.path.join(ROOT, "bench.py")] + args
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0]), out.stderr


@pytest.mark.parametrize("episodes", [28, 16])
def test_bench_five_ranks_share_one_gpu_over_the_host_transport(episodes):
    """BASELINE.json configs[3]'s code path with as many ranks as a one-GPU box allows next to the test process (its
    guard stops a run with more than 6 processes on the GPU): bench.py starts a supervisor + worker per rank, every
    worker drives device 0, collectives go over the host-staged transport.  The fingerprinting is cut by hashes: 28
    episodes are 5.6 episodes' worth per rank; 16 episodes 3.2 each -- by whole videos rank 4 would own none (4, 4, 4,
    4, 0), as rank 7 of an 8-rank job over 28 episodes would."""
    line, err = _bench(["--gpus", "5", "--episodes", str(episodes), "--minutes", "2", "--intro-seconds", "30", "--steps", "3",
                        "--warmup", "1", "--no-extras", "--no-cpu-baseline", "--launch-timeout", "300"],
                       {"NEEDLE_HIP_COMM": "host"})
    assert line["n_gpus"] == 5 and line["config"]["comm"] == "host" and line["detected"] == episodes
    assert line["config"]["pairs"] == episodes * (episodes - 1) // 2 and line["value"] > 0
    assert err.count("communicator up over host") == 5
    total = 2 * 60 * 11025
    lib = capi.Library(episodes)
    assert all(lib.rank_videos([total] * episodes, 5, r)[1] >= 1 for r in range(5))


def test_bench_under_torchrun_the_drivers_form():
    """`python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2`: each launched process becomes a
    supervisor (never touching the GPU) with a worker child; RANK / LOCAL_RANK / WORLD_SIZE come from the launcher.
    Two ranks on the one device, so the transport is the host-staged one."""
    pytest.importorskip("torch")
    launcher = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                "127.0.0.1", "--master-port", "29517"]
    line, err = _bench(["--gpus", "2", "--episodes", "6", "--minutes", "2", "--intro-seconds", "30", "--steps", "3",
                        "--warmup", "1", "--no-extras", "--no-cpu-baseline"], {"NEEDLE_HIP_COMM": "host"}, launcher)
    assert line["n_gpus"] == 2 and line["detected"] == 6 and line["config"]["comm"] == "host"


def test_rccl_refusing_two_ranks_per_device_falls_back_in_process():
    """Default transport (RCCL) with two ranks on ONE device: ncclCommInitRank fails (or is refused) on this box, every
    rank votes through the rendezvous and all switch to the host transport inside the same processes; the line says
    which transport carried the job."""
    line, err = _bench(["--gpus", "2", "--episodes", "5", "--minutes", "2", "--intro-seconds", "30", "--steps", "2",
                        "--warmup", "1", "--no-extras", "--no-cpu-baseline", "--launch-timeout", "90"], {})
    assert line["n_gpus"] == 2 and line["detected"] == 5
    if capi.device_count() < 2:
        assert line["config"]["comm"] == "host"
        assert "host-staged transport" in err


def test_analyzer_keeps_progress_in_front_of_a_bad_file(tmp_path):
    """analyzer.rs:414-417,447-451: the reference's sequential map has analysed AND persisted every video in front of
    the one that fails.  Here: two good files, one that is not RIFF/WAVE, one more good file -> the call fails with the
    bad file's error, the first two .needle.dat files exist and are complete, the fourth is untouched."""
    eps = synth.make_library(3, 60.0, 15.0)
    paths = [str(tmp_path / f"ep{k}.wav") for k in range(4)]
    for p, e in zip([paths[0], paths[1], paths[3]], eps):
        synth.write_wav(p, e.pcm)
    open(paths[2], "wb").write(b"definitely not a wave file" * 20)
    with pytest.raises(capi.NeedleError):
        capi.Analyzer.from_files(paths).run(0.3, persist=True)
    hd = O.duration_from_secs_f32(0.3)
    for k in (0, 1):
        fh = capi.FrameHashes.from_path(os.path.splitext(paths[k])[0] + ".needle.dat")
        want = O.analyze_batch([eps[k].pcm[: len(eps[k].pcm) // 2]], 1, hd)[0]
        assert fh.opening_data()[0].tolist() == [h for h, _ in want.opening]
    assert not os.path.exists(os.path.splitext(paths[3])[0] + ".needle.dat")


def test_skip_files_are_checked_video_by_video_in_order(tmp_path, capfd):
    """comparator.rs:593-626 interleaves check, best match, display and skip-file write per video.  The same path
    twice in the list: the second occurrence is skipped because the first one's skip file has just been written."""
    eps = synth.make_library(2, 90.0, 20.0)
    a, b = str(tmp_path / "a.wav"), str(tmp_path / "b.wav")
    synth.write_wav(a, eps[0].pcm)
    synth.write_wav(b, eps[1].pcm)
    paths = [a, b, a]
    capi.Analyzer.from_files([a, b]).run(0.3, persist=True)
    capfd.readouterr()
    cmp = capi.Comparator(paths, min_opening_duration=10)
    cmp.run(analyze=False, display=True, use_skip_files=True, write_skip_files=True)
    out = capfd.readouterr().out
    blocks = out.strip().split("\n\n")
    assert out.count("Skipping due to existing skip file...") == 1
    assert out.rstrip().endswith("Skipping due to existing skip file...")
    assert out.count("* Opening - ") == 2 and len(blocks) >= 3
    assert os.path.exists(os.path.splitext(a)[0] + ".needle.skip.json")


def test_bench_line_carries_the_contract(tmp_path):
    """`python bench.py` (N = 1) on a small library: ONE JSON line with the contract's keys, a roofline block for the
    dominant kernel (HIP events in the timed region), an honest scan roofline (issued evaluations against the ceiling
    measured in the run: a fraction <= 1), the certified dtype with its fallback share, device clocks, a CPU baseline
    that cross-checked the GPU's hashes and results."""
    line, _ = _bench(["--episodes", "6", "--minutes", "3", "--intro-seconds", "40", "--steps", "4", "--warmup", "2",
                      "--search-only-episodes", "8"], {})
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["n_gpus"] == 1 and line["steps"] == 4 and line["warmup"] == 2
    # BASELINE.md publishes no number for this metric on this hardware: the baseline is the reference's CPU path timed in the
    # same run, against the GPU path that also starts from host memory (never the resident figure `value` holds)
    assert line["vs_baseline"] == pytest.approx(line["end_to_end"]["pinned"]["pairs_per_s"] / line["cpu_baseline"]["value"], rel=1e-2)
    assert line["latency_ms"]["resident"] > 0 and line["latency_ms"]["from_pinned_host_pcm"] > 0
    assert "workload" in line["config"] and line["data"] == "synthetic" and line["detected"] == 6
    r = line["roofline"]
    assert 0 < r["frac"] < 1 and r["achieved"] > 0 and r["avg_launch_ms"] > 0 and r["kernel"] in ("stft_chroma32", "hamming_runs")
    if r["kernel"] == "stft_chroma32":                           # the dominant kernel at BASELINE's sizes
        assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    else:                                                        # a library this small is dominated by the scan's latency:
        assert r["bound"] == "int valu" and r["unit"] == "lane-instructions/s" and "hbm" in r   # never labelled HBM-bound
    rs = line["roofline_search"]
    assert rs["unit"] == "lane-instructions/s" and 0 < rs["frac"] <= 1.0 and rs["pruning_factor"] > 1
    assert line["dtype"].startswith("f64-certified") and 0 <= line["fallback_frac"]["items"] < 0.05
    c = line["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["gpu_matches_oracle"] is True
    assert line["device_state"]["source"] and "end_to_end" in line and "search_only" in line
    assert 0 < line["search_only"]["roofline"]["frac"] <= 1.0
