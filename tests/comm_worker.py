"""One rank of a multi-process run of the product's N-rank path (include/needle_hip.h "multi-GPU"), for
tests/test_comm_cpu.py and the -m gpu tests in tests/test_gpu_multi.py.

  gpu <out> <n> <seconds> : needle_hip_comm_init + Library.set_pcm (own block) + 3 pipelined job_begin/_end;
                            writes every job's results, the run count and every video's hashes.
  lib <out> <n> <minutes> : BASELINE.json configs[4]'s shape between ranks: this rank's share of n episodes of `minutes`
                            (opening half) generated in HBM (synth.DeviceLibrary), three pipelined jobs; writes the
                            results, the run count, a digest of the SORTED complete run list of every job, what the
                            job's collectives moved, the host threads this rank used and its epilogue wait.
  cpu <out> <n>           : no device.  The exchange + sharded-epilogue logic on host data: run lists of this
                            rank's pair range (from the oracle's table-free scan -- the checker standing in for the
                            scan kernel, tests only) are all-gathered twice, through torch.distributed/gloo and
                            through the library's host-staged communicator; each rank runs the C++ epilogue for its
                            own block of videos; blocks are gathered and compared with the unsharded epilogue.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from needle_amd import capi, rendezvous, synth  # noqa: E402


def _res(rs):
    return [None if r is None else [r.opening, r.ending] for r in rs]


def gpu_main(out, n, seconds):
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    rdzv = rendezvous.init_comm(capi, rank, world, int(os.environ.get("LOCAL_RANK", "0")),
                                key=os.environ["NEEDLE_TEST_RDZV_KEY"])
    # NEEDLE_TEST_RAGGED: episode k lasts seconds + 3.7 k; NEEDLE_TEST_HASH_DURATION: another step between kept hashes
    secs = [seconds + (3.7 * k if os.environ.get("NEEDLE_TEST_RAGGED") else 0.0) for k in range(n)]
    totals = [int(round(v * synth.RATE)) for v in secs]
    lib = capi.Library(n, hash_duration=float(os.environ.get("NEEDLE_TEST_HASH_DURATION", "0.3")))
    if os.environ.get("NEEDLE_TEST_ENDINGS"):
        lib.include_endings()
    first, count = lib.rank_videos(totals, world, rank)           # the episodes this rank's block of hashes depends on
    mine = {k: synth.make_episode(k, secs[k], 20.0) for k in range(first, first + count)}
    lib.set_pcm([mine[k].pcm if k in mine else None for k in range(n)], totals)
    cmp = capi.Comparator([f"ep{k}.wav" for k in range(n)], min_opening_duration=10,
                          include_endings=bool(os.environ.get("NEEDLE_TEST_ENDINGS")))
    jobs = []
    capi.set_kernel_timing("stft_chroma32,stft_chroma")
    lib.job_begin(cmp, 0)
    lib.job_begin(cmp, 1)
    comm = []

    def end(slot):
        jobs.append(lib.job_end(cmp, slot))
        comm.append(dict(lib.job_comm_bytes(slot), held=int(len(lib.job_runs(slot)))))

    end(0)
    lib.job_begin(cmp, 0)
    end(1)
    end(0)
    hashes = [lib.frame_hashes(v).opening_data()[0].tolist() for v in range(n)]
    stft_ms = max(capi.last_kernel_ms("stft_chroma32"), capi.last_kernel_ms("stft_chroma"))
    capi.set_kernel_timing(None)
    with open(f"{out}.{rank}", "w") as f:
        json.dump({"rank": rank, "backend": capi.comm_backend(), "world": capi.comm_world_size(),
                   "videos_held": [first, count], "stft_ms": stft_ms,
                   "jobs": [{"results": _res(r), "runs": k, "comm": c} for (r, k), c in zip(jobs, comm)], "hashes": hashes}, f)
    capi.comm_barrier()
    capi.comm_finalize()
    rdzv.close()


def run_digest(runs):
    """Order-free digest of a complete run list: within a rank's slab the order is whatever the atomics produced."""
    import hashlib
    keys = np.stack([runs[f].astype(np.uint32) for f in ("problem", "src_end", "dst_end", "len", "src_match_hash",
                                                         "dst_match_hash")], axis=1)
    keys = keys[np.lexsort((keys[:, 2], keys[:, 1], keys[:, 0]))]
    return hashlib.sha256(np.ascontiguousarray(keys).tobytes()).hexdigest()


def lib_main(out, n, minutes):
    import time
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    rdzv = rendezvous.init_comm(capi, rank, world, int(os.environ.get("LOCAL_RANK", "0")),
                                key=os.environ["NEEDLE_TEST_RDZV_KEY"])
    samples = int(round(minutes * 60.0 / 2 * synth.RATE))
    lib = capi.Library(n, opening_search_percentage=1.0)
    first, count = lib.rank_videos([samples] * n, world, rank)
    gen = synth.DeviceLibrary(count, samples, 90.0, first_episode=first)
    ptrs = gen.pointers()
    lib.set_pcm_device([ptrs[k - first] if first <= k < first + count else None for k in range(n)], [samples] * n)
    gen.free()
    cmp = capi.Comparator([f"episode-{k:04d}.wav" for k in range(n)])
    jobs = []

    def end(slot):
        t0 = time.perf_counter()
        res, found = lib.job_end(cmp, slot)
        wait_ms = 1e3 * (time.perf_counter() - t0)
        held = lib.job_runs(slot)      # the complete list, or -- owner-directed exchange -- the runs of this rank's own videos' pairs
        jobs.append({"results": _res(res), "runs": found, "held": int(len(held)), "digest": run_digest(held),
                     "comm": lib.job_comm_bytes(slot), "end_ms": wait_ms})

    lib.job_begin(cmp, 0)
    lib.job_begin(cmp, 1)
    end(0)
    lib.job_begin(cmp, 0)
    end(1)
    end(0)
    d_arena, stride = lib.hash_arena()
    arena = np.zeros(n * stride, dtype=np.uint32)
    capi.check(capi.lib().needle_hip_memcpy_d2h(arena.ctypes.data, d_arena, arena.nbytes))
    import hashlib
    kept = int(capi.lib().needle_hip_fingerprint_num_kept(samples, 2))      # the padding beyond it depends on the world size
    with open(f"{out}.{rank}", "w") as f:
        json.dump({"rank": rank, "backend": capi.comm_backend(), "world": capi.comm_world_size(),
                   "videos_held": [first, count], "host_threads": capi.host_threads(), "jobs": jobs, "audit": lib.audit(),
                   "arena_digest": hashlib.sha256(np.ascontiguousarray(arena.reshape(n, stride)[:, :kept]).tobytes()).hexdigest(),
                   "stride": stride}, f)
    capi.comm_barrier()
    capi.comm_finalize()
    rdzv.close()


def cpu_main(out, n):
    import torch
    import torch.distributed as dist
    from oracle import oracle as O
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    os.environ["NEEDLE_HIP_COMM"] = "host"
    rdzv = rendezvous.init_comm(capi, rank, world, None, key=os.environ["NEEDLE_TEST_RDZV_KEY"])
    assert capi.comm_backend() == "host" and capi.comm_world_size() == world and capi.comm_rank() == rank

    # the same synthetic hash library on every rank (metadata every rank has: lengths and timestamps)
    rng = np.random.default_rng(99)
    hd = O.duration_from_secs_f32(0.3)
    intro = rng.integers(0, 2 ** 32, 130, dtype=np.uint64).astype(np.uint32)
    seqs, fhs = [], []
    for v in range(n):
        length = 300 + 17 * v
        h = rng.integers(0, 2 ** 32, length, dtype=np.uint64).astype(np.uint32)
        a = 20 + (31 * v) % 120
        flips = (np.uint32(1) << rng.integers(0, 32, 130).astype(np.uint32)) * (rng.random(130) < 0.4)
        h[a:a + 130] = intro ^ flips
        ts = [t for _, t in O.step_and_timestamp(np.zeros(2 * length, dtype=np.uint32), hd)][:length]
        seqs.append(h)
        fhs.append(capi.FrameHashes.new(list(zip(h.tolist(), ts)), (), hd, ""))
    cmp = capi.Comparator([f"v{v}.wav" for v in range(n)], min_opening_duration=20)
    min_len = 82

    # this rank's pair range, scanned by the checker
    pairs = [(i, j) for i in range(n) for j in range(i + 1, n)]
    pfirst, pcount = capi.comm_shard(len(pairs), world, rank)
    _, all_runs = O.diagonal_runs_all_pairs(seqs, 10, min_len, capacity=1 << 16)
    local = np.zeros(0, dtype=capi.RUN_DTYPE)
    rows = [r for r in all_runs.tolist() if pfirst <= r[0] < pfirst + pcount]
    local = np.zeros(len(rows), dtype=capi.RUN_DTYPE)
    for q, (p, se, de, ln) in enumerate(rows):
        i, j = pairs[p]
        local[q] = (p, se, de, ln, O.simhash32(seqs[i][se - ln: se + 1].tolist()),
                    O.simhash32(seqs[j][de - ln: de + 1].tolist()))

    # exchange 1: gloo (counts, then padded rows)
    counts = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([len(local)], dtype=torch.int64))
    counts = [int(c.item()) for c in counts]
    width = max(max(counts), 1)
    padded = np.zeros((width, capi.RUN_WORDS), dtype=np.int32)
    padded[: len(local)] = local.view(np.uint32).reshape(-1, capi.RUN_WORDS).view(np.int32)
    gathered = [torch.zeros((width, capi.RUN_WORDS), dtype=torch.int32) for _ in range(world)]
    dist.all_gather(gathered, torch.from_numpy(padded))
    via_gloo = np.concatenate([g.numpy()[: counts[r]] for r, g in enumerate(gathered)], axis=0)
    # exchange 2: the library's own host-staged communicator (slab layout of the product: count row + runs)
    slab = np.zeros((1 + width, capi.RUN_WORDS), dtype=np.int32)
    slab[0, 0] = len(local)
    slab[1: 1 + len(local)] = padded[: len(local)]
    allslabs = capi.comm_all_gather(slab)
    via_comm = np.concatenate([allslabs[r, 1: 1 + int(allslabs[r, 0, 0])] for r in range(world)], axis=0)
    assert [int(allslabs[r, 0, 0]) for r in range(world)] == counts
    assert np.array_equal(via_gloo, via_comm)
    merged = np.ascontiguousarray(via_comm).view(np.uint32).view(capi.RUN_DTYPE).reshape(-1)

    # sharded epilogue: own block of videos, then gather the blocks
    vfirst, vcount = capi.comm_shard(n, world, rank)
    mine = cmp.results_from_runs(fhs, merged, vfirst, vcount)
    assert all(r is None for k, r in enumerate(mine) if not (vfirst <= k < vfirst + vcount))
    blocks = [None] * world
    dist.all_gather_object(blocks, _res(mine[vfirst: vfirst + vcount]))
    sharded = [r for b in blocks for r in b]
    full = _res(cmp.results_from_runs(fhs, merged))
    ofh = [O.FrameHashes(list(zip(s.tolist(), [t for _, t in O.step_and_timestamp(np.zeros(2 * len(s), dtype=np.uint32), hd)][: len(s)])),
                         [], hd, "") for s in seqs]
    want = O.run_with_frame_hashes(O.Comparator(), ofh)
    oracle = [None if r is None else [r.opening, r.ending] for r in want]
    capi.comm_barrier()
    with open(f"{out}.{rank}", "w") as f:
        json.dump({"rank": rank, "sharded": sharded, "full": full, "oracle": _res_json(oracle), "runs": len(merged),
                   "counts": counts}, f)
    capi.comm_finalize()
    rdzv.close()
    dist.barrier()
    dist.destroy_process_group()


def _res_json(rs):
    return [None if r is None else [None if x is None else list(x) for x in r] for r in rs]


if __name__ == "__main__":
    if sys.argv[1] == "gpu":
        gpu_main(sys.argv[2], int(sys.argv[3]), float(sys.argv[4]))
    elif sys.argv[1] == "lib":
        lib_main(sys.argv[2], int(sys.argv[3]), float(sys.argv[4]))
    else:
        cpu_main(sys.argv[2], int(sys.argv[3]))
