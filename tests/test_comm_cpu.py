"""The product's multi-rank path, the parts that need no GPU: the sharding plan exported by the C ABI, the
host-staged communicator between real processes, and -- world_size 2 and 3 over torch.distributed/gloo beside the
library's own transport -- run-list exchange + the C++ epilogue sharded by video (tests/comm_worker.py cpu)."""
import json
import multiprocessing as mp
import os
import socket
import subprocess
import sys
import uuid

import numpy as np
import pytest

from needle_amd import capi

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch(mode, world, out, args, extra_env=None, local_ranks=None, timeout=600):
    env = dict(os.environ, WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               NEEDLE_TEST_RDZV_KEY=uuid.uuid4().hex, OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.update(extra_env or {})
    procs = []
    for rank in range(world):
        renv = dict(env, RANK=str(rank), LOCAL_RANK=str(rank if local_ranks is None else local_ranks[rank]))
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "comm_worker.py"), mode, out] + [str(a) for a in args],
                                      env=renv))
    try:
        codes = [p.wait(timeout=timeout) for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    assert codes == [0] * world, codes
    return [json.load(open(f"{out}.{r}")) for r in range(world)]


def test_shard_plan_of_the_c_abi_covers_everything_once():
    for units in [0, 1, 2, 7, 28, 378, 1999000]:
        for world in [1, 2, 3, 4, 8]:
            ranges = [capi.comm_shard(units, world, r) for r in range(world)]
            assert sum(c for _, c in ranges) == units
            at = 0
            for first, count in ranges:
                assert first == min(at, units) and count <= -(-units // world)
                at += count
    assert capi.comm_shard(28, 8, 7) == (28, 0) and capi.comm_shard(378, 8, 7) == (336, 42)
    assert capi.comm_rank() == 0 and capi.comm_world_size() == 1 and capi.comm_backend() == "none"


def _host_comm_rank(rank, world, key, slot_bytes, q):
    os.environ["NEEDLE_HIP_COMM"] = "host"
    os.environ["NEEDLE_HIP_COMM_SLOT_BYTES"] = str(slot_bytes)
    from needle_amd import rendezvous
    rdzv = rendezvous.init_comm(capi, rank, world, None, key=key)
    ok = capi.comm_backend() == "host" and capi.comm_rank() == rank and capi.comm_world_size() == world
    for size in (1, 5, 64, 1000, 4099):                # below, at and above the slot size; odd byte counts
        mine = (np.arange(size, dtype=np.uint8) * (rank + 3) + rank).astype(np.uint8)
        got = capi.comm_all_gather(mine)
        for r in range(world):
            ok = ok and np.array_equal(got[r], (np.arange(size, dtype=np.uint8) * (r + 3) + r).astype(np.uint8))
        capi.comm_barrier()
    threads = capi.host_threads()                      # this rank's share of the node's CPUs while the communicator is up
    capi.comm_finalize()
    ok = ok and capi.host_threads() >= threads         # back to every usable CPU without one
    rdzv.close()
    q.put((rank, bool(ok), threads) if os.environ.get("NEEDLE_TEST_REPORT_THREADS") else (rank, bool(ok)))


@pytest.mark.parametrize("world", [2, 3])
def test_host_staged_communicator_between_processes(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    key = uuid.uuid4().hex
    procs = [ctx.Process(target=_host_comm_rank, args=(r, world, key, 256, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert got == [(r, True) for r in range(world)]
    assert not [f for f in os.listdir("/dev/shm") if f.startswith("needle_comm_")]      # unlinked once everybody attached


def test_eight_ranks_share_the_hosts_cpus(monkeypatch):
    """Every rank sizes its host pools (epilogue, readers, upload staging) to usable CPUs / ranks of the node: eight
    ranks of one node never run more host threads than the node gives this job.  LOCAL_WORLD_SIZE,
    which torchrun exports, wins over the world size; NEEDLE_HOST_THREADS wins over both."""
    usable = capi.host_threads()
    assert usable == min(len(os.sched_getaffinity(0)), usable) >= 1 and capi.comm_world_size() == 1
    monkeypatch.setenv("NEEDLE_TEST_REPORT_THREADS", "1")
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    key = uuid.uuid4().hex
    procs = [ctx.Process(target=_host_comm_rank, args=(r, world, key, 4096, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert [g[:2] for g in got] == [(r, True) for r in range(world)]
    threads = [g[2] for g in got]
    assert threads == [max(1, usable // world)] * world and sum(threads) <= max(usable, world)
    monkeypatch.setenv("NEEDLE_HOST_THREADS", "3")
    assert capi.host_threads() == 3


@pytest.mark.parametrize("world,n", [(2, 7), (3, 8)])
def test_gloo_world_run_exchange_and_sharded_epilogue(tmp_path, world, n):
    got = launch("cpu", world, str(tmp_path / "cpu"), [n])
    assert got[0]["runs"] >= n * (n - 1) // 2                 # every pair shares the planted run
    assert sum(got[0]["counts"]) == got[0]["runs"] and all(c > 0 for c in got[0]["counts"])
    for g in got:
        assert g["sharded"] == g["full"] == g["oracle"]
        assert all(r is not None and r[0] is not None for r in g["sharded"])
    assert all(g["sharded"] == got[0]["sharded"] for g in got)


def test_rank_videos_cuts_the_hashes_not_the_videos():
    """needle_hip_library_rank_videos: 28 episodes x 24 min on 8 ranks are 3.5 episodes' worth of hashes each -- every
    rank holds 4 or 5 (partly needed) episodes and none idles; the shares tile the library in order."""
    total = 24 * 60 * 11025
    lib = capi.Library(28)
    assert lib.rank_videos([total] * 28, 1, 0) == (0, 28)
    shares = [lib.rank_videos([total] * 28, 8, r) for r in range(8)]
    assert all(4 <= c <= 5 for _, c in shares), shares
    assert shares[0][0] == 0 and shares[-1][0] + shares[-1][1] == 28
    for (f0, c0), (f1, _) in zip(shares, shares[1:]):
        assert f0 + c0 - 1 <= f1 <= f0 + c0                       # neighbours share at most the episode on their border
    # ragged lengths and endings: still a tiling, still nobody idle
    lens = [int((10 + 3 * (k % 5)) * 60 * 11025) for k in range(9)]
    lib2 = capi.Library(9).include_endings()
    shares = [lib2.rank_videos(lens, 4, r) for r in range(4)]
    assert shares[0][0] == 0 and shares[-1][0] + shares[-1][1] == 9 and all(c >= 1 for _, c in shares)
    with pytest.raises(capi.NeedleError):
        lib.rank_videos([total] * 28, 8, 8)


def test_config4_plan_for_eight_ranks():
    """BASELINE.json configs[4] on 8 ranks, the parts that need no device: the hash-block plan at 5 441 hashes per episode
    (needle_hip_library_rank_videos: 2000 episodes cut into 8 equal blocks of hashes -- every rank holds 250 or 251
    episodes, neighbours share at most the episode on their border, the shares tile the library) and the pair plan
    (1 999 000 pairs in 8 contiguous ranges).  The GPU suite runs this shape with 4 and 5 ranks (a one-GPU box allows no
    more processes); the plan for 8 is the same function."""
    n, samples = 2000, int(round(45 * 60.0 / 2 * 11025))
    lib = capi.Library(n, opening_search_percentage=1.0)
    shares = [lib.rank_videos([samples] * n, 8, r) for r in range(8)]
    assert shares[0][0] == 0 and shares[-1][0] + shares[-1][1] == n
    assert all(250 <= c <= 251 for _, c in shares), shares
    for (f0, c0), (f1, _) in zip(shares, shares[1:]):
        assert f0 + c0 - 1 <= f1 <= f0 + c0
    pairs = n * (n - 1) // 2
    ranges = [capi.comm_shard(pairs, 8, r) for r in range(8)]
    assert sum(c for _, c in ranges) == pairs and max(c for _, c in ranges) - min(c for _, c in ranges) <= 8
    assert all(ranges[r][0] + ranges[r][1] == ranges[r + 1][0] for r in range(7))
