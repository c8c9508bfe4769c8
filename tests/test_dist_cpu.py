"""world_size-2 (and 3, uneven shards) gloo runs of the multi-GPU exchange plan on CPU: after the row
all-gather every rank holds every hash row, and the gathered run list equals the single-process one."""
import json
import os
import socket
import subprocess
import sys

import pytest

from tests import dist_plan as ndist

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _launch(world, n, out, slab=0):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        if slab:
            env["NEEDLE_TEST_SLAB"] = str(slab)
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "dist_worker.py"), out, str(n)], env=env))
    for p in procs:
        assert p.wait(timeout=300) == 0
    return [json.load(open(f"{out}.{r}")) for r in range(world)]


def test_shard_plan_covers_everything_once():
    for n in [1, 2, 7, 28, 280]:
        for world in [1, 2, 3, 4, 8]:
            ranges = [ndist.shard(n, world, r) for r in range(world)]
            assert sum(c for _, c in ranges) == n
            assert all(ranges[r][0] + ranges[r][1] == ranges[r + 1][0] or ranges[r + 1][1] == 0
                       for r in range(world - 1))
            assert ndist.block(n, world) * world >= n
    assert ndist.pair_count(28) == 378 and ndist.pair_count(1) == 0


@pytest.mark.parametrize("world,n", [(2, 7), (3, 8)])
def test_gloo_job_equals_single_process(tmp_path, world, n):
    single = _launch(1, n, str(tmp_path / "single"))[0]
    multi = _launch(world, n, str(tmp_path / f"w{world}"))
    assert all(m["arena_complete"] for m in multi)
    assert multi[0]["runs"] is not None and all(m["runs"] is None for m in multi[1:])
    assert len(single["runs"]) >= n * (n - 1) // 2           # every pair shares the planted intro
    assert sorted(map(tuple, multi[0]["runs"])) == sorted(map(tuple, single["runs"]))
    # the single-collective slab gather agrees, both when every rank fits its slab and when one overflows
    for slab in (512, 4):
        got = _launch(world, n, str(tmp_path / f"s{world}_{slab}"), slab=slab)
        assert sorted(map(tuple, got[0]["runs"])) == sorted(map(tuple, single["runs"]))
        # bench.py's pipelined form (reused buffers, epilogue deferred into the next job): every job, every rank
        for m in got:
            assert len(m["pipelined"]) == 6           # 3 deferred-epilogue jobs + 3 JobPipeline steps
            for job in m["pipelined"]:
                assert sorted(map(tuple, job)) == sorted(map(tuple, single["runs"]))
