"""bench.py's N-rank launcher without a GPU: every rank is a supervisor + a worker; a worker that hangs in (or dies
before) its first collective is killed from outside and every rank restarts over the host-staged transport; exactly
one JSON line comes out.  The workers here are bench.py's own `--selftest-worker` stubs (no GPU, no library)."""
import json
import os
import subprocess
import sys
import time

import pytest

from needle_amd import rendezvous

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(mode, world, timeout_s, tmp_path, extra_env=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "NEEDLE_HIP_COMM")}
    env["NEEDLE_RDZV_DIR"] = str(tmp_path)
    env.update(extra_env or {})
    t0 = time.monotonic()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--selftest-worker", mode,
                          "--launch-timeout", str(timeout_s)], env=env, capture_output=True, text=True, timeout=120)
    return out, time.monotonic() - t0


def test_all_ranks_finish_first_attempt(tmp_path):
    out, _ = _run("ok", 4, 30, tmp_path)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    assert json.loads(lines[0]) == {"selftest": "ok", "n_gpus": 4, "comm": "rccl", "attempt": 0}
    assert os.listdir(tmp_path) == []                            # the supervisors' directory is gone


@pytest.mark.parametrize("mode", ["hang", "fail"])
def test_stuck_or_dead_rank_restarts_everyone_over_the_host_transport(tmp_path, mode):
    out, wall = _run(mode, 3, 3, tmp_path)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1                                       # one line, from the attempt that succeeded
    assert json.loads(lines[0]) == {"selftest": mode, "n_gpus": 3, "comm": "host", "attempt": 1}
    assert "restarts over the host-staged transport" in out.stderr
    assert wall < 60


def test_host_transport_that_hangs_too_gives_up(tmp_path):
    out, wall = _run("hang", 2, 2, tmp_path, {"NEEDLE_HIP_COMM": "host"})   # selftest "hang" only hangs off-host ...
    assert out.returncode == 0                                   # ... so this passes at once
    out, wall = _run("ok", 2, 2, tmp_path, {"NEEDLE_HIP_COMM": "host"})
    assert out.returncode == 0 and json.loads(out.stdout.strip())["comm"] == "host"


def test_stale_session_of_a_dead_launch_is_not_joined(tmp_path, monkeypatch):
    """A pointer left behind by a rank 0 that no longer exists must not be accepted (torchrun keeps its agent PID and
    port across worker-group restarts, so the base directory is the same)."""
    monkeypatch.setenv("NEEDLE_RDZV_DIR", str(tmp_path))
    r0 = rendezvous.FileRendezvous(0, 2, key="k")
    r0.set("id0", b"old")
    ptr = os.path.join(r0.base, "session.ranks")
    p = json.loads(open(ptr).read())
    p["pid"], p["start"] = 1, "1"                                # "rank 0" is some other process now
    open(ptr, "w").write(json.dumps(p))
    with pytest.raises(TimeoutError):
        rendezvous.FileRendezvous(1, 2, key="k", timeout_s=0.3)
    fresh = rendezvous.FileRendezvous(0, 2, key="k")             # the restarted rank 0 sweeps the old session away
    assert not os.path.exists(r0.dir) and fresh.try_get("id0") is None
    r1 = rendezvous.FileRendezvous(1, 2, key="k", timeout_s=5)
    assert r1.dir == fresh.dir
    fresh.remove()
    assert not os.path.exists(fresh.base)


def test_base_directory_must_be_private(tmp_path, monkeypatch):
    monkeypatch.setenv("NEEDLE_RDZV_DIR", str(tmp_path))
    base = tmp_path / f"needle_rdzv_{os.getuid()}_k2"
    base.mkdir(mode=0o777)
    os.chmod(base, 0o777)
    with pytest.raises(RuntimeError):
        rendezvous.FileRendezvous(0, 1, key="k2")
