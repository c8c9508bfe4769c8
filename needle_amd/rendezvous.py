"""Out-of-band exchange of the communicator id between the rank processes of ONE node.

needle_hip_comm_init needs the 128 bytes rank 0 got from needle_hip_comm_create_id (the same bootstrap contract as
ncclGetUniqueId / ncclCommInitRank); how they travel is the host program's business.  This is the small file-based
exchange bench.py, the tests and the tools use.  No torch, no sockets.

Layout.  A BASE directory under /tmp, private to the user (mode 0700, owner checked, never a symlink), keyed by what
all ranks of one launch share: the launcher's PID and start time (torchrun's agent, or bench.py's own parent process),
MASTER_PORT, torchrun's run id and restart count, and NEEDLE_RDZV_NONCE when the launcher exports one.  Inside it one
SESSION directory per group of processes that talk to each other.  Rank 0 of a group creates a fresh session (random
name), removes whatever older sessions it finds, and publishes the name in a pointer file together with its own PID
and process start time; the other ranks accept a pointer only while that process is alive with that start time.  So a
directory left behind by a launch that was killed -- or by an earlier attempt of the same launcher: torchrun keeps its
agent PID and port across worker-group restarts -- can never be mistaken for the current one: its pointer names a dead
process.
"""
from __future__ import annotations

import json
import os
import secrets
import shutil
import stat
import time


def _proc_start(pid: int) -> str:
    try:
        with open(f"/proc/{pid}/stat") as f:
            return f.read().rsplit(")", 1)[1].split()[19]      # field 22: start time in clock ticks
    except (OSError, IndexError):
        return ""


def _launcher_key() -> str:
    ppid = os.getppid()
    parts = [str(ppid), _proc_start(ppid) or "0", os.environ.get("MASTER_PORT", "0")]
    for name in ("TORCHELASTIC_RUN_ID", "TORCHELASTIC_RESTART_COUNT", "NEEDLE_RDZV_NONCE"):
        v = os.environ.get(name)
        if v:
            parts.append("".join(c if c.isalnum() else "-" for c in v)[:40])
    return "_".join(parts)


def _private_dir(path: str) -> None:
    """Creates `path` with mode 0700, or verifies that an existing one is a real directory owned by this user that
    nobody else can write to (a predictable name in a world-writable /tmp must not be trusted blindly)."""
    try:
        os.mkdir(path, 0o700)
    except FileExistsError:
        pass
    st = os.lstat(path)
    if not stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o022):
        raise RuntimeError(f"rendezvous: {path} is not a private directory of uid {os.getuid()}; refusing to use it")


class FileRendezvous:
    def __init__(self, rank: int, world: int, key: str | None = None, timeout_s: float = 300.0, role: str = "ranks"):
        self.rank, self.world, self.timeout_s = rank, world, timeout_s
        self.base = os.path.join(os.environ.get("NEEDLE_RDZV_DIR", "/tmp"),
                                 f"needle_rdzv_{os.getuid()}_{key or _launcher_key()}")
        _private_dir(self.base)
        pointer = os.path.join(self.base, f"session.{role}")
        if rank == 0:
            for name in os.listdir(self.base):                  # leftovers of dead sessions of this role
                if name.startswith(f"{role}-"):
                    shutil.rmtree(os.path.join(self.base, name), ignore_errors=True)
            session = f"{role}-{secrets.token_hex(8)}"
            self.dir = os.path.join(self.base, session)
            os.mkdir(self.dir, 0o700)
            self._write(pointer, json.dumps({"session": session, "pid": os.getpid(),
                                             "start": _proc_start(os.getpid())}).encode())
        else:
            deadline = time.monotonic() + timeout_s
            while True:
                try:
                    with open(pointer, "rb") as f:
                        p = json.loads(f.read())
                    if p["start"] and _proc_start(int(p["pid"])) == p["start"]:
                        self.dir = os.path.join(self.base, p["session"])
                        if os.path.isdir(self.dir):
                            break
                except (OSError, ValueError, KeyError):
                    pass
                if time.monotonic() > deadline:
                    raise TimeoutError(f"rendezvous: no live session under {self.base} (rank {rank} of {world})")
                time.sleep(0.005)

    @staticmethod
    def _write(path: str, value: bytes) -> None:
        tmp = os.path.join(os.path.dirname(path), f".{os.path.basename(path)}.{os.getpid()}.tmp")
        with open(tmp, "wb") as f:
            f.write(value)
        os.replace(tmp, path)                                    # atomic: a reader sees nothing or everything

    def set(self, name: str, value: bytes) -> None:
        self._write(os.path.join(self.dir, name), value)

    def try_get(self, name: str):
        try:
            with open(os.path.join(self.dir, name), "rb") as f:
                return f.read()
        except OSError:
            return None

    def get(self, name: str, timeout_s: float | None = None) -> bytes:
        deadline = time.monotonic() + (self.timeout_s if timeout_s is None else timeout_s)
        while True:
            v = self.try_get(name)
            if v is not None:
                return v
            if time.monotonic() > deadline:
                raise TimeoutError(f"rendezvous: {os.path.join(self.dir, name)} never appeared "
                                   f"(rank {self.rank} of {self.world})")
            time.sleep(0.002)

    def broadcast(self, name: str, value: bytes | None) -> bytes:
        """Rank 0's `value` on every rank."""
        if self.rank == 0:
            self.set(name, value)
            return value
        return self.get(name)

    def barrier(self, name: str) -> None:
        self.set(f"{name}.{self.rank}", b"1")
        for r in range(self.world):
            self.get(f"{name}.{r}")

    def all_gather(self, name: str, value: bytes) -> list:
        self.set(f"{name}.{self.rank}", value)
        return [self.get(f"{name}.{r}") for r in range(self.world)]

    def close(self) -> None:
        """Collective: every rank has passed; rank 0 removes the session once nobody reads it any more."""
        self.barrier("exit")
        self.set(f"done.{self.rank}", b"1")
        if self.rank != 0:
            return
        for r in range(self.world):
            self.get(f"done.{r}")
        self.remove()

    def remove(self) -> None:
        """Rank 0 (or a launcher cleaning up after ranks it killed): delete the session and, if empty, the base."""
        shutil.rmtree(self.dir, ignore_errors=True)
        try:
            for name in os.listdir(self.base):
                if name.startswith("session.") or name.startswith("."):
                    os.unlink(os.path.join(self.base, name))
            os.rmdir(self.base)
        except OSError:
            pass                                                 # another role's session is still in there


def init_comm(capi, rank: int, world: int, local_rank: int | None, key: str | None = None) -> FileRendezvous:
    """set_device + communicator for this rank process.  Tries the backend NEEDLE_HIP_COMM names (default RCCL); if
    RCCL cannot be brought up on ANY rank, all ranks agree (through the rendezvous) to use the host-staged transport.
    (A bring-up that HANGS is the launcher's business: bench.py's per-rank supervisor kills the worker and starts it
    again over the host transport.)  local_rank None: no device is bound (host-only use of the host transport)."""
    rdzv = FileRendezvous(rank, world, key)
    if local_rank is not None:
        capi.set_device(local_rank)
    if world == 1:
        capi.comm_init(capi.comm_create_id(), 0, 1)
        return rdzv
    attempt = 0
    while True:
        ok, err = True, ""
        try:
            cid = rdzv.broadcast(f"id{attempt}", capi.comm_create_id() if rank == 0 else None)
            capi.comm_init(cid, rank, world)
        except Exception as e:                                   # noqa: BLE001 -- any failure means "not this backend"
            ok, err = False, str(e)
            if rank == 0 and rdzv.try_get(f"id{attempt}") is None:
                rdzv.set(f"id{attempt}", b"\0" * 128)           # unblock the others: their init fails too
        votes = rdzv.all_gather(f"ok{attempt}", b"1" if ok else b"0")
        if all(v == b"1" for v in votes):
            return rdzv
        if ok:
            capi.comm_finalize()
        if os.environ.get("NEEDLE_HIP_COMM") == "host" or attempt > 0:
            raise RuntimeError(f"communicator initialisation failed on rank {rank}: {err or 'another rank failed'}")
        print(f"[needle] rank {rank}: RCCL bring-up failed ({err or 'on another rank'}); all ranks switch to the "
              "host-staged transport", file=__import__("sys").stderr, flush=True)
        os.environ["NEEDLE_HIP_COMM"] = "host"                   # agreed fallback: host-staged shared memory
        attempt += 1
