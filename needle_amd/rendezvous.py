"""Out-of-band exchange of the communicator id between the rank processes of ONE node.

needle_hip_comm_init needs the 128 bytes rank 0 got from needle_hip_comm_create_id (the same bootstrap contract as
ncclGetUniqueId / ncclCommInitRank); how they travel is the host program's business.  This is the small file-based
exchange bench.py, the tests and the tools use: a directory under /tmp keyed by what all ranks of one launch share --
the launcher's PID and start time (torchrun's agent, or bench.py's own parent process) and MASTER_PORT -- so two
launches never see each other's files and a stale directory from a dead launch cannot match.  No torch, no sockets.
"""
from __future__ import annotations

import os
import time


def _launcher_key() -> str:
    ppid = os.getppid()
    start = "0"
    try:
        with open(f"/proc/{ppid}/stat") as f:
            start = f.read().rsplit(")", 1)[1].split()[19]      # field 22: start time in clock ticks
    except (OSError, IndexError):
        pass
    return f"{ppid}_{start}_{os.environ.get('MASTER_PORT', '0')}"


class FileRendezvous:
    def __init__(self, rank: int, world: int, key: str | None = None, timeout_s: float = 300.0):
        self.rank, self.world, self.timeout_s = rank, world, timeout_s
        self.dir = os.path.join(os.environ.get("NEEDLE_RDZV_DIR", "/tmp"), f"needle_rdzv_{key or _launcher_key()}")
        os.makedirs(self.dir, exist_ok=True)

    def set(self, name: str, value: bytes) -> None:
        tmp = os.path.join(self.dir, f".{name}.{os.getpid()}.tmp")
        with open(tmp, "wb") as f:
            f.write(value)
        os.replace(tmp, os.path.join(self.dir, name))           # atomic: a reader sees nothing or everything

    def get(self, name: str) -> bytes:
        path = os.path.join(self.dir, name)
        deadline = time.monotonic() + self.timeout_s
        while True:
            try:
                with open(path, "rb") as f:
                    return f.read()
            except FileNotFoundError:
                if time.monotonic() > deadline:
                    raise TimeoutError(f"rendezvous: {path} never appeared (rank {self.rank} of {self.world})")
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
        """Collective: every rank has passed; rank 0 removes the directory once nobody reads it any more."""
        self.barrier("exit")
        self.set(f"done.{self.rank}", b"1")
        if self.rank != 0:
            return
        for r in range(self.world):
            self.get(f"done.{r}")
        try:
            for f in os.listdir(self.dir):
                os.unlink(os.path.join(self.dir, f))
            os.rmdir(self.dir)
        except OSError:
            pass


def init_comm(capi, rank: int, world: int, local_rank: int | None, key: str | None = None) -> FileRendezvous:
    """set_device + communicator for this rank process.  Tries the backend NEEDLE_HIP_COMM names (default RCCL); if
    RCCL cannot be brought up on ANY rank, all ranks agree (through the rendezvous) to use the host-staged transport.
    local_rank None: no device is bound (host-only use of the host-staged transport)."""
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
            if rank == 0 and not os.path.exists(os.path.join(rdzv.dir, f"id{attempt}")):
                rdzv.set(f"id{attempt}", b"\0" * 128)           # unblock the others: their init fails too
        votes = rdzv.all_gather(f"ok{attempt}", b"1" if ok else b"0")
        if all(v == b"1" for v in votes):
            return rdzv
        if ok:
            capi.comm_finalize()
        if os.environ.get("NEEDLE_HIP_COMM") == "host" or attempt > 0:
            raise RuntimeError(f"communicator initialisation failed on rank {rank}: {err or 'another rank failed'}")
        os.environ["NEEDLE_HIP_COMM"] = "host"                   # agreed fallback: host-staged shared memory
        attempt += 1
