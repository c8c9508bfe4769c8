"""Worker for tests/test_dist_cpu.py: one rank of a world_size-N gloo job that pushes the sharding plan of
tests/dist_plan.py through real collectives on CPU tensors.  Compute is a CPU stand-in (planted hash rows;
runs from the kernel-emulation fixture) — the point is the exchange logic, not the arithmetic."""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import dist_plan as ndist  # noqa: E402


class EmuRun(C.Structure):
    _fields_ = [("src_end", C.c_uint32), ("dst_end", C.c_uint32), ("len", C.c_uint32)]


def library_rows(n, stride, kept):
    rng = np.random.default_rng(1234)
    rows = rng.integers(0, 2 ** 32, (n, stride), dtype=np.uint64).astype(np.uint32)
    intro = rng.integers(0, 2 ** 32, 60, dtype=np.uint64).astype(np.uint32)
    for v in range(n):
        rows[v, 5 + 2 * v: 65 + 2 * v] = intro
        rows[v, kept[v]:] = 0
    return rows


def pair_at(n, index):
    i = 0
    while index >= n - 1 - i:
        index -= n - 1 - i
        i += 1
    return i, i + 1 + index


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    out_path = sys.argv[1]
    n, stride = int(sys.argv[2]), 192
    dist.init_process_group("gloo", rank=rank, world_size=world)
    emu = C.CDLL(os.path.join(ROOT, "tests", "cpu_emu", "libemu.so"))
    emu.emu_hamming_runs.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t]
    emu.emu_hamming_runs.restype = C.c_size_t
    kept = [150 + (7 * v) % 40 for v in range(n)]
    truth = library_rows(n, stride, kept)
    b = ndist.block(n, world)
    arena = torch.zeros((b * world, stride), dtype=torch.int32)

    def analyze_rows(first, count):
        arena[first:first + count] = torch.from_numpy(truth[first:first + count].view(np.int32))

    def search_pairs(pfirst, pcount):
        rows = arena.numpy().view(np.uint32)
        found = []
        buf = (EmuRun * 4096)()
        for p in range(pfirst, pfirst + pcount):
            i, j = pair_at(n, p)
            s = np.ascontiguousarray(rows[i, :kept[i]])
            t = np.ascontiguousarray(rows[j, :kept[j]])
            k = emu.emu_hamming_runs(s.ctypes.data, len(s), t.ctypes.data, len(t), 10, 20, buf, 4096)
            found += [(p, buf[x].src_end, buf[x].dst_end, buf[x].len) for x in range(k)]
        runs = torch.tensor(found, dtype=torch.int32).reshape(-1, 4)
        if os.environ.get("NEEDLE_TEST_SLAB"):      # exercise the single-collective path too
            buf = torch.zeros((4096, 4), dtype=torch.int32)
            buf[: len(runs)] = runs
            return buf, torch.tensor([len(runs)], dtype=torch.int32)
        return runs

    res = ndist.run_job(n, world, rank, arena, analyze_rows, search_pairs, lambda runs: runs.tolist(), lambda: None,
                        slab=int(os.environ.get("NEEDLE_TEST_SLAB", "512")))
    ok_arena = bool(np.array_equal(arena.numpy().view(np.uint32)[:n], truth))
    pipelined = None
    if os.environ.get("NEEDLE_TEST_SLAB"):
        # bench.py's form: reusable gather and row-block buffers, the epilogue deferred to the next job's analyze
        slab = int(os.environ["NEEDLE_TEST_SLAB"])
        gather = ndist.SlabGather(torch.zeros((4096, 4), dtype=torch.int32), world, slab)
        row_block = torch.zeros((b, stride), dtype=torch.int32)
        done, pending = [], []

        def finish_previous():
            if pending:
                done.append(pending.pop().tolist())

        for _ in range(3):
            arena.zero_()
            pending.append(ndist.run_job(n, world, rank, arena, analyze_rows, search_pairs, None, lambda: None,
                                         slab=slab, gather=gather, defer_finalize=True,
                                         while_analyzing=finish_previous, row_block=row_block))
        finish_previous()
        pipelined = done
        # and the prefetching pipeline (next job's analyze issued before this job's run list is awaited)
        pipe = ndist.JobPipeline(n, world, rank, arena, analyze_rows, search_pairs, lambda runs: runs.tolist(), gather,
                                 row_block)
        pipelined = pipelined + [pipe.step(prefetch=(k < 2)) or pipelined[0] for k in range(3)]
    with open(f"{out_path}.{rank}", "w") as f:
        json.dump({"rank": rank, "arena_complete": ok_arena, "runs": res, "pipelined": pipelined}, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
