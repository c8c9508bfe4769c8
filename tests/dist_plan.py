"""TEST HELPER (not product code): sharding plan and exchange steps of a multi-GPU analyze+search job (one process per GPU).

The reference parallelises with rayon over videos (analyzer.rs:440-444) and over pairs
(comparator.rs:553-563); both are independent units, so across G GPUs:

  1. videos are split into G contiguous blocks; rank r fingerprints block r into rows of a padded
     arena u32[rows][stride] (rows = G * block);
  2. ONE all-gather of row blocks gives every rank every hash row (RCCL over xGMI on GPUs);
  3. the lexicographic pair list (comparator.rs:534-545) is split into G contiguous ranges; rank r scans
     its range;
  4. run lists are variable length: ranks all-gather their counts, pad to the largest and all-gather the
     runs (the "final cross-shard pair list" of north_star); rank 0 runs the order-sensitive epilogue.

This module is backend-agnostic plumbing over torch.distributed tensors (nccl on GPUs, gloo on CPU in
the tests); the compute steps are callables supplied by the caller (bench.py passes the C-ABI library
calls; the gloo test passes CPU stand-ins).  Results are identical for every G by construction: the run
set is a union over disjoint pair ranges and the epilogue sorts it.
"""
from __future__ import annotations

from typing import Callable, List, Tuple

import numpy as np


def block(n: int, world: int) -> int:
    """Rows per rank so that world * block >= n."""
    return (n + world - 1) // world


def shard(n: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous [first, first+count) of n units owned by `rank`."""
    b = block(n, world)
    first = min(rank * b, n)
    return first, min(b, n - first)


def pair_count(n: int) -> int:
    return n * (n - 1) // 2 if n >= 2 else 0


def gather_rows(arena, world: int, rank: int, row_block=None):
    """In-place all-gather of the row blocks of `arena` (torch tensor [world*block, stride]); rank r's block
    must already hold its rows.  `row_block`: optional preallocated [block, stride] staging tensor."""
    import torch.distributed as dist
    if world == 1:
        return arena
    b = arena.shape[0] // world
    mine = arena[rank * b:(rank + 1) * b]
    if row_block is None:
        row_block = mine.clone()
    else:
        row_block.copy_(mine)
    dist.all_gather_into_tensor(arena, row_block)
    return arena


def gather_runs(local_runs, world: int):
    """All-gather variable-length run lists.  `local_runs` is a torch tensor [k, words] (u32 stored as int32) on the
    collective's device.  Returns the concatenation over ranks, in rank order, on the same device."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return local_runs
    count = torch.tensor([local_runs.shape[0]], dtype=torch.int64, device=local_runs.device)
    counts = torch.zeros(world, dtype=torch.int64, device=local_runs.device)
    dist.all_gather_into_tensor(counts, count)
    counts = counts.cpu().tolist()
    width = max(max(counts), 1)
    cols = local_runs.shape[1]
    padded = torch.zeros((width, cols), dtype=local_runs.dtype, device=local_runs.device)
    padded[: local_runs.shape[0]] = local_runs
    out = torch.zeros((world * width, cols), dtype=local_runs.dtype, device=local_runs.device)
    dist.all_gather_into_tensor(out, padded)
    return torch.cat([out[r * width: r * width + counts[r]] for r in range(world)], dim=0)


class SlabGather:
    """Latency-lean gather for short run lists: ONE fixed-size all-gather and no host round trip before it.
    Each rank contributes a slab of 1 + `slab` rows whose row 0 carries its run count (which may exceed the slab);
    buffers are allocated once and reused by every job."""

    def __init__(self, run_buffer, world: int, slab: int = 512):
        import torch
        self.world, self.slab, self.cols = world, slab, run_buffer.shape[1]
        self.mine = torch.zeros((1 + slab, self.cols), dtype=run_buffer.dtype, device=run_buffer.device)
        self.out = torch.zeros((world, 1 + slab, self.cols), dtype=run_buffer.dtype, device=run_buffer.device)
        self.host = torch.zeros((world, 1 + slab, self.cols), dtype=run_buffer.dtype)
        if run_buffer.device.type == "cuda":
            self.host = self.host.pin_memory()

    def __call__(self, run_buffer, count_tensor):
        """`run_buffer` [capacity, words] and `count_tensor` [1] (int32, total runs found) live on the collective's
        device.  Returns (runs ndarray [total, words], complete): `complete` is False when some rank found more than
        `slab` runs, in which case the caller falls back to gather_runs."""
        import torch.distributed as dist
        self.mine[0, 0:1] = count_tensor
        self.mine[1:] = run_buffer[: self.slab]
        if self.world == 1:
            self.out[0] = self.mine
        else:
            dist.all_gather_into_tensor(self.out.view(self.world * (1 + self.slab), self.cols), self.mine)
        self.host.copy_(self.out)          # device -> (pinned) host, synchronous for the host
        host = self.host.numpy()
        counts = [int(host[r, 0, 0]) for r in range(self.world)]
        complete = all(c <= self.slab for c in counts)
        runs = np.concatenate([host[r, 1:1 + min(counts[r], self.slab)] for r in range(self.world)], axis=0)
        return runs, complete


def gather_runs_slab(run_buffer, count_tensor, world: int, slab: int = 512):
    """One-shot form of SlabGather."""
    return SlabGather(run_buffer, world, slab)(run_buffer, count_tensor)


def run_job(n_videos: int, world: int, rank: int, arena, analyze_rows: Callable[[int, int], None],
            search_pairs: Callable[[int, int], "object"], finalize: Callable[[np.ndarray], "object"],
            sync: Callable[[], None], slab: int = 512, gather=None, defer_finalize: bool = False,
            while_analyzing: Callable[[], None] = None, row_block=None):
    """One analyze+search pass.  analyze_rows(first, count) fills this rank's arena rows; search_pairs(first,
    count) returns this rank's runs, either as a torch tensor [k, words] or as a (run_buffer, count_tensor) pair
    of device tensors for the single-collective slab gather (`gather`: a reusable SlabGather); finalize(runs
    ndarray) builds the per-video results (rank 0 only; other ranks get None).  sync() orders the compute stream
    against the collective.  while_analyzing() runs on the host right after this rank's analyze work has been
    enqueued (the caller's place for the previous job's epilogue); with defer_finalize the gathered runs are
    returned instead of the results, for the caller to finalize later.  row_block: a reusable staging tensor for
    this rank's row block (the all-gather's input must not alias its output)."""
    first, count = shard(n_videos, world, rank)
    if count:
        analyze_rows(first, count)
    if while_analyzing is not None:
        while_analyzing()
    sync()
    gather_rows(arena, world, rank, row_block)
    sync()
    pfirst, pcount = shard(pair_count(n_videos), world, rank)
    local = search_pairs(pfirst, pcount)
    sync()
    if isinstance(local, tuple):
        run_buffer, count_tensor = local
        if gather is None:
            gather = SlabGather(run_buffer, world, slab)
        runs, complete = gather(run_buffer, count_tensor)
        if not complete:  # a rank overflowed its slab: exact two-step gather
            runs = gather_runs(run_buffer[: int(count_tensor.item())], world).cpu().numpy()
    else:
        runs = gather_runs(local, world).cpu().numpy()
    if defer_finalize:
        return runs
    if rank != 0:
        return None
    return finalize(runs)


class JobPipeline:
    """Repeated analyze+search jobs over the same sharding plan, with job k+1's fingerprinting enqueued before
    the host blocks on job k's gathered run list: the device then works on the next job's fingerprints while the
    run-list collective, its download and rank 0's epilogue of job k proceed.  All device work is issued in ONE
    stream order (analyze_k, rows gather_k, search_k, analyze_k+1, ...), so a single hash arena and a single run
    buffer suffice: analyze_k+1 runs after search_k on the device, and search_k+1 is enqueued only after job k's
    runs have reached the host.

    step(prefetch) completes one job -- its analyze was enqueued by the previous step's prefetch, or is enqueued
    now -- and returns its results (rank 0; None elsewhere).  A caller that times K steps passes prefetch=False
    on the last of them, so exactly K analyses and K searches are issued inside the timed steps."""

    def __init__(self, n_videos: int, world: int, rank: int, arena, analyze_rows, search_pairs, finalize,
                 gather: "SlabGather", row_block=None, side_stream=None):
        """side_stream (GPU only): a second stream for the run-list staging copies, collective and download.  In
        the main stream they would queue behind the prefetched fingerprint kernels and the host would wait for
        those too; on the side stream they wait only for an event recorded after the search."""
        self.n, self.world, self.rank, self.arena = n_videos, world, rank, arena
        self.analyze_rows, self.search_pairs, self.finalize = analyze_rows, search_pairs, finalize
        self.gather, self.row_block, self.side = gather, row_block, side_stream
        self.analyzed = False
        self.results = None

    def _analyze(self):
        first, count = shard(self.n, self.world, self.rank)
        if count:
            self.analyze_rows(first, count)
        self.analyzed = True

    def _gather_runs(self, run_buffer, count_tensor):
        runs, complete = self.gather(run_buffer, count_tensor)   # blocks the host until this job's runs are here
        if not complete:
            # a rank overflowed its slab: exact two-step gather.  The run buffer is intact (the prefetched analyze
            # does not touch it).
            runs = gather_runs(run_buffer[: int(count_tensor.item())], self.world).cpu().numpy()
        return runs

    def step(self, prefetch: bool = True):
        if not self.analyzed:
            self._analyze()
        gather_rows(self.arena, self.world, self.rank, self.row_block)
        pfirst, pcount = shard(pair_count(self.n), self.world, self.rank)
        run_buffer, count_tensor = self.search_pairs(pfirst, pcount)
        self.analyzed = False
        if self.side is None:
            if prefetch:
                self._analyze()
            runs = self._gather_runs(run_buffer, count_tensor)
        else:
            import torch
            searched = torch.cuda.Event()
            searched.record()                    # main stream: after this job's search (and simhash) kernels
            if prefetch:
                self._analyze()                  # next job's fingerprints run under the gather below
            with torch.cuda.stream(self.side):
                self.side.wait_event(searched)
                runs = self._gather_runs(run_buffer, count_tensor)
        if self.rank == 0:
            self.results = self.finalize(runs)
        return self.results
