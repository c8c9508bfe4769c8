"""Sharding plan and exchange steps of a multi-GPU analyze+search job (one process per GPU).

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


def gather_rows(arena, world: int, rank: int):
    """In-place all-gather of the row blocks of `arena` (torch tensor [world*block, stride]); rank r's block
    must already hold its rows."""
    import torch.distributed as dist
    if world == 1:
        return arena
    b = arena.shape[0] // world
    mine = arena[rank * b:(rank + 1) * b]
    dist.all_gather_into_tensor(arena, mine.clone())
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


def run_job(n_videos: int, world: int, rank: int, arena, analyze_rows: Callable[[int, int], None],
            search_pairs: Callable[[int, int], "object"], finalize: Callable[[np.ndarray], "object"],
            sync: Callable[[], None]):
    """One analyze+search pass.  analyze_rows(first, count) fills this rank's arena rows; search_pairs(first,
    count) returns this rank's runs as a torch tensor [k, 4]; finalize(runs ndarray) builds the per-video
    results (rank 0 only; other ranks get None).  sync() orders the compute stream against the collective."""
    first, count = shard(n_videos, world, rank)
    if count:
        analyze_rows(first, count)
    sync()
    gather_rows(arena, world, rank)
    sync()
    pfirst, pcount = shard(pair_count(n_videos), world, rank)
    local = search_pairs(pfirst, pcount)
    sync()
    merged = gather_runs(local, world)
    if rank != 0:
        return None
    return finalize(merged.cpu().numpy())
