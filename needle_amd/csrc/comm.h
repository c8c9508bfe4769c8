// Multi-GPU communicator of libneedle_capi.so: one process per GPU, collectives over RCCL (xGMI).
//
// The reference parallelises inside one process with rayon (needle/src/audio/analyzer.rs:437-445 over videos,
// comparator.rs:549-564 over pairs).  Across GPUs the same two fan-outs become: videos in contiguous blocks per
// rank, pairs in contiguous ranges of the lexicographic pair list per rank, with two all-gathers in between
// (hash rows after analyze, run lists after search).  This file is the transport: an all-gather on device
// buffers in stream order.  librccl is loaded with dlopen only when a communicator is created (it is a 570 MB
// library that a single-GPU process never needs), so the product has no link-time dependency on it and none on
// torch.  A second, host-staged backend (POSIX shared memory, NEEDLE_HIP_COMM=host) carries the same collectives
// between processes that share ONE device: it exists so that the N-rank code path can be exercised on a one-GPU
// box, and as the fallback SURVEY.md §8(e) allows; it is not a compute fallback.
#pragma once

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cstddef>
#include <cstdint>

#include "common.h"

namespace needle {

struct Comm;  // opaque

// The process-wide communicator (nullptr: single rank).  rank / world of a null communicator are 0 / 1.
Comm *comm_get();
int comm_rank();
int comm_world();
const char *comm_backend();  // "none", "rccl", "host"

Status comm_create_id(uint8_t id[128]);                      // rank 0; backend from NEEDLE_HIP_COMM
Status comm_init(const uint8_t id[128], int rank, int world);  // collective over all ranks; binds the current device
void comm_finalize();

// Which channel a collective belongs to: kData is used on the library stream (hash rows), kSide on the download
// stream (run lists, results).  Both map to ONE RCCL communicator unless NEEDLE_HIP_COMM_DUAL=1 (comm.cpp comm_init).
enum CommChannel { kData = 0, kSide = 1 };

// All-gather of `bytes` per rank in stream order: rank r's block lands at d_recv + r * bytes on every rank.
// In place when d_send == d_recv + rank * bytes.  `bytes` must be a multiple of 4.
Status comm_all_gather(CommChannel ch, const void *d_send, void *d_recv, size_t bytes, hipStream_t stream);
// All-to-all of blocks of DIFFERENT sizes in stream order (round 6: the owner-directed run exchange): bytes
// send_bytes[q] at d_send + send_off[q] go to rank q, recv_bytes[r] arrive from rank r at d_recv + recv_off[r]; sizes are
// multiples of 4 and known on the host of every rank (recv_bytes[r] here = send_bytes[me] on rank r); max_block_bytes =
// the largest block any rank sends in this call (the same number on every rank: the host transport steps by it).
// RCCL: one group of ncclSend / ncclRecv per peer; NeedleError_InvalidArgument with "unsupported" in the message when
// the loaded librccl lacks them (the caller keeps its all-gather).
Status comm_all_to_all_v(CommChannel ch, const void *d_send, const size_t *send_off, const size_t *send_bytes, void *d_recv,
                         const size_t *recv_off, const size_t *recv_bytes, size_t max_block_bytes, hipStream_t stream);
// Host buffers (staged through a small device buffer on the side channel); synchronous.
Status comm_all_gather_host(const void *send, void *recv, size_t bytes);
Status comm_barrier();

// Contiguous block plan: `units` split into `world` blocks of ceil(units / world); rank's [first, first + count).
inline size_t shard_block(size_t units, int world) { return world > 0 ? (units + (size_t)world - 1) / (size_t)world : units; }
inline void shard_range(size_t units, int world, int rank, size_t *first, size_t *count) {
  const size_t b = shard_block(units, world);
  const size_t f = std::min(units, (size_t)rank * b);
  *first = f;
  *count = std::min(b, units - f);
}

}  // namespace needle
