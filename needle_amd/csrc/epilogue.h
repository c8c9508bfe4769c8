// Device form of the per-video epilogue of a library job (epilogue.hip); the host form is comparator.cpp.
#pragma once

#include <hip/hip_runtime_api.h>

#include <atomic>
#include <cstdint>
#include <vector>

#include "../../include/needle_hip.h"
#include "common.h"

namespace needle {

struct EpilogueJob {
  int slot = 0;                      // job slot: two jobs in flight keep two workspaces
  uint32_t n = 0;                    // videos
  uint32_t regions = 1;              // of the COMPARATOR: NeedleHipRun.problem = pair * regions + region
  uint32_t rows_per_video = 1;       // of the library's hash arena: row = video * rows_per_video + region
  uint32_t v0 = 0, v1 = 0;           // results wanted for videos [v0, v1)
  uint32_t threshold = 0;
  bool include_endings = false;
  ns_t min_opening_duration = 0, min_ending_duration = 0, time_padding = 0, hash_duration = 0;
  // the run list: num_segments pieces in order (a rank's slab, the gathered heads in rank order, or the buffer of a
  // host-side search call), each a device word holding the runs found and up to segment_capacity runs
  const uint32_t *segment_count[64] = {nullptr};
  const NeedleHipRun *segment_runs[64] = {nullptr};
  int num_segments = 0;
  uint32_t segment_capacity = 0;
  uint32_t segment_capacities[64] = {0};  // per segment where they differ (the owner-directed exchange's blocks); 0 = segment_capacity
  uint64_t max_runs = 0;             // upper bound of the runs in all segments (sizes the workspaces)
  // per arena row: hashes kept, offset of its (un-seeked) timestamps in `ts`, seek added to each of them
  const std::vector<uint32_t> *row_len = nullptr, *row_ts = nullptr;
  const std::vector<uint64_t> *row_seek = nullptr, *ts = nullptr;
};

// The device form gives a pair's runs to ONE lane (insertion order + heap): a pair with more runs than this (silence against
// silence, one sustained tone against itself) would keep that lane busy for seconds; such a bucket is a workgroup's (below).
// Bit 31 of the failure word: the caller computes the results with the host form (threaded, n log n) from the run list.
// NEEDLE_HIP_EPILOGUE_BUCKET_LIMIT is not a switch: tests lower the constant by building with -D.
#ifndef NEEDLE_EPILOGUE_BUCKET_LIMIT
#define NEEDLE_EPILOGUE_BUCKET_LIMIT 24
#endif
constexpr uint32_t kEpilogueBucketLimit = NEEDLE_EPILOGUE_BUCKET_LIMIT;
// Round 6: buckets beyond that limit (up to kEpilogueLargeLimit runs: two fully silent 24-minute windows are 5 800) go to a
// WORKGROUP each (pair_entries_large_kernel: bitonic sort into the walk order and the heap's sift-ups on packed 64-bit keys in
// LDS) instead of failing the job over to the host.  Only a bucket beyond kEpilogueLargeLimit, a row of 65 536 hashes or more, or
// timestamps that do not strictly increase (the packed key stands for them) still set the bit.
#ifndef NEEDLE_EPILOGUE_LARGE_LIMIT
#define NEEDLE_EPILOGUE_LARGE_LIMIT 8192
#endif
constexpr uint32_t kEpilogueLargeLimit = NEEDLE_EPILOGUE_LARGE_LIMIT;
constexpr uint32_t kEpilogueBucketTooLarge = 0x80000000u;
// Jobs whose device epilogue was handed back to the host because of that bit (needle_hip_epilogue_host_fallbacks): a
// silent performance cliff otherwise -- one pair of silent stretches moves a whole library's epilogue to the host.
std::atomic<uint64_t> &epilogue_host_fallbacks();
// (host side of both callers: counts, and says so under NEEDLE_HIP_TRACE)
void note_epilogue_host_fallback(const char *where, size_t runs, size_t videos);

// The owner-directed exchange's send side (round 6, library.cpp): the runs of a rank's slab sorted into one block per
// destination rank -- a run of pair (i, j) goes to the owners of video i and of video j (videos in blocks of `videos_per_rank`:
// the sharded epilogue's blocks, comparator.rs:583-588 needs a video's pairs and nothing else).  A block = a slab: 32-byte
// header (word 0: runs DIRECTED to it, which may exceed what it holds -- the caller's overflow test) + capacity[q] runs, at
// send + offset[q] bytes.  Enqueued on `stream`; the headers are cleared first.
struct DirectPlan {
  uint32_t offset[64];    // bytes
  uint32_t capacity[64];  // runs
};
Status gpu_direct_runs(const uint32_t *d_found, const NeedleHipRun *d_runs, uint32_t slab_capacity, uint32_t n, uint32_t regions,
                       uint32_t videos_per_rank, int world, uint8_t *d_send, const DirectPlan &plan, hipStream_t stream);

// Enqueues the epilogue kernels on `stream` behind whatever fills the segments, then the copies of results[n] and of the
// failure count (videos whose padding / hash duration exceed the match end: the reference panics) into HOST memory
// (pinned: the copies are asynchronous).  Counts beyond a segment's capacity are clamped: the host redoes such a job.
Status gpu_epilogue_enqueue(const EpilogueJob &job, hipStream_t stream, NeedleHipSearchResult *host_results, uint32_t *host_failed);

// Comparator::run_with_frame_hashes with both halves on the device (search.hip): host hash arena in, scan + simhash, the
// epilogue above on the run list where it lies, the n per-video results out; the run list itself never crosses PCIe.
// `job` carries everything but the segments.  *failed != 0 (a failing video, or kEpilogueBucketTooLarge): the caller falls
// back to the host epilogue (which reports the video that fails, in the reference's order) -- `runs` then holds the
// downloaded list.
Status gpu_search_results_host(const uint32_t *hashes, size_t num_hashes, const NeedleHipSeq *seqs, size_t num_seqs,
                               const NeedleHipProblem *problems, size_t num_problems, uint32_t threshold, EpilogueJob job,
                               std::vector<NeedleHipSearchResult> *results, uint32_t *failed, std::vector<NeedleHipRun> *runs,
                               size_t *num_runs);

}  // namespace needle
