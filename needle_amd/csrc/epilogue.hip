// The order-sensitive epilogue of a library job ON THE DEVICE: from the complete run list of all pairs to one
// SearchResult per video (needle/src/audio/comparator.rs:191-249 validity + BinaryHeap order, :405-515 find_best_match,
// :583-626 per-video walk), bit for bit what comparator.cpp computes on host threads.  At BASELINE.json configs[4]
// (2000 x 45 min: 3.95 M runs, ~3 950 candidates per video) the host form takes 68 ms on 16 threads -- and every rank of
// an 8-GPU job has a sixteenth of those threads -- while the work is 3e10 popcount-compares and a few small sorts.
//
//   bucket_count / scan / bucket_scatter : counting sort of the runs by problem (pair * regions + region)
//   pair_entries   : one thread per bucket -- the reference's reverse table walk order (src_end, then dst_end, descending),
//                    the duration validity tests (:212-223), std's BinaryHeap::push (append + sift up on the derived
//                    lexicographic Ord, :22-35) -> the bucket's entries in the heap's backing-array order (:249)
//   best_match     : one workgroup per video -- its pairs in lexicographic order, openings before endings inside a
//                    pair (:414-431) = the candidate numbering; links[k] = #{b : popcount(h_k ^ h_b) < bound} over ALL
//                    its candidates (:434-454): every pair, as int8 matrix products of the hashes' bits as +-1 whose
//                    sign bits are counted (exact: a dot product of +-1 bytes IS 32 - 2 popcount); score = -(links * 0.3f + secs * 0.7f) in unfused f32
//                    (:469); arg-min over (score, index) (:473-475); padding and hash duration (:479-481)
//
// Nothing here approximates: ties are broken by the candidate index exactly as the sorted (f32, usize) list of the
// reference breaks them.  tests: the GPU suite and tools/fuzz_pipeline.py with NEEDLE_HIP_DEVICE_EPILOGUE=1.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <map>
#include <mutex>

#include "../../include/needle_hip.h"
#include "epilogue.h"
#include "hipctx.h"

namespace needle {

namespace {

constexpr int kMaxSegments = 64;

struct RunSegments {  // the run list in pieces: a device word with the runs found, and the runs
  const uint32_t *found[kMaxSegments];
  const NeedleHipRun *runs[kMaxSegments];
  uint32_t capacity[kMaxSegments];
  int count;
};

struct DeviceEntry {  // ComparatorHeapEntry (:22-35) without the fields that are constant inside a bucket
  uint64_t src_start, src_end, dst_start, dst_end;
  uint32_t score, src_hash, dst_hash, pad;
};

struct Candidate {  // :410-432
  uint64_t start, end;
  uint32_t hash, is_opening;
};

struct EpilogueParams {
  uint32_t n, regions, buckets;          // videos, comparator regions, np * regions
  uint32_t large_ok;                     // buckets beyond kEpilogueBucketLimit may go to pair_entries_large_kernel (see there)
  uint32_t rows_per_video;               // rows of the hash arena per video
  uint32_t v0, v1;                       // the videos whose results are wanted
  uint32_t bound;                        // threshold + threshold / 2 (:441)
  uint32_t include_endings;
  uint64_t min_duration[2];              // [0] opening, [1] ending
  uint64_t time_padding, hash_duration;
};

__device__ __forceinline__ uint32_t segment_count(const RunSegments &s, int k) {
  return min(*s.found[k], s.capacity[k]);  // an overflowed slab is redone by the host
}

// run g of the concatenated list (segments in rank order)
__device__ __forceinline__ bool locate_run(const RunSegments &s, uint64_t g, NeedleHipRun *out) {
  for (int k = 0; k < s.count; k++) {
    const uint32_t c = segment_count(s, k);
    if (g < c) {
      *out = s.runs[k][g];
      return true;
    }
    g -= c;
  }
  return false;
}

__global__ __launch_bounds__(256) void bucket_count_kernel(RunSegments segs, uint32_t buckets, uint32_t *__restrict__ count) {
  uint64_t total = 0;
  for (int k = 0; k < segs.count; k++) total += segment_count(segs, k);
  for (uint64_t g = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; g < total; g += (uint64_t)gridDim.x * blockDim.x) {
    NeedleHipRun r;
    if (locate_run(segs, g, &r) && r.problem < buckets) atomicAdd(&count[r.problem], 1u);
  }
}

// exclusive scan of count[0..n) into start[0..n], start[n] = total: per-block sums, one block over them, then the blocks
constexpr int kScanBlock = 1024;
__global__ __launch_bounds__(256) void scan_block_sums_kernel(const uint32_t *__restrict__ count, uint32_t n, uint32_t *__restrict__ sums) {
  __shared__ uint32_t part[256];
  const uint32_t base = blockIdx.x * kScanBlock;
  uint32_t s = 0;
  for (int k = 0; k < 4; k++) {
    const uint32_t i = base + threadIdx.x * 4 + k;
    s += i < n ? count[i] : 0u;
  }
  part[threadIdx.x] = s;
  __syncthreads();
  for (int d = 128; d > 0; d >>= 1) {
    if ((int)threadIdx.x < d) part[threadIdx.x] += part[threadIdx.x + d];
    __syncthreads();
  }
  if (threadIdx.x == 0) sums[blockIdx.x] = part[0];
}
__global__ __launch_bounds__(256) void scan_sums_kernel(uint32_t *__restrict__ sums, uint32_t blocks) {  // one workgroup
  __shared__ uint32_t carry;
  __shared__ uint32_t part[256];
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (uint32_t base = 0; base < blocks; base += 256) {
    const uint32_t i = base + threadIdx.x;
    const uint32_t v = i < blocks ? sums[i] : 0u;
    part[threadIdx.x] = v;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {  // inclusive Hillis-Steele
      const uint32_t add = (int)threadIdx.x >= d ? part[threadIdx.x - d] : 0u;
      __syncthreads();
      part[threadIdx.x] += add;
      __syncthreads();
    }
    if (i < blocks) sums[i] = carry + part[threadIdx.x] - v;  // exclusive
    __syncthreads();
    if (threadIdx.x == 255) carry += part[255];
    __syncthreads();
  }
}
__global__ __launch_bounds__(256) void scan_apply_kernel(const uint32_t *__restrict__ count, uint32_t n, const uint32_t *__restrict__ sums,
                                                         uint32_t *__restrict__ start) {
  __shared__ uint32_t part[256];
  const uint32_t base = blockIdx.x * kScanBlock;
  uint32_t v[4], s = 0;
  for (int k = 0; k < 4; k++) {
    const uint32_t i = base + threadIdx.x * 4 + k;
    v[k] = i < n ? count[i] : 0u;
    s += v[k];
  }
  part[threadIdx.x] = s;
  __syncthreads();
  for (int d = 1; d < 256; d <<= 1) {
    const uint32_t add = (int)threadIdx.x >= d ? part[threadIdx.x - d] : 0u;
    __syncthreads();
    part[threadIdx.x] += add;
    __syncthreads();
  }
  uint32_t run = sums[blockIdx.x] + part[threadIdx.x] - s;
  for (int k = 0; k < 4; k++) {
    const uint32_t i = base + threadIdx.x * 4 + k;
    if (i < n) start[i] = run;
    run += v[k];
    if (i + 1 == n) start[n] = run;
  }
}

__global__ __launch_bounds__(256) void bucket_scatter_kernel(RunSegments segs, uint32_t buckets, const uint32_t *__restrict__ start,
                                                             uint32_t *__restrict__ fill, NeedleHipRun *__restrict__ sorted) {
  uint64_t total = 0;
  for (int k = 0; k < segs.count; k++) total += segment_count(segs, k);
  for (uint64_t g = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; g < total; g += (uint64_t)gridDim.x * blockDim.x) {
    NeedleHipRun r;
    if (locate_run(segs, g, &r) && r.problem < buckets) sorted[start[r.problem] + atomicAdd(&fill[r.problem], 1u)] = r;
  }
}

// pairs (i, j), i < j, i-major (comparator.rs:534-545): the inverse of the numbering, exact (hostutil's pair_at)
__device__ __forceinline__ uint64_t row_start(uint64_t n, uint64_t i) { return i * (2 * n - i - 1) / 2; }
__device__ __forceinline__ void pair_at_device(uint64_t n, uint64_t index, uint32_t *pi, uint32_t *pj) {
  const double b = 2.0 * (double)n - 1.0;
  const double disc = b * b - 8.0 * (double)index;
  uint64_t i = disc > 0.0 ? (uint64_t)((b - sqrt(disc)) / 2.0) : 0;
  if (i + 2 > n) i = n >= 2 ? n - 2 : 0;
  while (i > 0 && row_start(n, i) > index) i--;
  while (i + 2 < n && row_start(n, i + 1) <= index) i++;
  *pi = (uint32_t)i;
  *pj = (uint32_t)(i + 1 + (index - row_start(n, i)));
}

// A slab's runs into one block per destination rank (gpu_direct_runs).  A wave asks a block's counter once per destination
// for all its lanes' runs (a returning atomic per run on `world` addresses would be the kernel).
__global__ __launch_bounds__(256) void direct_runs_kernel(const uint32_t *__restrict__ found, const NeedleHipRun *__restrict__ runs,
                                                          uint32_t slab_capacity, uint32_t n, uint32_t regions, uint32_t videos_per_rank,
                                                          int world, uint8_t *__restrict__ send, DirectPlan plan) {
  const uint32_t total = min(*found, slab_capacity);
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t base = blockIdx.x * blockDim.x; base < total; base += stride) {  // (whole waves stay in the loop: ballots)
    const uint32_t g = base + threadIdx.x;
    NeedleHipRun r{};
    uint32_t oi = 0xFFFFFFFFu, oj = 0xFFFFFFFFu;
    if (g < total) {
      r = runs[g];
      uint32_t vi, vj;
      pair_at_device(n, r.problem / regions, &vi, &vj);
      oi = vi / videos_per_rank;
      oj = vj / videos_per_rank;
      if (oj == oi) oj = 0xFFFFFFFFu;
    }
    for (int q = 0; q < world; q++) {
      const bool mine = oi == (uint32_t)q || oj == (uint32_t)q;
      const unsigned long long mask = __builtin_amdgcn_ballot_w64(mine);
      if (mask == 0ull) continue;  // wave-uniform
      uint32_t *header = reinterpret_cast<uint32_t *>(send + plan.offset[q]);
      uint32_t first = 0u;
      if (lane == (uint32_t)(__ffsll((long long)mask) - 1)) first = atomicAdd(header, (uint32_t)__popcll(mask));
      first = (uint32_t)__shfl((int)first, __ffsll((long long)mask) - 1);
      if (mine) {
        const uint32_t at = first + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
        if (at < plan.capacity[q]) reinterpret_cast<NeedleHipRun *>(send + plan.offset[q] + 32)[at] = r;
      }
    }
  }
}

// #[derive(Ord)] over (score, src_start, src_end, dst_start, dst_end, src_match_hash, dst_match_hash, ...): the rest of
// the fields are equal for all entries of one bucket.  true: a > b.
__device__ __forceinline__ bool entry_greater(const DeviceEntry &a, const DeviceEntry &b) {
  if (a.score != b.score) return a.score > b.score;
  if (a.src_start != b.src_start) return a.src_start > b.src_start;
  if (a.src_end != b.src_end) return a.src_end > b.src_end;
  if (a.dst_start != b.dst_start) return a.dst_start > b.dst_start;
  if (a.dst_end != b.dst_end) return a.dst_end > b.dst_end;
  if (a.src_hash != b.src_hash) return a.src_hash > b.src_hash;
  return a.dst_hash > b.dst_hash;
}

// One thread per bucket.  row tables: length, offset of the row's timestamps in `ts` (un-seeked, shared by rows of equal
// length), seek added to every timestamp of the row.
__global__ __launch_bounds__(64) void pair_entries_kernel(EpilogueParams pr, const uint32_t *__restrict__ start,
                                                          NeedleHipRun *__restrict__ sorted, const uint32_t *__restrict__ row_len,
                                                          const uint32_t *__restrict__ row_ts, const uint64_t *__restrict__ row_seek,
                                                          const uint64_t *__restrict__ ts, DeviceEntry *__restrict__ entries,
                                                          uint32_t *__restrict__ valid, uint32_t *__restrict__ failed,
                                                          uint32_t *__restrict__ large_count, uint32_t *__restrict__ large_list) {
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= pr.buckets) return;
  const uint32_t lo = start[b], hi = start[b + 1];
  uint32_t out = 0;
  if (hi - lo > kEpilogueBucketLimit) {
    // One lane orders a bucket by insertion (quadratic) and builds its heap alone: right for the handful of runs a pair of
    // episodes has, not for the hundreds two stretches of silence or of one sustained tone produce (an S x S block of equal
    // hashes is ~2 S runs).  Such a bucket goes on the list of pair_entries_large_kernel (a workgroup each; the list cannot
    // overflow: every entry stands for more than kEpilogueBucketLimit of the runs it was sized by); beyond what that kernel
    // holds in LDS the library is handed back to the host form (threaded, n log n): bit 31 of `failed`.
    if (pr.large_ok && hi - lo <= kEpilogueLargeLimit) large_list[atomicAdd(large_count, 1u)] = b;
    else atomicOr(failed, kEpilogueBucketTooLarge);
    return;  // (valid[b]: the large kernel's)
  } else if (hi > lo) {
    // the reference walks its table backwards: i = n-1..1 and, inside, j = m-1..1 (:191-192)
    for (uint32_t a = lo + 1; a < hi; a++) {
      const NeedleHipRun x = sorted[a];
      uint32_t q = a;
      while (q > lo) {
        const NeedleHipRun y = sorted[q - 1];
        const bool before = y.src_end != x.src_end ? y.src_end > x.src_end : y.dst_end > x.dst_end;
        if (before) break;
        sorted[q] = y;
        q--;
      }
      sorted[q] = x;
    }
    const uint32_t region = b % pr.regions;
    uint32_t vi, vj;
    pair_at_device(pr.n, b / pr.regions, &vi, &vj);
    const uint32_t src_row = vi * pr.rows_per_video + region, dst_row = vj * pr.rows_per_video + region;
    const uint32_t src_len = row_len[src_row], dst_len = row_len[dst_row];
    const uint64_t *src_ts = ts + row_ts[src_row], *dst_ts = ts + row_ts[dst_row];
    const uint64_t src_seek = row_seek[src_row], dst_seek = row_seek[dst_row];
    const uint64_t min_duration = pr.min_duration[region];
    DeviceEntry *heap = entries + lo;
    for (uint32_t a = lo; a < hi; a++) {
      const NeedleHipRun r = sorted[a];
      const uint32_t i = r.src_end, j = r.dst_end, len = r.len;
      if (len == 0 || len > i || len > j || i >= src_len || j >= dst_len) continue;
      DeviceEntry e;
      e.src_start = src_ts[i - len] + src_seek;  // one BEFORE the first matched cell (:206-207)
      e.src_end = src_ts[i] + src_seek;
      e.dst_start = dst_ts[j - len] + dst_seek;
      e.dst_end = dst_ts[j] + dst_seek;
      if (e.src_end < e.src_start || e.dst_end < e.dst_start) continue;
      if (e.src_end - e.src_start < min_duration || e.dst_end - e.dst_start < min_duration) continue;  // :212-223
      e.score = len;
      e.src_hash = r.src_match_hash;
      e.dst_hash = r.dst_match_hash;
      e.pad = 0;
      // BinaryHeap::push: append, sift up while greater than the parent
      uint32_t pos = out++;
      while (pos > 0) {
        const uint32_t parent = (pos - 1) / 2;
        const DeviceEntry p = heap[parent];
        if (!entry_greater(e, p)) break;
        heap[pos] = p;
        pos = parent;
      }
      heap[pos] = e;
    }
  }
  valid[b] = out;
}

// One WORKGROUP per bucket of more than kEpilogueBucketLimit runs (round 6; the hostile corpus: silence against silence).
// The same three steps as the lane above, on packed keys in LDS (8 bytes per run):
//   1. walk order: bitonic sort by (src_end, dst_end) descending -- key (0xFFFF - src_end) << 16 | (0xFFFF - dst_end), the
//      run's index in the low word;
//   2. every thread turns its sorted elements into (rank << 16 | valid << 15 | index): rank = len << 32 | (src_end - len) << 16
//      | dst_end IS the derived Ord of :22-35 inside one bucket -- score = len, and with timestamps that strictly increase along
//      a row (checked on the host: large_ok) src_start / src_end / dst_start / dst_end order as src_end - len, src_end,
//      dst_end - len, dst_end; two runs of a bucket never share (src_end, dst_end), so the hashes are never reached;
//   3. ONE lane replays BinaryHeap::push over the valid elements in walk order, in place (the heap never holds more than the
//      elements already consumed); then every thread builds the DeviceEntry of its heap slots.
__global__ __launch_bounds__(256) void pair_entries_large_kernel(EpilogueParams pr, const uint32_t *__restrict__ start,
                                                                 const NeedleHipRun *__restrict__ sorted, const uint32_t *__restrict__ row_len,
                                                                 const uint32_t *__restrict__ row_ts, const uint64_t *__restrict__ row_seek,
                                                                 const uint64_t *__restrict__ ts, DeviceEntry *__restrict__ entries,
                                                                 uint32_t *__restrict__ valid, const uint32_t *__restrict__ large_count,
                                                                 const uint32_t *__restrict__ large_list) {
  extern __shared__ unsigned long long arr[];  // kEpilogueLargeLimit elements
  __shared__ uint32_t heap_size;
  const uint32_t t = threadIdx.x;
  const uint32_t listed = *large_count;
  for (uint32_t item = blockIdx.x; item < listed; item += gridDim.x) {
    const uint32_t b = large_list[item];
    const uint32_t lo = start[b], n = start[b + 1] - lo;
    uint32_t p2 = 1;
    while (p2 < n) p2 <<= 1;
    for (uint32_t a = t; a < p2; a += 256) {
      unsigned long long v = ~0ull;
      if (a < n) {
        const NeedleHipRun r = sorted[lo + a];
        v = ((unsigned long long)(((0xFFFFu - (r.src_end & 0xFFFFu)) << 16) | (0xFFFFu - (r.dst_end & 0xFFFFu))) << 32) | a;
      }
      arr[a] = v;
    }
    __syncthreads();
    for (uint32_t k = 2; k <= p2; k <<= 1)
      for (uint32_t j = k >> 1; j > 0; j >>= 1) {
        for (uint32_t a = t; a < p2; a += 256) {
          const uint32_t partner = a ^ j;
          if (partner > a) {
            const unsigned long long x = arr[a], y = arr[partner];
            const bool up = (a & k) == 0;
            if ((x > y) == up) {
              arr[a] = y;
              arr[partner] = x;
            }
          }
        }
        __syncthreads();
      }
    const uint32_t region = b % pr.regions;
    uint32_t vi, vj;
    pair_at_device(pr.n, b / pr.regions, &vi, &vj);
    const uint32_t src_row = vi * pr.rows_per_video + region, dst_row = vj * pr.rows_per_video + region;
    const uint32_t src_len = row_len[src_row], dst_len = row_len[dst_row];
    const uint64_t *src_ts = ts + row_ts[src_row], *dst_ts = ts + row_ts[dst_row];
    const uint64_t src_seek = row_seek[src_row], dst_seek = row_seek[dst_row];
    const uint64_t min_duration = pr.min_duration[region];
    auto entry_of = [&](const NeedleHipRun &r, DeviceEntry *e) {  // false: the reference skips the run (:212-223)
      const uint32_t i = r.src_end, j = r.dst_end, len = r.len;
      if (len == 0 || len > i || len > j || i >= src_len || j >= dst_len) return false;
      e->src_start = src_ts[i - len] + src_seek;
      e->src_end = src_ts[i] + src_seek;
      e->dst_start = dst_ts[j - len] + dst_seek;
      e->dst_end = dst_ts[j] + dst_seek;
      if (e->src_end < e->src_start || e->dst_end < e->dst_start) return false;
      if (e->src_end - e->src_start < min_duration || e->dst_end - e->dst_start < min_duration) return false;
      e->score = len;
      e->src_hash = r.src_match_hash;
      e->dst_hash = r.dst_match_hash;
      e->pad = 0;
      return true;
    };
    for (uint32_t a = t; a < n; a += 256) {
      const uint32_t idx = (uint32_t)arr[a];
      const NeedleHipRun r = sorted[lo + idx];
      DeviceEntry e;
      const bool ok = entry_of(r, &e);
      const unsigned long long rank = ((unsigned long long)r.len << 32) | ((unsigned long long)((r.src_end - r.len) & 0xFFFFu) << 16) | (r.dst_end & 0xFFFFu);
      arr[a] = (rank << 16) | (ok ? 0x8000ull : 0ull) | idx;
    }
    __syncthreads();
    if (t == 0) {
      uint32_t out = 0;
      for (uint32_t a = 0; a < n; a++) {
        const unsigned long long e = arr[a];
        if (!(e & 0x8000ull)) continue;
        uint32_t pos = out++;
        while (pos > 0) {  // BinaryHeap::push: append, sift up while greater than the parent
          const uint32_t parent = (pos - 1) / 2;
          const unsigned long long p = arr[parent];
          if (!((e >> 16) > (p >> 16))) break;
          arr[pos] = p;
          pos = parent;
        }
        arr[pos] = e;
      }
      heap_size = out;
      valid[b] = out;
    }
    __syncthreads();
    const uint32_t out = heap_size;
    for (uint32_t pos = t; pos < out; pos += 256) {
      const NeedleHipRun r = sorted[lo + (uint32_t)(arr[pos] & 0x1FFFull)];
      DeviceEntry e;
      (void)entry_of(r, &e);
      entries[lo + pos] = e;
    }
    __syncthreads();
  }
}

struct BestKey {
  float score;
  uint32_t index;
  uint32_t have;
};
__device__ __forceinline__ bool better(const BestKey &a, const BestKey &b) {  // a before b in the ascending (score, index) order
  if (!a.have) return false;
  if (!b.have) return true;
  return a.score < b.score || (a.score == b.score && a.index < b.index);
}

__device__ __forceinline__ float as_secs_f32(uint64_t d) {  // Duration::as_secs_f32 (hostutil.cpp duration_as_secs_f32)
  const float secs = (float)(d / 1000000000ull);
  const float frac = (float)(uint32_t)(d % 1000000000ull) / 1000000000.0f;
  return secs + frac;
}

constexpr int kImageRows = 512;   // candidates of a stage of the links' b side (best_match_kernel)
constexpr int kImagePitch = 12;   // words per row of the stage's image: 8 of +-1 bytes + 4 (16-byte reads of sixteen rows: sixteen groups of banks)
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
// One workgroup per wanted video.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void best_match_kernel(EpilogueParams pr, const uint32_t *__restrict__ start,
                                                         const uint32_t *__restrict__ valid, const DeviceEntry *__restrict__ entries,
                                                         Candidate *__restrict__ cand_pool, unsigned long long *__restrict__ cand_cursor,
                                                         uint32_t *__restrict__ links_pool, NeedleHipSearchResult *__restrict__ results,
                                                         uint32_t *__restrict__ failed) {
  __shared__ uint32_t scan[256];
  __shared__ uint32_t carry;
  __shared__ unsigned long long pool_base;
  __shared__ __attribute__((aligned(16))) uint32_t image[kImageRows * kImagePitch];
  __shared__ uint32_t ntab[16];
  __shared__ uint32_t dlinks[2048];
  __shared__ uint32_t distinct;
  __shared__ uint32_t slot_cnt[256], slot_at[256];
  __shared__ uint16_t occupied[1536 + 256];
  __shared__ BestKey best[2][256];
  // a bucket too large for one lane (pair_entries_kernel): the whole job is the host form's, nothing here would be read
  if (__builtin_nontemporal_load(failed) & kEpilogueBucketTooLarge) return;
  const uint32_t v = pr.v0 + blockIdx.x, t = threadIdx.x;
  const uint64_t n = pr.n;
  const uint32_t slots = pr.n - 1;  // the video's pairs in lexicographic order: (q, v) for q < v, then (v, q + 1)
  auto bucket_of = [&](uint32_t q) -> uint64_t {
    const uint64_t p = q < v ? row_start(n, q) + (v - q - 1) : row_start(n, v) + (q - v);
    return p * pr.regions;
  };
  // pass 1: candidates per pair slot -> total
  if (t == 0) carry = 0;
  __syncthreads();
  uint32_t mine_total = 0;
  for (uint32_t q = t; q < slots; q += 256) {
    const uint64_t b = bucket_of(q);
    mine_total += valid[b] + (pr.regions == 2 ? valid[b + 1] : 0u);
  }
  scan[t] = mine_total;
  __syncthreads();
  for (int d = 128; d > 0; d >>= 1) {
    if ((int)t < d) scan[t] += scan[t + d];
    __syncthreads();
  }
  const uint32_t c = scan[0];
  __syncthreads();
  NeedleHipSearchResult res;
  memset(&res, 0, sizeof(res));
  if (c == 0) {  // no pair of this video has an entry: the reference pushes nothing for it (:608-617)
    if (t == 0) results[v] = res;
    return;
  }
  if (t == 0) pool_base = atomicAdd(cand_cursor, (unsigned long long)c);
  __syncthreads();
  Candidate *cand = cand_pool + pool_base;
  uint32_t *links = links_pool + pool_base;
  // pass 2: candidate index of every pair slot (exclusive scan in slot order, 256 slots at a time), then the fill
  for (uint32_t base = 0; base < slots; base += 256) {
    const uint32_t q = base + t;
    uint32_t cnt = 0;
    uint64_t b = 0;
    if (q < slots) {
      b = bucket_of(q);
      cnt = valid[b] + (pr.regions == 2 ? valid[b + 1] : 0u);
    }
    scan[t] = cnt;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
      const uint32_t add = (int)t >= d ? scan[t - d] : 0u;
      __syncthreads();
      scan[t] += add;
      __syncthreads();
    }
    uint32_t at = carry + scan[t] - cnt;
    // a slot's entries -> candidates: by its own thread while they are few; a slot with many (a pair of silent stretches: hundreds)
    // by the whole workgroup -- one thread copying 400 entries while 255 wait was most of this kernel on the hostile corpus
    auto copy_entries = [&](const uint64_t bb, const bool as_source, uint32_t first, const uint32_t k0, const uint32_t kstep) {
      for (uint32_t r = 0; r < pr.regions; r++) {  // openings first, then endings (:414-431)
        const DeviceEntry *e = entries + start[bb + r];
        const uint32_t k1 = valid[bb + r];
        for (uint32_t k = k0; k < k1; k += kstep) {
          Candidate cd;
          cd.start = as_source ? e[k].src_start : e[k].dst_start;
          cd.end = as_source ? e[k].src_end : e[k].dst_end;
          cd.hash = as_source ? e[k].src_hash : e[k].dst_hash;
          cd.is_opening = r == 0 ? 1u : 0u;
          cand[first + k] = cd;
        }
        first += k1;
      }
    };
    constexpr uint32_t kOwnCopy = 16;
    slot_cnt[t] = cnt;
    slot_at[t] = at;
    if (cnt && cnt <= kOwnCopy) copy_entries(b, q >= v, at, 0u, 1u);
    __syncthreads();
    for (uint32_t qq = 0; qq < 256; qq++) {
      if (slot_cnt[qq] <= kOwnCopy) continue;  // (uniform)
      const uint32_t q2 = base + qq;
      copy_entries(bucket_of(q2), q2 >= v, slot_at[qq], t, 256u);
    }
    __syncthreads();
    if (t == 255) carry += scan[255];
    __syncthreads();
  }
  __threadfence_block();
  __syncthreads();
  // links[k] = #{b : popcount(h_k ^ h_b) < bound}, k itself included (:434-454): all candidates against all -- c x c Hamming
  // distances, 25 million per video at 2000 videos, and as in the scan a matrix product: with a hash as 32 bytes of +-1,
  // dot(a, b) = 32 - 2 d(a, b).  One v_mfma_i32_32x32x32_i8 is 32 candidates b (rows, the A side NEGATED) x 32 candidates k
  // (columns): with the accumulator preset to 32 - 2 bound its sign bit says d < bound.  A lane holds sixteen rows of ITS
  // column: the sixteen sign bits are shifted into a word, two tiles' words counted with one v_bcnt -- 18 vector
  // instructions per 1024 pairs (on the vector ALU, one lane per k: xor, popcount, compare, add = 64).  The b side is
  // staged through LDS as +-1 bytes, kImageRows candidates at a time, built by the workgroup and read by its four waves
  // as A fragments; a wave owns every fourth block of 32 k and keeps their sums in links[] between the stages.
  // (Rows beyond c are zero bytes: dot 0, "d = 16" -- counted as a match when bound > 16 and taken out again below.)
  // Round 6: a video with thousands of candidates has them from stretches of ONE repeated hash (silence, a sustained chord: every
  // diagonal of an S x S block is a run, and the simhash of a constant stretch is that constant) -- 33 000 candidates per video on
  // the hostile corpus at 280 files, a handful of DISTINCT hashes among them.  Candidates of equal hash have equal link counts:
  // links = sum over the distinct hashes within the bound of their multiplicities.  A 2048-slot table in LDS (the image's bytes,
  // unused on this path) takes the hashes by 64-bit compare-and-swap; beyond 1536 distinct values the all-pairs products below run.
  bool deduped = false;
  if (c >= 512) {
    constexpr uint32_t kSlots = 2048, kMaxDistinct = 1536;
    unsigned long long *keys = reinterpret_cast<unsigned long long *>(image);   // 1 << 32 | hash, 0 = empty
    uint32_t *mult = image + 2 * kSlots;
    for (uint32_t i = t; i < kSlots; i += 256) {
      keys[i] = 0ull;
      mult[i] = 0u;
      dlinks[i] = 0u;
    }
    if (t == 0) distinct = 0;
    __syncthreads();
    auto slot_of = [](uint32_t hash) { return (hash * 0x9E3779B1u) >> 21; };
    for (uint32_t k = t; k < c; k += 256) {
      const uint32_t hash = cand[k].hash;
      const unsigned long long want = (1ull << 32) | hash;
      uint32_t sl = slot_of(hash);
      for (uint32_t probe = 0; probe < kSlots; probe++, sl = (sl + 1) & (kSlots - 1)) {
        if (*reinterpret_cast<volatile uint32_t *>(&distinct) > kMaxDistinct) break;  // (overflowing: the direct path will run)
        unsigned long long old = *reinterpret_cast<volatile unsigned long long *>(&keys[sl]);  // (mostly there already: no atomic)
        if (old == 0ull) {
          old = atomicCAS(&keys[sl], 0ull, want);
          if (old == 0ull) atomicAdd(&distinct, 1u);
        }
        if (old == 0ull || old == want) {
          atomicAdd(&mult[sl], 1u);
          break;
        }
      }
    }
    __syncthreads();
    deduped = distinct <= kMaxDistinct;
    if (deduped) {
      if (t == 0) distinct = 0;  // now: the occupied slots, listed
      __syncthreads();
      for (uint32_t a = t; a < kSlots; a += 256)
        if (keys[a] != 0ull) occupied[atomicAdd(&distinct, 1u)] = (uint16_t)a;
      __syncthreads();
      const uint32_t u = distinct;
      for (uint32_t ia = t; ia < u; ia += 256) {
        const uint32_t a = occupied[ia];
        const uint32_t ha = (uint32_t)keys[a];
        uint32_t sum = 0;
        for (uint32_t ib = 0; ib < u; ib++) {
          const uint32_t b2 = occupied[ib];
          if ((uint32_t)__popc(ha ^ (uint32_t)keys[b2]) < pr.bound) sum += mult[b2];
        }
        dlinks[a] = sum;
      }
      __syncthreads();
      for (uint32_t k = t; k < c; k += 256) {
        const uint32_t hash = cand[k].hash;
        const unsigned long long want = (1ull << 32) | hash;
        uint32_t sl = slot_of(hash);
        while (keys[sl] != want) sl = (sl + 1) & (kSlots - 1);                 // present by construction
        links[k] = dlinks[sl];
      }
    }
    __syncthreads();
  }
  if (!deduped) {
    const uint32_t lane = t & 63, wave = t >> 6, r = lane & 31, h = lane >> 5;
    if (t < 16) {
      uint32_t w = 0;
      for (int i = 0; i < 4; i++) w |= (((t >> i) & 1) ? 0x01u : 0xFFu) << (8 * i);
      ntab[t] = w;
    }
    const int preset = 32 - 2 * (int)pr.bound;
    v16i presets;
#pragma unroll
    for (int q = 0; q < 16; q++) presets[q] = preset;
    const uint32_t k_blocks = (c + 31) / 32;
    for (uint32_t b0 = 0; b0 < c; b0 += kImageRows) {
      const uint32_t len = min((uint32_t)kImageRows, c - b0);
      const uint32_t b_blocks = (len + 31) / 32;
      __syncthreads();
      // the stage's image: row i = candidate b0 + i as 32 NEGATED +-1 bytes (a set bit: -1), 8 words at a pitch of 12
      for (uint32_t i = t; i < b_blocks * 32; i += 256) {
        const uint32_t hb = i < len ? ~cand[b0 + i].hash : 0u;
#pragma unroll
        for (int q = 0; q < 8; q++) image[i * kImagePitch + q] = i < len ? ntab[(hb >> (4 * q)) & 0xFu] : 0u;
      }
      __syncthreads();
      const uint32_t pad = b_blocks * 32 - len;   // zero rows of the stage's last block
      for (uint32_t kb = wave; kb < k_blocks; kb += 4) {
        const uint32_t k = kb * 32 + r;
        const uint32_t half = (k < c ? cand[k].hash : 0u) >> (16 * h);
        v4i fb;                                   // B fragment: column r = candidate k, bits 16 h .. 16 h + 15 as +-1 bytes
#pragma unroll
        for (int q = 0; q < 4; q++) fb[q] = (int)ntab[(half >> (4 * q)) & 0xFu];
        uint32_t cnt = 0, word = 0;
        for (uint32_t bb = 0; bb < b_blocks; bb++) {
          const v4i fa = *reinterpret_cast<const v4i *>(image + (bb * 32 + r) * kImagePitch + 4 * h);
          const v16i acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa, fb, presets, 0, 0, 0);
#pragma unroll
          for (int q = 0; q < 16; q++) word = __builtin_amdgcn_alignbit(word, (uint32_t)acc[q], 31);
          if (bb & 1) {
            cnt += (uint32_t)__popc(word);
            word = 0;
          }
        }
        cnt += (uint32_t)__popc(word);
        cnt += (uint32_t)__shfl_xor((int)cnt, 32);   // the column's other sixteen rows of every tile
        if (preset < 0) cnt -= pad;               // bound > 16: the zero rows counted
        if (h == 0 && k < c) links[k] = (b0 == 0 ? 0u : links[k]) + cnt;
      }
    }
  }
  __syncthreads();
  // score = -(count * 0.3 + secs * 0.7) in f32, no fused multiply-add (:469); ascending (score, index), first (:473-475)
  BestKey mine[2] = {{0.f, 0u, 0u}, {0.f, 0u, 0u}};
  for (uint32_t k = t; k < c; k += 256) {
    const Candidate cd = cand[k];
    const uint32_t l = links[k];
    if (l == 0) continue;
    const float count = (float)(long long)l;
    const float secs = as_secs_f32(cd.end - cd.start);
    const float a = count * 0.3f;
    const float b = secs * 0.7f;
    const float weighted = a + b;
    const BestKey key = {-weighted, k, 1u};
    const int which = cd.is_opening ? 0 : 1;
    if (better(key, mine[which])) mine[which] = key;
  }
  best[0][t] = mine[0];
  best[1][t] = mine[1];
  __syncthreads();
  for (int d = 128; d > 0; d >>= 1) {
    if ((int)t < d) {
      if (better(best[0][t + d], best[0][t])) best[0][t] = best[0][t + d];
      if (better(best[1][t + d], best[1][t])) best[1][t] = best[1][t + d];
    }
    __syncthreads();
  }
  if (t == 0) {
    res.has_result = true;  // Some(best) even if neither side is found (:514)
    bool bad = false;
    for (int which = 0; which < 2; which++) {
      if (which == 1 && !pr.include_endings) break;  // :486
      const BestKey w = best[which][0];
      if (!w.have) continue;
      const Candidate cd = cand[w.index];
      if (cd.end < pr.time_padding || cd.end - pr.time_padding < pr.hash_duration) {  // Duration underflow panics upstream
        bad = true;
        break;
      }
      const uint64_t s = cd.start + pr.time_padding;                    // :479
      const uint64_t e = cd.end - pr.time_padding - pr.hash_duration;   // :481
      if (which == 0) {
        res.has_opening = true;
        res.opening_start_ns = s;
        res.opening_end_ns = e;
      } else {
        res.has_ending = true;
        res.ending_start_ns = s;
        res.ending_end_ns = e;
      }
    }
    if (bad) atomicAdd(failed, 1u);
    results[v] = res;
  }
}

struct EpilogueWorkspace {
  DeviceBuffer<uint32_t> count, start, fill, sums, valid, links, ctl, row_len, row_ts, large_list;
  bool large_checked = false;
  bool large_ok = false, large_attr_set = false;  // of the resident row tables: every row under 65 536 hashes, timestamps strictly increasing
  DeviceBuffer<uint64_t> row_seek, ts;
  DeviceBuffer<NeedleHipRun> sorted;
  DeviceBuffer<DeviceEntry> entries;
  DeviceBuffer<Candidate> cand;
  DeviceBuffer<NeedleHipSearchResult> results;
  std::vector<uint32_t> h_row_len, h_row_ts;  // what the device tables hold
  std::vector<uint64_t> h_row_seek, h_ts;
};
std::mutex g_mu;
std::map<std::pair<int, int>, EpilogueWorkspace *> g_ws;  // (device, job slot)

EpilogueWorkspace *workspace(int slot) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lock(g_mu);
  EpilogueWorkspace *&w = g_ws[{dev, slot}];
  if (!w) w = new EpilogueWorkspace();
  return w;
}

template <class T>
Status upload_if_changed(DeviceBuffer<T> *dst, std::vector<T> *resident, const std::vector<T> &want, hipStream_t stream) {
  if (dst->ptr && *resident == want) return Status::Ok();
  Status s = dst->reserve(std::max<size_t>(want.size(), 1));
  if (!s.ok()) return s;
  // (pageable source: the copy is staged by the runtime before the call returns)
  if (!want.empty()) NEEDLE_HIP_TRY(hipMemcpyAsync(dst->ptr, want.data(), want.size() * sizeof(T), hipMemcpyHostToDevice, stream));
  NEEDLE_HIP_TRY(hipStreamSynchronize(stream));
  *resident = want;
  return Status::Ok();
}

}  // namespace

std::atomic<uint64_t> &epilogue_host_fallbacks() {
  static std::atomic<uint64_t> count{0};
  return count;
}

void note_epilogue_host_fallback(const char *where, size_t runs, size_t videos) {
  epilogue_host_fallbacks().fetch_add(1);
  if (getenv("NEEDLE_HIP_TRACE"))
    std::fprintf(stderr, "[needle_hip] %s: a pair's bucket holds more than %u runs (silence / a sustained tone on both sides) or the rows are too long for the packed keys: the "
                         "per-video epilogue of this job (%zu runs, %zu videos) falls back to the HOST form\n",
                 where, kEpilogueLargeLimit, runs, videos);
}

Status gpu_direct_runs(const uint32_t *d_found, const NeedleHipRun *d_runs, uint32_t slab_capacity, uint32_t n, uint32_t regions,
                       uint32_t videos_per_rank, int world, uint8_t *d_send, const DirectPlan &plan, hipStream_t stream) {
  if (world < 1 || world > 64 || n < 2 || regions < 1 || videos_per_rank < 1)
    return Status::Make(NeedleError_InvalidArgument, "directed run exchange: invalid plan");
  for (int q = 0; q < world; q++) NEEDLE_HIP_TRY(hipMemsetAsync(d_send + plan.offset[q], 0, 32, stream));
  const uint32_t grid = (uint32_t)std::min<uint64_t>(2048, ((uint64_t)slab_capacity + 255) / 256);
  hipLaunchKernelGGL(direct_runs_kernel, dim3(std::max(grid, 1u)), dim3(256), 0, stream, d_found, d_runs, slab_capacity, n, regions,
                     videos_per_rank, world, d_send, plan);
  NEEDLE_HIP_TRY(hipGetLastError());
  return Status::Ok();
}

Status gpu_epilogue_enqueue(const EpilogueJob &job, hipStream_t stream, NeedleHipSearchResult *host_results, uint32_t *host_failed) {
  std::lock_guard<std::recursive_mutex> gpu_lock(gpu_mutex());  // the per-device workspaces are shared, as everywhere else
  if (job.num_segments < 1 || job.num_segments > kMaxSegments)
    return Status::Make(NeedleError_InvalidArgument, "device epilogue: too many run segments");
  const uint64_t np = (uint64_t)job.n * (job.n - 1) / 2;
  const uint64_t buckets = np * job.regions;
  if (job.n < 2 || buckets >= 0xFFFFFFF0ull || job.max_runs >= 0xFFFFFFF0ull)
    return Status::Make(NeedleError_InvalidArgument, "device epilogue: library too large");
  EpilogueWorkspace *ws = workspace(job.slot);
  Status s;
  // what pair_entries_large_kernel's packed keys stand on, recomputed when the tables change (before they are taken over below):
  // every row under 65 536 hashes, every timestamp table strictly increasing (tables are shared by rows of equal length)
  if (!ws->large_checked || ws->h_ts != *job.ts || ws->h_row_len != *job.row_len || ws->h_row_ts != *job.row_ts) {
    static_assert(kEpilogueLargeLimit <= 8192, "13 bits of index in the packed key");
    bool ok = true;
    std::map<uint32_t, uint32_t> longest;  // table offset -> the longest row that reads it
    for (size_t r = 0; r < job.row_len->size() && ok; r++) {
      const uint32_t len = (*job.row_len)[r];
      ok = len < 65536u;
      uint32_t &m = longest[(*job.row_ts)[r]];
      m = std::max(m, len);
    }
    for (const auto &kv : longest) {
      const uint64_t *t = job.ts->data() + kv.first;
      for (uint32_t k = 1; k < kv.second && ok; k++) ok = t[k] > t[k - 1];
    }
    ws->large_ok = ok;
    ws->large_checked = true;
  }
  // row tables: only re-uploaded when the geometry changes
  if (!(s = upload_if_changed(&ws->row_len, &ws->h_row_len, *job.row_len, stream)).ok() ||
      !(s = upload_if_changed(&ws->row_ts, &ws->h_row_ts, *job.row_ts, stream)).ok() ||
      !(s = upload_if_changed(&ws->row_seek, &ws->h_row_seek, *job.row_seek, stream)).ok() ||
      !(s = upload_if_changed(&ws->ts, &ws->h_ts, *job.ts, stream)).ok())
    return s;
  const size_t runs = std::max<uint64_t>(job.max_runs, 1);
  if (!(s = ws->count.reserve(buckets)).ok() || !(s = ws->fill.reserve(buckets)).ok() || !(s = ws->start.reserve(buckets + 1)).ok() ||
      !(s = ws->valid.reserve(buckets)).ok() || !(s = ws->sums.reserve((buckets + kScanBlock - 1) / kScanBlock + 1)).ok() ||
      !(s = ws->sorted.reserve(runs)).ok() || !(s = ws->entries.reserve(runs)).ok() || !(s = ws->cand.reserve(2 * runs)).ok() ||
      !(s = ws->links.reserve(2 * runs)).ok() || !(s = ws->ctl.reserve(4)).ok() || !(s = ws->results.reserve(job.n)).ok() ||
      !(s = ws->large_list.reserve(runs / (kEpilogueBucketLimit + 1) + 1)).ok())
    return s;
  if (!ws->large_attr_set) {
    NEEDLE_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(pair_entries_large_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)(kEpilogueLargeLimit * sizeof(unsigned long long))));
    ws->large_attr_set = true;
  }
  RunSegments segs;
  std::memset(&segs, 0, sizeof(segs));
  segs.count = job.num_segments;
  for (int k = 0; k < job.num_segments; k++) {
    segs.found[k] = job.segment_count[k];
    segs.runs[k] = job.segment_runs[k];
    segs.capacity[k] = job.segment_capacities[k] ? job.segment_capacities[k] : job.segment_capacity;
  }
  EpilogueParams pr;
  std::memset(&pr, 0, sizeof(pr));
  pr.n = job.n;
  pr.regions = job.regions;
  pr.rows_per_video = job.rows_per_video;
  pr.buckets = (uint32_t)buckets;
  pr.v0 = job.v0;
  pr.v1 = job.v1;
  pr.bound = job.threshold + job.threshold / 2;
  pr.include_endings = job.include_endings ? 1u : 0u;
  pr.min_duration[0] = job.min_opening_duration;
  pr.min_duration[1] = job.min_ending_duration;
  pr.time_padding = job.time_padding;
  pr.hash_duration = job.hash_duration;
  pr.large_ok = ws->large_ok && getenv("NEEDLE_HIP_EPILOGUE_NO_LARGE") == nullptr ? 1u : 0u;  // (tests: the host fallback itself)
  NEEDLE_HIP_TRY(hipMemsetAsync(ws->count.ptr, 0, buckets * sizeof(uint32_t), stream));
  NEEDLE_HIP_TRY(hipMemsetAsync(ws->fill.ptr, 0, buckets * sizeof(uint32_t), stream));
  NEEDLE_HIP_TRY(hipMemsetAsync(ws->ctl.ptr, 0, 4 * sizeof(uint32_t), stream));
  NEEDLE_HIP_TRY(hipMemsetAsync(ws->results.ptr, 0, (size_t)job.n * sizeof(NeedleHipSearchResult), stream));
  const uint32_t run_grid = (uint32_t)std::min<uint64_t>(4096, (runs + 255) / 256);
  const uint32_t scan_blocks = (uint32_t)((buckets + kScanBlock - 1) / kScanBlock);
  {
    KernelTimer timer("epilogue_buckets", stream);
    hipLaunchKernelGGL(bucket_count_kernel, dim3(run_grid), dim3(256), 0, stream, segs, pr.buckets, ws->count.ptr);
    hipLaunchKernelGGL(scan_block_sums_kernel, dim3(scan_blocks), dim3(256), 0, stream, ws->count.ptr, pr.buckets, ws->sums.ptr);
    hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(256), 0, stream, ws->sums.ptr, scan_blocks);
    hipLaunchKernelGGL(scan_apply_kernel, dim3(scan_blocks), dim3(256), 0, stream, ws->count.ptr, pr.buckets, ws->sums.ptr, ws->start.ptr);
    hipLaunchKernelGGL(bucket_scatter_kernel, dim3(run_grid), dim3(256), 0, stream, segs, pr.buckets, ws->start.ptr, ws->fill.ptr,
                       ws->sorted.ptr);
  }
  {
    KernelTimer timer("epilogue_entries", stream);
    hipLaunchKernelGGL(pair_entries_kernel, dim3((pr.buckets + 63) / 64), dim3(64), 0, stream, pr, ws->start.ptr, ws->sorted.ptr,
                       ws->row_len.ptr, ws->row_ts.ptr, ws->row_seek.ptr, ws->ts.ptr, ws->entries.ptr, ws->valid.ptr, ws->ctl.ptr + 2,
                       ws->ctl.ptr + 3, ws->large_list.ptr);
    // the buckets one lane should not order (a stride loop over a list that is empty on ordinary audio: ~2 us then)
    hipLaunchKernelGGL(pair_entries_large_kernel, dim3(512), dim3(256), kEpilogueLargeLimit * sizeof(unsigned long long), stream, pr,
                       ws->start.ptr, ws->sorted.ptr, ws->row_len.ptr, ws->row_ts.ptr, ws->row_seek.ptr, ws->ts.ptr, ws->entries.ptr,
                       ws->valid.ptr, ws->ctl.ptr + 3, ws->large_list.ptr);
  }
  if (pr.v1 > pr.v0) {
    KernelTimer timer("epilogue_best_match", stream);
    hipLaunchKernelGGL(best_match_kernel, dim3(pr.v1 - pr.v0), dim3(256), 0, stream, pr, ws->start.ptr, ws->valid.ptr, ws->entries.ptr,
                       ws->cand.ptr, reinterpret_cast<unsigned long long *>(ws->ctl.ptr), ws->links.ptr, ws->results.ptr, ws->ctl.ptr + 2);
  }
  NEEDLE_HIP_TRY(hipGetLastError());
  NEEDLE_HIP_TRY(hipMemcpyAsync(host_results, ws->results.ptr, (size_t)job.n * sizeof(NeedleHipSearchResult), hipMemcpyDeviceToHost, stream));
  NEEDLE_HIP_TRY(hipMemcpyAsync(host_failed, ws->ctl.ptr + 2, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
  return Status::Ok();
}

}  // namespace needle
