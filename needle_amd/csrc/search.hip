// GPU replacement of Comparator::longest_common_hash_match's two table sweeps
// (needle/src/audio/comparator.rs:175-247).
//
// The reference fills an (n+1)x(m+1) table with t[i][j] = t[i-1][j-1]+1 when
// popcount(src[i]^dst[j]) <= threshold (rows/cols 0 forced to 0, :179-180) and then walks it backwards
// keeping cells that END a maximal diagonal run (:196-200).  Every diagonal is independent, so no table
// is ever built here: one lane walks one diagonal d = j - i over its valid cells (i >= 1, j >= 1),
// carrying the current run length in a register, and appends (i_end, j_end, L) for each maximal run
// with L >= min_len.  The duration test (:212-223) needs timestamps and stays on the host; min_len is
// the host's lower bound on the run length that can pass it, so the output list stays short.
//
// Integer-only (xor, popcount, compare, add): results are exact by construction.
#include "epilogue.h"
#include "hipctx.h"

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <unordered_map>

namespace needle {

namespace {

constexpr int kDiagsPerBlock = 256;
constexpr int kBandR = 7, kBandU = 3;          // band kernel: 7 diagonals per lane, checkpoint every 21 rows
constexpr int kBandB = 64 * kBandR;
constexpr int kSampleW = 8;                     // sampled kernel: rows per aligned window
constexpr int kSampleHead = 3;                  // rows of a window evaluated before the first early-out
// Which rows: the first, a middle and the last one of the window.  Neighbouring rows of a diagonal are correlated in
// audio (a note lasts several hashes): between unrelated synthetic episodes three CONSECUTIVE cells all match with
// probability 0.7 %, the cells of rows {0, W/2, W-1} with 0.15 % -- 4.5 times fewer diagonals to finish afterwards.
// H head rows of a window of W: H = 3 (the default) takes rows {0, W/2, W-1}; other counts are spread evenly over the
// window (H = 2: {0, W-1}; H = 4: {0, (W-1)/3, 2(W-1)/3, W-1}) -- tools/scan_shape_sweep.py measures the choices.
__host__ __device__ constexpr int head_row(int k, int W, int H = kSampleHead) {
  return H == 3 ? (k == 0 ? 0 : (k == 1 ? W / 2 : W - 1)) : (k * (W - 1)) / (H - 1);
}
__host__ __device__ constexpr bool is_head_row(int s, int W, int H = kSampleHead) {
  for (int k = 0; k < H; k++)
    if (head_row(k, W, H) == s) return true;
  return false;
}
// k-th row of the window that is NOT a head row (k = 0 .. W - H - 1)
__host__ __device__ constexpr int tail_row(int k, int W, int H = kSampleHead) {
  int seen = 0;
  for (int s = 0; s < W; s++) {
    if (is_head_row(s, W, H)) continue;
    if (seen == k) return s;
    seen++;
  }
  return W - 1;
}

struct SearchProblem {
  uint32_t src_off, n;  // hash arena offset + length of the source sequence
  uint32_t dst_off, m;
  uint32_t min_len, tag;
  uint32_t block_base;  // first workgroup of this problem in the grid
  uint32_t pad;
};

// One workgroup = 256 consecutive diagonals of one problem; both sequences staged in LDS.
// STAGED = false: the sequences are read where they lie in HBM (lanes of a wave read consecutive destination hashes,
// the source hash of a row is the same for all): the path for a pair too long for the LDS of the faster kernels
// (a window of more than ~2.7 h of audio), which must not fail the whole library's search.  `first_problem` = index
// of problems[0] in the launch's table (what the emitted run carries until simhash_runs_kernel swaps in the tag).
template <bool STAGED>
__global__ __launch_bounds__(256) void hamming_runs_kernel(const uint32_t *__restrict__ hashes,
                                                           const SearchProblem *__restrict__ problems,
                                                           int num_problems, uint32_t first_problem, uint32_t threshold,
                                                           NeedleHipRun *__restrict__ runs, uint32_t capacity,
                                                           uint32_t *__restrict__ count) {
  extern __shared__ uint32_t lds[];
  // problem lookup: last problem with block_base <= blockIdx.x
  int lo = 0, hi = num_problems - 1;
  while (lo < hi) {
    int mid = (lo + hi + 1) >> 1;
    if (problems[mid].block_base <= blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const SearchProblem pr = problems[lo];
  const int n = (int)pr.n, m = (int)pr.m;
  const uint32_t *s = hashes + pr.src_off, *t = hashes + pr.dst_off;
  if (STAGED) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) lds[i] = s[i];
    for (int j = threadIdx.x; j < m; j += blockDim.x) lds[n + j] = t[j];
    __syncthreads();
    s = lds;
    t = lds + n;
  }

  // diagonals d = j - i with at least one cell i>=1, j>=1: d in [-(n-2), m-2]
  const int dd = (int)(blockIdx.x - pr.block_base) * kDiagsPerBlock + (int)threadIdx.x;
  const int num_diags = n + m - 3;
  if (dd >= num_diags) return;
  const int d = dd - (n - 2);
  const int i_lo = d < 0 ? 1 - d : 1;
  const int i_hi = min(n - 1, m - 1 - d);
  const uint32_t min_len = pr.min_len;
  uint32_t run = 0;
  for (int i = i_lo; i <= i_hi; i++) {
    const bool match = (uint32_t)__popc(s[i] ^ t[i + d]) <= threshold;
    if (match) {
      run++;
    } else {
      if (run >= min_len) {  // run ended at the previous cell
        const uint32_t slot = atomicAdd(count, 1u);
        if (slot < capacity)
          runs[slot] = NeedleHipRun{first_problem + (uint32_t)lo, (uint32_t)(i - 1), (uint32_t)(i - 1 + d), run, 0u, 0u};
      }
      run = 0;
    }
  }
  if (run >= min_len) {  // run reaches the table edge (i == n-1 or j == m-1, comparator.rs:197)
    const uint32_t slot = atomicAdd(count, 1u);
    if (slot < capacity)
      runs[slot] = NeedleHipRun{first_problem + (uint32_t)lo, (uint32_t)i_hi, (uint32_t)(i_hi + d), run, 0u, 0u};
  }
}

// ---- fast path: register-window band scan ------------------------------------------------------------------
// One wave owns a band of B = 64*R consecutive diagonals, lane l the R diagonals d_l + r (d_l = D0 + l*R).
// At row i the lane holds the window W[r] = dst[i + d_l + r] in registers; moving to row i+1 the window
// slides by one element, so with the row loop unrolled R-fold the registers are simply renamed and ONE new
// element per lane per row comes from LDS (lane stride R dwords: conflict-free for odd R).  src[i] is
// wave-uniform and comes through the scalar cache.  Per cell that leaves xor, popcount, compare, and one
// select that keeps Z[r] = the last row at which diagonal r mismatched (run length = row - Z[r]).
//
// No per-cell bounds masking: the band is a parallelogram, so near the table edges some cells lie outside
// rows/cols [1, n-1] x [1, m-1]; they are computed on zero padding (or on row/col 0) and can only EXTEND a
// run backwards.  Exactness is restored where runs are reported: every K rows (a "checkpoint") a lane whose
// current run is >= min_len - K + 1 walks forward from the checkpoint with explicit bounds to find the true
// end of the run, clamps the start to the valid range, and emits it iff it is the last checkpoint inside
// the run (so each maximal run with L >= min_len >= K is emitted exactly once).
template <int R, int U>
__global__ __launch_bounds__(256) void hamming_runs_band_kernel(const uint32_t *__restrict__ hashes,
                                                                const SearchProblem *__restrict__ problems,
                                                                int num_problems, uint32_t threshold,
                                                                NeedleHipRun *__restrict__ runs, uint32_t capacity,
                                                                uint32_t *__restrict__ count) {
  constexpr int K = R * U;   // rows per checkpoint
  constexpr int B = 64 * R;  // diagonals per wave
  extern __shared__ uint32_t lds[];
  int lo = 0, hi = num_problems - 1;
  while (lo < hi) {
    int mid = (lo + hi + 1) >> 1;
    if (problems[mid].block_base <= blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const SearchProblem pr = problems[lo];
  const int n = (int)pr.n, m = (int)pr.m;
  const uint32_t *__restrict__ src = hashes + pr.src_off;
  const uint32_t *__restrict__ dst = hashes + pr.dst_off;
  // dst staged with B zero-padded slots on both sides: lds[B + j] = dst[j]
  for (int k = threadIdx.x; k < m + 2 * B; k += blockDim.x) {
    const int j = k - B;
    lds[k] = (j >= 0 && j < m) ? dst[j] : 0u;
  }
  __syncthreads();

  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = (int)(threadIdx.x & 63);
  const int band = (int)(blockIdx.x - pr.block_base) * 4 + wave;
  const int D0 = band * B - (n - 2);  // first diagonal of the band
  if (D0 > m - 2) return;
  const int i_start = max(1, 2 - D0 - B);   // first row on which some diagonal of the band has j >= 1
  const int i_end = min(n - 1, m - 1 - D0); // last row on which some diagonal of the band has j <= m-1
  if (i_start > i_end) return;
  const int d_l = D0 + lane * R;
  const int min_len = (int)pr.min_len;
  const int flag_len = min_len - K + 1;     // >= 1: the host only selects this kernel when min_len >= K

  uint32_t W[R];
  int Z[R];
#pragma unroll
  for (int r = 0; r < R; r++) {
    W[r] = lds[B + i_start + d_l + r];
    Z[r] = i_start - 1;
  }

  auto checkpoint = [&](int c) {
    // c = row just processed.  Lanes with a long enough current run resolve it exactly.
#pragma unroll
    for (int r = 0; r < R; r++) {
      if (c - Z[r] >= flag_len) {
        const int d = d_l + r;
        const int ilo = d < 0 ? 1 - d : 1;
        const int ihi = min(n - 1, m - 1 - d);
        if (c >= ilo && c <= ihi) {
          const int a = max(Z[r] + 1, ilo);  // first matched row of the real run
          const int limit = min(ihi, c + K);
          int e = c + 1;
          while (e <= limit && (uint32_t)__popc(src[e] ^ lds[B + e + d]) <= threshold) e++;
          const bool reaches_next_checkpoint = (e > limit) && (limit == c + K);
          if (!reaches_next_checkpoint) {
            const int b = e - 1;
            const int len = b - a + 1;
            if (len >= min_len) {
              const uint32_t slot = atomicAdd(count, 1u);
              if (slot < capacity) runs[slot] = NeedleHipRun{(uint32_t)lo, (uint32_t)b, (uint32_t)(b + d), (uint32_t)len, 0u, 0u};
            }
          }
        }
      }
    }
  };

  int i = i_start;
  for (; i + K - 1 <= i_end; i += K) {
    const uint32_t *__restrict__ nxt = lds + (B + i + d_l + R);  // element entering the window after row i
#pragma unroll
    for (int s = 0; s < K; s++) {
      const int row = i + s;
      const uint32_t sv = src[row];
#pragma unroll
      for (int r = 0; r < R; r++) {
        const uint32_t c = (uint32_t)__popc(sv ^ W[(r + s) % R]);
        Z[r] = (c <= threshold) ? Z[r] : row;
        if (r == 0) W[s % R] = nxt[s];  // slot of the consumed oldest element takes the newest
      }
    }
    bool any = false;
#pragma unroll
    for (int r = 0; r < R; r++) any |= (i + K - 1 - Z[r] >= flag_len);
    if (__any(any)) checkpoint(i + K - 1);
  }
  // remainder (< K rows), rolled: window kept in order W[0..R-1] again (K is a multiple of R)
  for (; i <= i_end; i++) {
    const uint32_t sv = src[i];
#pragma unroll
    for (int r = 0; r < R; r++) {
      const uint32_t c = (uint32_t)__popc(sv ^ W[r]);
      Z[r] = (c <= threshold) ? Z[r] : i;
    }
#pragma unroll
    for (int r = 0; r + 1 < R; r++) W[r] = W[r + 1];
    W[R - 1] = lds[B + i + d_l + R];
  }
}

// LDS written by a wave and read back by other lanes of the SAME wave: the stores have to be complete (and the compiler
// must not move the loads above them); no other wave is involved.
__device__ __forceinline__ void wave_lds_fence_search() {  // (stft_kernel.h wave_lds_fence: the hardware runs one wave's LDS operations in order)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- sampled path: aligned-window candidates + exact cooperative resolution ------------------------------------
// A maximal run of L >= min_len cells on a diagonal covers at least one ALIGNED window of W rows out of every
// P = min_len - W + 1 rows (windows at rows [1 + kP, 1 + kP + W)): any W + P - 1 = min_len consecutive rows
// contain one.  So only W of every P rows are evaluated — for each (diagonal, window) the AND of its W cell
// tests — and a diagonal/window whose W cells all match is a candidate.  Candidates are rare; each one is
// resolved exactly by the whole wave: 64 cells at a time are tested forwards and backwards from the window
// with explicit bounds, giving the true maximal run [a, b]; it is emitted by the LAST aligned window it
// covers, so exactly once, and only if b - a + 1 >= min_len.  Nothing is approximated: the output equals
// the reference's table walk (comparator.rs:191-247) for every run long enough to matter.
//
// Per evaluated cell: v_xor, v_bcnt, v_cmp (+ one scalar AND of lane masks); at the default 20 s minimum
// (min_len 82, W 8, P 75) that is 3 VALU on at most 10.7 % of the cells -- typically 4 % (the first three rows of
// each window, plus the few diagonals that survive them, finished one at a time) -- instead of 4 VALU on all of them.
// COUNT (diagnostic instantiation, NEEDLE_HIP_SCAN_COUNT=1; never the timed one): every wave also counts the
// xor / popcount / compare groups it ISSUES (one group = the three instructions over its 64 lanes, whether a lane's
// cell lies inside the table, on the padding or repeats another lane's) and adds the total to *eval_groups when it
// leaves -- the numerator of an honest roofline: issued cell evaluations per second against the measured
// integer-VALU ceiling, <= 1 by construction, unlike the cells of the reference's table the scan merely COVERS.
// LEGACY: with the two older ways of finishing the head rows' survivors compiled in (sparse_max >= 0: NEEDLE_HIP_SPARSE_MAX, tests and
// tuning).  The kernel every job runs has the default way only (round 6: a third fewer instructions; the scan runs beside the next
// job's first pass, and the size of this kernel shows in that job's time -- profiles/NOTES.md).
template <int R, int W, bool COUNT = false, int H = kSampleHead, bool LEGACY = false>
__global__ __launch_bounds__(256) void hamming_runs_sampled_kernel(const uint32_t *__restrict__ hashes,
                                                                   const SearchProblem *__restrict__ problems,
                                                                   int num_problems, uint32_t threshold,
                                                                   NeedleHipRun *__restrict__ runs,
                                                                   uint32_t capacity, uint32_t *__restrict__ count,
                                                                   int bands_per_wave, int sparse_max,
                                                                   unsigned long long *__restrict__ eval_groups) {
  constexpr int B = 64 * R;
  unsigned long long groups = 0, survived = 0;  // COUNT only
  extern __shared__ uint32_t lds[];
  __builtin_amdgcn_s_setprio(3);  // a tail kernel beside the next job's first pass: its waves issue first (fingerprint.hip, shared-CU overlap)
  int lo = 0, hi = num_problems - 1;
  while (lo < hi) {
    int mid = (lo + hi + 1) >> 1;
    if (problems[mid].block_base <= blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const SearchProblem pr = problems[lo];
  const int n = (int)pr.n, m = (int)pr.m;
  const uint32_t *__restrict__ src = hashes + pr.src_off;
  const uint32_t *__restrict__ dst = hashes + pr.dst_off;
  // LDS: dst with B zero slots on both sides (ldst[B + j] = dst[j]).  src is read through the scalar cache in the
  // window loop and from global memory in the (rare) exact resolution of a candidate, so it is not staged.
  uint32_t *ldst = lds;
  for (int k = threadIdx.x; k < B; k += blockDim.x) {
    ldst[k] = 0u;
    ldst[B + m + k] = 0u;
  }
  {  // body: eight loads in flight per thread, then a plain tail
    constexpr int kU = 8;
    const int nt = (int)blockDim.x;
    int k = threadIdx.x;
    for (; k + (kU - 1) * nt < m; k += kU * nt) {
      uint32_t v[kU];
#pragma unroll
      for (int u = 0; u < kU; u++) v[u] = dst[k + u * nt];
#pragma unroll
      for (int u = 0; u < kU; u++) ldst[B + k + u * nt] = v[u];
    }
    for (; k < m; k += nt) ldst[B + k] = dst[k];
  }
  const uint32_t *__restrict__ lsrc = src;
  __syncthreads();

  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = (int)(threadIdx.x & 63);
  const int min_len = (int)pr.min_len;
  const int P = min_len - W + 1;  // >= W: the host selects this kernel only when min_len >= 2W - 1
  const uint32_t bias = 31u - min(threshold, 31u);  // (the host sends thresholds of 32 and more to the generic kernel)
  // a workgroup stages the destination once and its four waves walk bands_per_wave bands each (large launches:
  // one workgroup per pair instead of one per four bands, so the staging is not repeated)
  // Bands near the corners of the table have few rows, those around the main diagonal all of them.  A workgroup
  // therefore takes bands STRIDED across the table (slot q of workgroup b: band q nb + b, nb = workgroups of this
  // pair), so that every workgroup carries the same mix of short and long bands -- with consecutive bands per
  // workgroup the heavy middle workgroups of every pair land on the same CUs of a round-robin dispatch -- and the
  // slot a wave starts with rotates with the pair, so that no SIMD always gets the long ones.
  // A wave's runs wait in REGISTERS -- run k in lane k of three registers (the values are wave-uniform) -- and go
  // to the run list 64 at a time, or when the wave leaves, with ONE request for room.  (Round 6: one returning atomic per run
  // on the list's single counter is ~11 ns, device-wide and one after the other; stretches of one repeated hash -- silence
  // against silence: every diagonal of an S x S block is a run -- gave the 378 pairs of the hostile corpus 41 528 runs and
  // the launch 0.46 ms of waiting for that counter.)
  uint32_t held_b = 0u, held_e = 0u, held_len = 0u;
  int held = 0;  // wave-uniform
  auto flush_runs = [&]() {
    if (held == 0) return;
    uint32_t base = 0u;
    if (lane == 0) base = atomicAdd(count, (uint32_t)held);
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    if (lane < held && base + (uint32_t)lane < capacity)
      runs[base + (uint32_t)lane] = NeedleHipRun{(uint32_t)lo, held_b, held_e, held_len, 0u, 0u};
    held = 0;
  };
  auto emit_run = [&](const uint32_t b, const uint32_t e, const uint32_t len) {  // wave-uniform arguments
    const bool mine = lane == held;               // (a select per field: cheaper to write than v_writelane_b32 through inline asm)
    held_b = mine ? b : held_b;
    held_e = mine ? e : held_e;
    held_len = mine ? len : held_len;
    if (++held == 64) flush_runs();
  };
  const int slots = 4 * bands_per_wave;
  const int total_bands = (n + m - 3 + B - 1) / B;
  const int nb = (total_bands + slots - 1) / slots;
  const int b_in_pair = (int)(blockIdx.x - pr.block_base);
  for (int round = 0; round < bands_per_wave; round++) {
  const int q = (round * 4 + wave + lo) % slots;
  const int band = q * nb + b_in_pair;
  const int D0 = band * B - (n - 2);
  if (band >= total_bands || D0 > m - 2) continue;
  const int i_start = max(1, 2 - D0 - B);
  const int i_end = min(n - 1, m - 1 - D0);
  const int d_l = D0 + lane * R;

  // first aligned window (rows 1 + kP ...) that starts at or after i_start
  int k0 = (i_start - 1 + P - 1) / P;
  for (int w0 = 1 + k0 * P; w0 + W - 1 <= i_end; w0 += P) {
    // the W + R - 1 destination hashes this lane's R diagonals meet in rows w0 .. w0+W-1; the rows are taken in
    // two parts: after H of them (head_row) hardly any of the wave's 64 R diagonals still matches on
    // unrelated audio, and then the rest of the window is skipped -- nothing can pass that has already failed
    uint32_t E[W + R - 1];
#pragma unroll
    for (int q = 0; q < W + R - 1; q++) E[q] = ldst[B + w0 + d_l + q];
    // the window's W source hashes: wave-uniform, fetched together through the scalar cache (one latency for the
    // head rows, the sparse finish and the remaining rows alike)
    uint32_t sv[W];
#pragma unroll
    for (int s = 0; s < W; s++) sv[s] = src[w0 + s];
    // A cell matches iff popcount <= threshold.  One compare per cell (v_cmp to an SGPR mask + a scalar AND) is the most
    // expensive of its three instructions on gfx950 (profiles/r04_issue_rates2.log: v_xor 4.6, v_bcnt 8.7, v_cmp 9.1
    // cycles of a wave), so the verdicts of a diagonal's rows are gathered in a register instead: v_bcnt adds its second
    // operand, popcount + (31 - threshold) has bit 5 set exactly when the cell does NOT match (the sum stays below 64),
    // the rows' sums are ORed together, and ONE compare per diagonal and window reads the bit.  Same cells, same verdicts.
    bool ok[R];
    uint32_t miss[R];  // bit 5: some row of the window so far does not match on this diagonal
#pragma unroll
    for (int r = 0; r < R; r++) miss[r] = 0u;
#pragma unroll
    for (int k = 0; k < H; k++) {
      const int s = head_row(k, W, H);
#pragma unroll
      for (int r = 0; r < R; r++) miss[r] |= (uint32_t)__popc(sv[s] ^ E[s + r]) + bias;
    }
#pragma unroll
    for (int r = 0; r < R; r++) ok[r] = miss[r] < 32u;
    if (COUNT) groups += H * R;
    // exact resolution of one diagonal whose W window cells all match, by the whole wave (d is wave-uniform)
    auto resolve = [&](const int d) {
      const int ilo = d < 0 ? 1 - d : 1;
      const int ihi = min(n - 1, m - 1 - d);
      if (w0 < ilo || w0 + W - 1 > ihi) return;  // window not inside the table on this diagonal
      // forwards from the window: first mismatching row after it (or ihi + 1)
      int e = w0 + W;
      bool ended = false;
      const int fwd_limit = min(ihi, w0 + P + W - 1);  // last row of the NEXT aligned window
      while (e <= fwd_limit) {
        const int row = e + lane;
        const bool bad = row <= fwd_limit && (uint32_t)__popc(lsrc[row] ^ ldst[B + row + d]) > threshold;
        if (COUNT) groups += 1;
        const unsigned long long mm = __ballot(bad);
        if (mm) {
          e += __ffsll((long long)mm) - 1;
          ended = true;
          break;
        }
        e += 64;
      }
      if (!ended) {
        if (fwd_limit == w0 + P + W - 1) return;  // the run also covers the next window: it reports the run
        e = ihi + 1;                                // the run reaches the table edge (comparator.rs:197)
      }
      const int b = e - 1;
      // backwards from the window: last mismatching row before it (or ilo - 1)
      int a = ilo;
      int q = w0 - 1;
      while (q >= ilo) {
        const int row = q - lane;
        const bool bad = row >= ilo && (uint32_t)__popc(lsrc[row] ^ ldst[B + row + d]) > threshold;
        if (COUNT) groups += 1;
        const unsigned long long mm = __ballot(bad);
        if (mm) {
          a = q - (__ffsll((long long)mm) - 1) + 1;
          break;
        }
        q -= 64;
      }
      const int len = b - a + 1;
      if (len >= min_len) emit_run((uint32_t)b, (uint32_t)(b + d), (uint32_t)len);
    };

    // Survivors of the head rows.  On unrelated random hashes there are none and the window ends here; real audio
    // is self-similar (sustained notes: a diagonal that matched three rows running is likely to match the next
    // one), a few of the wave's 64 R diagonals do survive, and walking the remaining rows for all of them would
    // cost 64 R cells per row to test those few.  Up to sparse_max survivors are therefore finished one at a time,
    // the W - H remaining cells of a diagonal side by side in as many lanes; more than that (sustained
    // sounds, silence) goes on row by row as before.  Both paths test exactly the same cells.
    unsigned long long alive[R];
    int survivors = 0;
#pragma unroll
    for (int r = 0; r < R; r++) {
      alive[r] = __ballot(ok[r]);
      survivors += __popcll(alive[r]);
    }
    if (COUNT) survived += (unsigned long long)survivors;
    if (survivors == 0) continue;
    if (!LEGACY || sparse_max < 0) {
      // Default.  Every cell of the window on this lane's R diagonals is already in E: a diagonal that survived the head
      // rows in ANY lane has its remaining rows tested by all lanes straight from registers -- W - H cells per lane, no LDS
      // round trip, no lane shuffling, nothing the next window waits for but arithmetic.  (The sparse form below re-reads
      // the survivor's cells from LDS into neighbouring lanes: one dependent LDS latency per survivor, and with ~1.1
      // survivors per window on audio that latency, not the arithmetic, was what a wave spent its time on.)
#pragma unroll
      for (int r = 0; r < R; r++) {
        if (alive[r] == 0) continue;  // wave-uniform
        uint32_t mm = miss[r];
#pragma unroll
        for (int k = 0; k < W - H; k++) {
          const int s = tail_row(k, W, H);
          mm |= (uint32_t)__popc(sv[s] ^ E[s + r]) + bias;
        }
        if (COUNT) groups += W - H;
        unsigned long long cand = __ballot(mm < 32u);
        while (cand) {
          const int src_lane = __ffsll((long long)cand) - 1;
          cand &= cand - 1;
          resolve(D0 + src_lane * R + r);
        }
      }
      continue;
    }
    if constexpr (LEGACY) {
    if (survivors <= sparse_max) {
      // lane k (k < W - H) takes the k-th remaining row; lanes beyond repeat the last of them
      constexpr int kTail = W - H;
      uint32_t sv_tail = sv[tail_row(kTail - 1, W, H)];
      int tail_off = tail_row(kTail - 1, W, H);
#pragma unroll
      for (int k = kTail - 2; k >= 0; k--) {
        sv_tail = lane == k ? sv[tail_row(k, W, H)] : sv_tail;
        tail_off = lane == k ? tail_row(k, W, H) : tail_off;
      }
#pragma unroll
      for (int r = 0; r < R; r++) {
        unsigned long long cand = alive[r];
        while (cand) {
          const int src_lane = __ffsll((long long)cand) - 1;
          cand &= cand - 1;
          const int d = D0 + src_lane * R + r;  // wave-uniform
          const int row = w0 + tail_off;
          const bool bad = (uint32_t)__popc(sv_tail ^ ldst[B + row + d]) > threshold;
          if (COUNT) groups += 1;
          if (__ballot(bad) == 0) resolve(d);
        }
      }
      continue;
    }
#pragma unroll
    for (int k = 0; k < W - H; k++) {
      const int s = tail_row(k, W, H);
#pragma unroll
      for (int r = 0; r < R; r++) miss[r] |= (uint32_t)__popc(sv[s] ^ E[s + r]) + bias;
    }
#pragma unroll
    for (int r = 0; r < R; r++) ok[r] = miss[r] < 32u;
    if (COUNT) groups += (W - H) * R;
    // ---- candidates: resolved one at a time by the whole wave ----
#pragma unroll
    for (int r = 0; r < R; r++) {
      unsigned long long cand = __ballot(ok[r]);
      while (cand) {
        const int src_lane = __ffsll((long long)cand) - 1;
        cand &= cand - 1;
        resolve(D0 + src_lane * R + r);
      }
    }
    }  // LEGACY
  }
  }  // bands of this wave
  flush_runs();
  if (COUNT && lane == 0) {
    atomicAdd(eval_groups, groups);
    atomicAdd(eval_groups + 1, survived);  // diagonals that passed the head rows, over all windows
  }
}

// ---- sampled path on the matrix pipe (large launches by themselves, build_plan; NEEDLE_HIP_SCAN_MFMA=0 / 1 forces) ------
// The same aligned windows, the same head test and the same candidates as hamming_runs_sampled_kernel, with the first stage
// as int8 matrix products: scan_mfma_kernel.h (round 5; round 4's form -- one product per head row, OR of the four results,
// survivors queued -- is in the history: commit 76713c3, and what was measured on the way in profiles/NOTES.md).
__host__ __device__ constexpr int mfma_windows(int n, int min_len, int W) {
  return (n - 1 - W) >= 0 && min_len - W + 1 > 0 ? (n - 1 - W) / (min_len - W + 1) + 1 : 0;  // w0 = 1 + k P, w0 + W - 1 <= n - 1
}
typedef int mfma_v4i __attribute__((ext_vector_type(4)));
typedef int mfma_v8i __attribute__((ext_vector_type(8)));
typedef float mfma_v16f __attribute__((ext_vector_type(16)));

#include "scan_mfma_kernel.h"

// ---- simhash of every emitted run (comparator.rs:149-153,226-229) -----------------------------------------------
// The scan kernels leave NeedleHipRun.problem = index of the problem descriptor; this pass computes
// chromaprint's simhash32 over the L+1 hashes [end-len ..= end] of both sequences and replaces the index by
// the caller's tag.  One wave per run.
//
// simhash32 needs, per bit position, how many of the slice's hashes have it set: the column sums of a (hashes x 32) bit
// matrix.  Round 3 counted a column with one ballot + scalar popcount per bit and 64 hashes: 4 instructions x 32 bits per
// block, 1500 per run of 370 hashes on both sides -- at library scale (4 M runs) the kernel was bound by exactly that
// instruction count (10.9 ms of a 221 ms job).  Now a block of 64 hashes is TRANSPOSED in registers: lanes 0-31 and 32-63
// each hold a 32 x 32 bit matrix, one row (hash) per lane; five exchange steps (partner lane ^ 16, 8, 4, 2, 1 through
// ds_swizzle, a rotate and a bit-field insert with per-lane constants: three instructions a step) leave lane i with
// COLUMN 31 - i, and one v_bcnt adds that column's ones to the lane's counter: 16 instructions per 64 hashes instead of
// 128.  The slice's blocks are loaded together (eight in flight per side), so a run costs one trip to memory per side.
struct TransposeLane {  // per-lane constants of the five steps
  uint32_t rot[5], mask[5];
};
__device__ __forceinline__ TransposeLane transpose_lane(uint32_t lane) {
  TransposeLane t;
  uint32_t m = 0x0000FFFFu;
#pragma unroll
  for (int k = 0; k < 5; k++) {
    const uint32_t j = 16u >> k;
    const bool hi = (lane & j) != 0u;
    t.rot[k] = hi ? 32u - j : j;          // rotate right: low lanes take the partner's bits from j places up, high lanes from j down
    t.mask[k] = hi ? m << j : m;
    m ^= m << (j >> 1);
  }
  return t;
}
template <int PATTERN>
__device__ __forceinline__ uint32_t swizzle_xor(uint32_t x) {  // lane ^ (PATTERN >> 10) within 32 lanes: LDS crossbar, no memory
  return (uint32_t)__builtin_amdgcn_ds_swizzle((int)x, PATTERN);
}
// x: one hash per lane -> lane i (of each half of the wave): bit column 31 - i of its half's 32 hashes
__device__ __forceinline__ uint32_t transpose32(uint32_t x, const TransposeLane &t) {
#define NEEDLE_TSTEP(K, J)                                                              \
  {                                                                                     \
    const uint32_t y = swizzle_xor<((J) << 10) | 0x1F>(x);                              \
    const uint32_t r = __builtin_amdgcn_alignbit(y, y, t.rot[K]);                      \
    x = (r & t.mask[K]) | (x & ~t.mask[K]);                                             \
  }
  NEEDLE_TSTEP(0, 16) NEEDLE_TSTEP(1, 8) NEEDLE_TSTEP(2, 4) NEEDLE_TSTEP(3, 2) NEEDLE_TSTEP(4, 1)
#undef NEEDLE_TSTEP
  return x;
}
__device__ __forceinline__ uint32_t wave_simhash32(const uint32_t *__restrict__ slice, uint32_t count, uint32_t lane, const TransposeLane &t) {
  constexpr int kBlocks = 8;                       // 512 hashes per turn
  uint32_t ones = 0;                               // lane i: hashes of its half (so far) with bit 31 - (i & 31) set
  for (uint32_t q0 = 0; q0 < count; q0 += 64 * kBlocks) {
    uint32_t h[kBlocks];
#pragma unroll
    for (int u = 0; u < kBlocks; u++) {
      const uint32_t q = q0 + 64 * u + lane;
      h[u] = q < count ? slice[q] : 0u;            // beyond the slice: no ones
    }
#pragma unroll
    for (int u = 0; u < kBlocks; u++) {
      if (q0 + 64 * u >= count) break;             // (wave-uniform)
      ones += (uint32_t)__popc(transpose32(h[u], t));
    }
  }
  ones += (uint32_t)__shfl_xor((int)ones, 32);     // both halves' hashes
  // bit 31 - i set iff ones > zeros (a tie leaves it clear): lanes 0 .. 31 give the word, bit-reversed
  const unsigned long long set = __builtin_amdgcn_ballot_w64(2u * ones > count);
  return __brev((uint32_t)set);
}

__global__ __launch_bounds__(256) void simhash_runs_kernel(const uint32_t *__restrict__ hashes,
                                                           const SearchProblem *__restrict__ problems,
                                                           NeedleHipRun *__restrict__ runs, uint32_t capacity,
                                                           const uint32_t *__restrict__ count) {
  const uint32_t total = min(*count, capacity);
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, waves = (gridDim.x * blockDim.x) >> 6;
  const uint32_t lane = threadIdx.x & 63;
  __builtin_amdgcn_s_setprio(3);  // a tail kernel beside the next job's first pass: its waves issue first (fingerprint.hip, shared-CU overlap)
  const TransposeLane t = transpose_lane(lane);
  for (uint32_t k = wave; k < total; k += waves) {
    const NeedleHipRun r = runs[k];
    const SearchProblem pr = problems[r.problem];
    const uint32_t src_hash = wave_simhash32(hashes + pr.src_off + (r.src_end - r.len), r.len + 1u, lane, t);
    const uint32_t dst_hash = wave_simhash32(hashes + pr.dst_off + (r.dst_end - r.len), r.len + 1u, lane, t);
    if (lane == 0) {
      runs[k].problem = pr.tag;
      runs[k].src_match_hash = src_hash;
      runs[k].dst_match_hash = dst_hash;
    }
  }
}

// What a launch needs besides its inputs: the device descriptors, the grid and the kernel choice.  Kept with the
// inputs it was derived from, so that a job that is run again (a library has ~n^2 / 2 pairs) rebuilds nothing.
// The sampled scan's window shape: rows per aligned window W and head rows H.  (8, 3) unless NEEDLE_HIP_SCAN_SHAPE="W,H"
// names another of the instantiated ones (tools/scan_shape_sweep.py; every shape tests the same cells of a candidate
// and emits the same runs -- the shapes differ in how many cells they look at before giving a window up).
struct ScanShape {
  int w = kSampleW, h = kSampleHead;
};
ScanShape scan_shape() {
  ScanShape sh;
  if (const char *e = getenv("NEEDLE_HIP_SCAN_SHAPE")) {
    int w = 0, h = 0;
    if (std::sscanf(e, "%d,%d", &w, &h) == 2 && (w == 4 || w == 8 || w == 16) && h >= 2 && h <= 4 && h < w) {
      sh.w = w;
      sh.h = h;
    }
  }
  return sh;
}
using SampledKernel = void (*)(const uint32_t *, const SearchProblem *, int, uint32_t, NeedleHipRun *, uint32_t, uint32_t *, int, int,
                               unsigned long long *);
template <bool COUNT>
SampledKernel sampled_kernel(ScanShape sh, bool legacy) {
#define NEEDLE_SHAPE(W_, H_) \
  if (sh.w == W_ && sh.h == H_) return hamming_runs_sampled_kernel<kBandR, W_, COUNT, H_, true>;
  NEEDLE_SHAPE(4, 2) NEEDLE_SHAPE(4, 3) NEEDLE_SHAPE(8, 2) NEEDLE_SHAPE(8, 4) NEEDLE_SHAPE(16, 2) NEEDLE_SHAPE(16, 3) NEEDLE_SHAPE(16, 4)
#undef NEEDLE_SHAPE
  if (legacy) return hamming_runs_sampled_kernel<kBandR, kSampleW, COUNT, kSampleHead, true>;
  return hamming_runs_sampled_kernel<kBandR, kSampleW, COUNT, kSampleHead, false>;   // (the other shapes are tuning runs: LEGACY form)
}

struct SearchPlan {
  std::vector<NeedleHipSeq> seqs;          // inputs ...
  std::vector<NeedleHipProblem> problems;
  int mode[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // ... and the switches that steer the choice ([3]: threshold beyond the sampled kernel's bit trick,
                                           // [4]: the sampled path's first stage on the matrix pipe, [5]: LDS limit of a staged pair in bytes
                                           // (NEEDLE_HIP_SEARCH_LDS_LIMIT), [6]: the matrix-pipe form's waves per workgroup, [7]: its test
                                           // switches -- bits 0..6 NEEDLE_HIP_MFMA_SPLITS, bit 8 _SINGLE, bit 9 _NO_IMAGES).  EVERY switch
                                           // build_plan looks at is in here: the cached plan is keyed by it
  std::vector<SearchProblem> meta;         // derived: the pairs the chosen kernel can stage, then the oversize ones
  size_t staged = 0;                       // how many of meta go to the chosen kernel
  uint64_t oversize_blocks = 0;            // grid of the unstaged kernel over meta[staged..]
  uint64_t blocks = 0;
  size_t lds_bytes = 0;
  bool sampled = false, fast = false, mfma = false;
  int mfma_waves = 0, mfma_splits = 1;      // scan_mfma_kernel.h: waves per workgroup, workgroups per group
  std::vector<M2ImageSeq> image_seqs;       // ... and the source sequences whose window images a pre-pass builds (empty: workgroups build their own)
  uint32_t image_windows = 0, image_min_len = 0;
  uint64_t mfma_products = 0;                 // v_mfma instructions a launch of the matrix-pipe form issues
  int bands_per_wave = 1;
  bool valid = false;
  bool matches(const NeedleHipSeq *s, size_t ns, const NeedleHipProblem *p, size_t np, const int *m) const {
    return valid && seqs.size() == ns && problems.size() == np && std::memcmp(mode, m, sizeof(mode)) == 0 &&
           std::memcmp(seqs.data(), s, ns * sizeof(NeedleHipSeq)) == 0 &&
           std::memcmp(problems.data(), p, np * sizeof(NeedleHipProblem)) == 0;
  }
};

int device_cus() {  // compute units of the current device (256 on MI355X)
  int cus = 256, dev = 0;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  return cus > 0 ? cus : 256;
}

Status build_plan(const NeedleHipSeq *seqs, size_t num_seqs, const NeedleHipProblem *problems, size_t num_problems,
                  const int *mode, SearchPlan *plan) {
  plan->valid = false;
  std::vector<SearchProblem> &meta = plan->meta;
  meta.clear();
  meta.reserve(num_problems);
  uint64_t blocks = 0;
  for (size_t p = 0; p < num_problems; p++) {
    const NeedleHipProblem &pr = problems[p];
    if (pr.src_seq >= num_seqs || pr.dst_seq >= num_seqs)
      return Status::Make(NeedleError_InvalidArgument, "hamming_runs: problem references a missing sequence");
    if (pr.min_len == 0) return Status::Make(NeedleError_InvalidArgument, "hamming_runs: min_len must be >= 1");
    const NeedleHipSeq &a = seqs[pr.src_seq], &b = seqs[pr.dst_seq];
    if (a.len < 2 || b.len < 2) continue;  // no cell with i >= 1 and j >= 1 (comparator.rs:165-167,179)
    SearchProblem m;
    m.src_off = a.offset;
    m.n = a.len;
    m.dst_off = b.offset;
    m.m = b.len;
    m.min_len = pr.min_len;
    m.tag = pr.tag;
    m.block_base = 0;
    m.pad = 0;
    meta.push_back(m);
  }
  plan->staged = 0;
  plan->oversize_blocks = 0;
  if (!meta.empty()) {
    // The band kernel needs min_len >= its checkpoint spacing for every problem; otherwise (tiny minimum
    // durations) the general one-lane-per-diagonal kernel handles the launch.
    uint32_t smallest = 0xFFFFFFFFu;
    for (const SearchProblem &m : meta) smallest = std::min(smallest, m.min_len);
    const bool generic_only = mode[0] != 0;
    const bool sampled = !generic_only && smallest >= (uint32_t)(2 * scan_shape().w - 1 + 8) && mode[1] == 0 && mode[3] == 0;
    const bool fast = !generic_only && !sampled && smallest >= (uint32_t)(kBandR * kBandU);
    // What a pair needs in LDS under the chosen kernel.  A pair beyond the CU's 160 KiB (a window of more than ~2.7 h
    // of audio at step 1) goes to the end of the table and is scanned from HBM by the unstaged kernel: one such
    // video must not fail the search of the whole library (the reference has no bound).
    const size_t lds_limit = (size_t)mode[5];  // 160 KiB, or NEEDLE_HIP_SEARCH_LDS_LIMIT (tests)
    // The matrix-pipe form of the sampled scan pays on large launches whose windows fill their row tiles of 32 (two sources
    // of one destination share a workgroup's tiles): measured against the vector form, scan kernel, 1.41 x at 39 060 pairs of
    // 24-minute windows (2 x 39 windows in 3 tiles), 1.66 x at 499 500 pairs of 45-minute ones (2 x 73 in 5); 0.97 x on the
    // headline's 378 pairs, where the scan hides beside the next job's first pass anyway.
    auto windows_of = [&](const SearchProblem &m) { return (uint64_t)mfma_windows((int)m.n, (int)m.min_len, kSampleW); };
    auto same_group = [](const SearchProblem &a, const SearchProblem &b) {
      return a.dst_off == b.dst_off && a.m == b.m && a.min_len == b.min_len;
    };
    // The matrix-pipe form gives neighbours of the (sorted) table that share destination and minimum length to ONE
    // workgroup -- up to eight sources -- while its LDS still lets the intended number of workgroups share a CU (LDS comes
    // in granules of 1280 bytes: 53 760 is a third of a CU's, 81 920 half).
    const int waves = mode[6];
    const size_t group_budget = waves == 4 ? 53760 : waves == 8 ? 81920 : 163840;
    const size_t max_members = (size_t)kM2Members;
    auto group_need = [&](const SearchProblem &a, uint64_t windows) { return m2_lds_words(a.m, windows, waves) * sizeof(uint32_t); };
    bool mfma = sampled && mode[4] == 1;
    // From how many sequence pairs: with FP4 products the matrix-pipe kernel is the faster kernel from ~2000 pairs of
    // 24-minute windows up (0.127 against 0.142 ms at 2016, 0.19 / 0.33 at 4950, 0.26 / 0.62 at 9730, 0.38 / 0.83 at 16 110)
    // and jobs are no slower with it from there on either (two in flight, the scan beside the next job's first pass: 1.56 /
    // 1.56, 2.65 / 2.71, 4.08 / 4.12, 5.66 / 5.96 ms).  (The int8 form lost inside jobs below 16 384 pairs -- the vector
    // form's small workgroups fit the holes beside the first pass better -- and a job took it only from there; mode[5], "a
    // call with nothing beside it", is not looked at any more.)
    const size_t mfma_from = 2048;
    const bool candidate = sampled && (mode[4] == 1 || (mode[4] == 2 && meta.size() >= mfma_from));
    if (candidate) {  // the table sorted by destination (the runs carry the index of their own entry: its order is free)
      std::stable_sort(meta.begin(), meta.end(), [](const SearchProblem &a, const SearchProblem &b) {
        if (a.dst_off != b.dst_off) return a.dst_off < b.dst_off;
        if (a.m != b.m) return a.m < b.m;
        return a.min_len < b.min_len;
      });
      if (!mfma) {  // automatic: where the windows of a group fill at least 70 % of its row tiles of 32
        uint64_t windows = 0, rows = 0;
        for (size_t i = 0; i < meta.size(); i++) {
          uint64_t w = windows_of(meta[i]);
          for (size_t k = 1; k < max_members && i + 1 < meta.size() && same_group(meta[i], meta[i + 1]) &&
                             group_need(meta[i], w + windows_of(meta[i + 1])) <= group_budget; k++)
            w += windows_of(meta[++i]);
          windows += w;
          rows += (w + 31) / 32 * 32;
        }
        mfma = rows > 0 && 10 * windows >= 7 * rows;
      }
    }
    auto lds_need = [&](const SearchProblem &m) {
      if (mfma) return group_need(m, windows_of(m));
      return ((fast || sampled) ? (size_t)m.m + 2 * kBandB : (size_t)m.n + m.m) * sizeof(uint32_t);
    };
    // The matrix-pipe kernel packs its queue items as unit << 17 | batch of row tiles << 11 | lane << 5 | flag bit and a window /
    // member word as w0 | member << 28 (scan_mfma_kernel.h): a staged entry needs m < 65536, fewer than 8192 windows (per group) and n < 2^28.  The 160 KiB of LDS imply
    // the first two (m <= ~40 900) only while the limit is the hardware's; checked here so that an overridden limit or a
    // future layout cannot break the packing silently -- such an entry goes to the oversize partition.
    auto mfma_packable = [&](const SearchProblem &m) { return m.m < 65536u && windows_of(m) < 8192u && m.n < (1u << 28); };
    auto stageable = [&](const SearchProblem &m) { return lds_need(m) <= lds_limit && (!mfma || mfma_packable(m)); };
    std::stable_partition(meta.begin(), meta.end(), stageable);
    size_t staged = 0;
    while (staged < meta.size() && stageable(meta[staged])) staged++;
    size_t lds_bytes = 0;
    int bands_per_wave = 1;
    // pad of a group's first entry = how many followers; a follower (pad bit 31) owns no workgroups
    size_t groups = 0;
    if (mfma) {
      const bool single = (mode[7] & 0x100) != 0;  // NEEDLE_HIP_MFMA_SINGLE (tests, measurements): one source per workgroup
      for (size_t i = 0; i < staged;) {
        SearchProblem &a = meta[i];
        a.pad = 0;
        uint64_t windows = windows_of(a);
        size_t k = 1;
        while (!single && k < max_members && i + k < staged && same_group(a, meta[i + k]) &&
               windows + windows_of(meta[i + k]) < 8192 &&
               group_need(a, windows + windows_of(meta[i + k])) <= group_budget) {
          windows += windows_of(meta[i + k]);
          meta[i + k].pad = 0x80000000u;
          k++;
        }
        a.pad = (uint32_t)(k - 1);
        i += k;
        groups++;
      }
    }
    // the index of group G's first entry rides in the pad field of the table's G-th entry (bits 8 .. 30), so a workgroup
    // finds its group with one load
    if (mfma) {
      if (staged >= ((size_t)1 << 23)) return Status::Make(NeedleError_InvalidArgument, "hamming_runs: too many problems for one launch");
      size_t g = 0;
      for (size_t i = 0; i < staged; i++)
        if (!(meta[i].pad & 0x80000000u)) meta[g++].pad |= (uint32_t)i << 8;
    }
    // one workgroup per group; a launch of few groups splits each over several workgroups (column units strided)
    int splits = 1;
    if (mfma && groups > 0) {
      const uint64_t slots = (uint64_t)device_cus() * (uint64_t)(waves == 4 ? 3 : waves == 8 ? 2 : 1);
      splits = (int)std::min<uint64_t>(8, std::max<uint64_t>(1, (2 * slots + groups - 1) / groups));
      if (mode[7] & 0x7F) splits = mode[7] & 0x7F;  // NEEDLE_HIP_MFMA_SPLITS (tests, tuning), 1 .. 64
    }
    if (fast || sampled) {
      uint64_t fb = 0, total_bands = 0;
      for (size_t i = 0; i < staged; i++) total_bands += ((uint64_t)meta[i].n + meta[i].m - 3 + kBandB - 1) / kBandB;
      // enough workgroups to fill the chip several times over with one band per wave; beyond that, more bands
      // per wave so that each workgroup's staging of the destination sequence serves more of its pair
      if (sampled)
        while (bands_per_wave < 8 && total_bands / (4 * (uint64_t)bands_per_wave * 2) >= 16384) bands_per_wave *= 2;
      if (mode[2] > 0) bands_per_wave = mode[2];  // NEEDLE_HIP_BANDS_PER_WAVE: tests, tuning
      for (size_t i = 0; i < staged; i++) {
        SearchProblem &m = meta[i];
        m.block_base = (uint32_t)fb;
        if (mfma && (m.pad & 0x80000000u)) continue;  // its workgroups are its group's (same base as the entry after it)
        const uint64_t diags = (uint64_t)m.n + m.m - 3;
        const uint64_t bands = (diags + kBandB - 1) / kBandB;
        const uint64_t per_block = 4 * (uint64_t)(sampled ? bands_per_wave : 1);
        fb += mfma ? (uint64_t)splits : (bands + per_block - 1) / per_block;
        if (mfma) {
          uint64_t windows = windows_of(m);
          for (uint32_t f = 1; f <= (m.pad & 0xFFu); f++) windows += windows_of(meta[i + f]);
          lds_bytes = std::max(lds_bytes, group_need(m, windows));
        } else {
          lds_bytes = std::max(lds_bytes, lds_need(m));
        }
      }
      blocks = fb;
    } else {
      for (size_t i = 0; i < staged; i++) {
        SearchProblem &m = meta[i];
        m.block_base = (uint32_t)blocks;
        blocks += ((uint64_t)m.n + m.m - 3 + kDiagsPerBlock - 1) / kDiagsPerBlock;
        lds_bytes = std::max(lds_bytes, lds_need(m));
      }
    }
    uint64_t ob = 0;
    for (size_t i = staged; i < meta.size(); i++) {  // one workgroup = 256 consecutive diagonals, as in the staged form
      meta[i].block_base = (uint32_t)ob;
      ob += ((uint64_t)meta[i].n + meta[i].m - 3 + kDiagsPerBlock - 1) / kDiagsPerBlock;
    }
    if (blocks > 0x7FFFFFFFull || ob > 0x7FFFFFFFull)
      return Status::Make(NeedleError_InvalidArgument, "hamming_runs: too many problems for one launch");
    plan->staged = staged;
    plan->oversize_blocks = ob;
    plan->sampled = sampled;
    plan->mfma = mfma;
    plan->mfma_products = 0;
    if (mfma)
      for (size_t i = 0; i < staged; i++) {
        const SearchProblem &m = meta[i];
        if (m.pad & 0x80000000u) continue;
        uint64_t w = windows_of(m);
        for (uint32_t f = 1; f <= (m.pad & 0xFFu); f++) w += windows_of(meta[i + f]);
        if (w > 0 && m.m >= (uint32_t)kSampleW + 1) {
          uint64_t col_blocks = ((uint64_t)m.m - kSampleW) / 32 + 1;
          col_blocks = (col_blocks + kM2ColBlocks - 1) / kM2ColBlocks * kM2ColBlocks;  // whole units are multiplied
          plan->mfma_products += (w + 31) / 32 * col_blocks * (kM2Heads / 2);
        }
      }
    // The matrix-pipe form's window images (scan_mfma_kernel.h): one per distinct source sequence, when the launch has ONE
    // minimum length (a library of equal-length videos; otherwise the window positions differ per pair and workgroups
    // build their own).  block_base of a staged entry -- not read by that kernel otherwise -- = first window of its source's image.
    plan->image_seqs.clear();
    plan->image_windows = plan->image_min_len = 0;
    if (mfma && staged > 0 && !(mode[7] & 0x200)) {  // (bit 9: NEEDLE_HIP_MFMA_NO_IMAGES)
      bool one = true;
      for (size_t i = 1; i < staged && one; i++) one = meta[i].min_len == meta[0].min_len;
      if (one) {
        std::unordered_map<uint64_t, uint32_t> first_of;   // (src_off, n) -> first window
        uint64_t total = 0;
        for (size_t i = 0; i < staged; i++) {
          const uint64_t key = ((uint64_t)meta[i].src_off << 32) | meta[i].n;
          auto it = first_of.find(key);
          if (it == first_of.end()) {
            const uint32_t w = (uint32_t)windows_of(meta[i]);
            it = first_of.emplace(key, (uint32_t)total).first;
            plan->image_seqs.push_back(M2ImageSeq{meta[i].src_off, meta[i].n, (uint32_t)total, w});
            total += w;
          }
          meta[i].block_base = it->second;
        }
        if (total == 0 || total * kM2ImageWords * sizeof(uint32_t) > ((uint64_t)1 << 31)) {
          plan->image_seqs.clear();           // (nothing to build, or more than 2 GB of images: built in place)
        } else {
          plan->image_windows = (uint32_t)total;
          plan->image_min_len = meta[0].min_len;
        }
      }
    }
    plan->fast = fast;
    plan->mfma_waves = waves;
    plan->mfma_splits = splits;
    plan->bands_per_wave = bands_per_wave;
    plan->lds_bytes = lds_bytes;
  }
  plan->blocks = blocks;
  plan->seqs.assign(seqs, seqs + num_seqs);
  plan->problems.assign(problems, problems + num_problems);
  std::memcpy(plan->mode, mode, sizeof(plan->mode));
  plan->valid = true;
  return Status::Ok();
}

// NEEDLE_HIP_SCAN_MFMA: 1 = the matrix-pipe form of the sampled scan wherever it applies, 0 = never, unset = where it pays
// (build_plan).  It applies to thresholds <= 15 and the default window shape.  -> mode[4]: 0 off, 1 forced, 2 automatic
int mfma_request(uint32_t threshold) {
  if (threshold > 15u || scan_shape().w != kSampleW || getenv("NEEDLE_HIP_SCAN_SHAPE") || getenv("NEEDLE_HIP_SCAN_COUNT")) return 0;
  // (a counting launch and a launch of another window shape are the vector form's by definition)
  const char *e = getenv("NEEDLE_HIP_SCAN_MFMA");
  if (!e) return 2;
  return atoi(e) != 0 ? 1 : 0;
}

// The matrix-pipe form's workgroup shape (measurements; the default is what won, profiles/NOTES.md round 5):
// NEEDLE_HIP_MFMA_WAVES 4 / 8 / 12 / 16 waves per workgroup (3 / 2 / 1 / 1 workgroups per CU; 8 and 16: four waves per
// SIMD, one accumulator pair).
int mfma_waves() {
  int waves = 8;
  if (const char *e = getenv("NEEDLE_HIP_MFMA_WAVES")) {
    const int w = atoi(e);
    if (w == 4 || w == 8 || w == 12 || w == 16) waves = w;
  }
  return waves;
}
using Mfma2Kernel = void (*)(const uint32_t *, const SearchProblem *, int, uint32_t, NeedleHipRun *, uint32_t, uint32_t *, int, const uint32_t *);
Mfma2Kernel mfma2_kernel(int waves) {
  return waves == 4 ? hamming_runs_mfma2_kernel<kSampleW, 4, 3> : waves == 16 ? hamming_runs_mfma2_kernel<kSampleW, 16, 4>
       : waves == 12 ? hamming_runs_mfma2_kernel<kSampleW, 12, 3> : hamming_runs_mfma2_kernel<kSampleW, 8, 4>;
}

std::atomic<int32_t> g_last_form{0};
std::atomic<uint64_t> g_last_products{0};

struct SearchWorkspace {
  DeviceBuffer<SearchProblem> problems;
  PinnedStage stage;
  DescriptorUpload<SearchProblem> upload;
  SearchPlan plan;            // of the last launch
  DeviceBuffer<M2ImageSeq> image_seqs;   // matrix-pipe form: the sequences whose window images the pre-pass builds ...
  PinnedStage image_stage;
  DescriptorUpload<M2ImageSeq> image_upload;
  DeviceBuffer<uint32_t> images;         // ... and the images
  bool lds_attr_set = false;  // > 64 KiB of dynamic LDS needs an explicit opt-in, per device
  unsigned long long *eval_groups = nullptr;  // device; NEEDLE_HIP_SCAN_COUNT=1 launches add to it
};
std::mutex g_ws_mu;
std::map<int, SearchWorkspace *> g_ws;

SearchWorkspace *workspace() {
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lock(g_ws_mu);
  auto it = g_ws.find(dev);
  if (it != g_ws.end()) return it->second;
  SearchWorkspace *w = new SearchWorkspace();
  g_ws[dev] = w;
  return w;
}


}  // namespace

void gpu_scan_last_launch(int32_t *form, uint64_t *matrix_products) {
  *form = g_last_form.load();
  *matrix_products = g_last_products.load();
}

Status gpu_hamming_runs_device(const uint32_t *d_hashes, const NeedleHipSeq *seqs, size_t num_seqs,
                               const NeedleHipProblem *problems, size_t num_problems, uint32_t threshold,
                               NeedleHipRun *d_runs, uint32_t capacity, uint32_t *d_count, bool sync, bool count_is_zero) {
  std::lock_guard<std::recursive_mutex> gpu_lock(gpu_mutex());
  Status s = ensure_device();
  if (!s.ok()) return s;
  hipStream_t stream = library_stream();
  if (!count_is_zero) NEEDLE_HIP_TRY(hipMemsetAsync(d_count, 0, sizeof(uint32_t), stream));
  const int mode[8] = {getenv("NEEDLE_HIP_GENERIC_SEARCH") != nullptr, getenv("NEEDLE_HIP_BAND_SEARCH") != nullptr,
                       getenv("NEEDLE_HIP_BANDS_PER_WAVE") ? std::max(1, atoi(getenv("NEEDLE_HIP_BANDS_PER_WAVE"))) : 0,
                       threshold > 31u,   // every cell matches at 32: the band / generic kernels take such a launch
                       mfma_request(threshold),
                       getenv("NEEDLE_HIP_SEARCH_LDS_LIMIT") ? std::max(1, atoi(getenv("NEEDLE_HIP_SEARCH_LDS_LIMIT"))) : 160 * 1024,
                       mfma_waves(),
                       (getenv("NEEDLE_HIP_MFMA_SPLITS") ? std::max(1, std::min(64, atoi(getenv("NEEDLE_HIP_MFMA_SPLITS")))) : 0) |
                           (getenv("NEEDLE_HIP_MFMA_SINGLE") ? 0x100 : 0) | (getenv("NEEDLE_HIP_MFMA_NO_IMAGES") ? 0x200 : 0)};
  SearchWorkspace *ws = workspace();
  SearchPlan &plan = ws->plan;
  const bool reuse = plan.matches(seqs, num_seqs, problems, num_problems, mode);
  if (!reuse && !(s = build_plan(seqs, num_seqs, problems, num_problems, mode, &plan)).ok()) return s;
  const std::vector<SearchProblem> &meta = plan.meta;
  const uint64_t blocks = plan.blocks;
  const size_t lds_bytes = plan.lds_bytes;
  const bool sampled = plan.sampled, fast = plan.fast;
  const int bands_per_wave = plan.bands_per_wave;
  if (!meta.empty()) {
    // same inputs as last time and the table still where it was put: nothing to compare or upload
    if (!(reuse && ws->upload.resident_at == ws->problems.ptr && ws->problems.ptr) &&
        !(s = ws->upload.put(&ws->problems, &ws->stage, meta, stream)).ok())
      return s;
    if (lds_bytes > 64 * 1024 && !ws->lds_attr_set) {
      NEEDLE_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(hamming_runs_kernel<true>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      NEEDLE_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(hamming_runs_band_kernel<kBandR, kBandU>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      for (int w : {4, 8, 12, 16})
        NEEDLE_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(mfma2_kernel(w)),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      for (int w : {4, 8, 16})
        for (int h : {2, 3, 4}) {
          if (h >= w) continue;
          for (bool legacy : {false, true}) {
            NEEDLE_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(sampled_kernel<false>(ScanShape{w, h}, legacy)),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            NEEDLE_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(sampled_kernel<true>(ScanShape{w, h}, legacy)),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
          }
        }
      ws->lds_attr_set = true;
    }
    const int staged = (int)plan.staged, oversize = (int)(meta.size() - plan.staged);
    g_last_form.store(staged > 0 ? (sampled ? (plan.mfma ? 4 : 3) : fast ? 2 : 1) : 1);
    g_last_products.store(staged > 0 && plan.mfma ? plan.mfma_products : 0);
    if (staged > 0) {
      KernelTimer timer("hamming_runs");
      if (sampled) {
        int sparse_max = -1;  // survivors of the head rows finished from registers (the kernel's default path)
        if (const char *e = getenv("NEEDLE_HIP_SPARSE_MAX")) sparse_max = std::max(0, atoi(e));  // tests, tuning: the LDS forms; 0 = row by row
        if (getenv("NEEDLE_HIP_SCAN_COUNT")) {  // diagnostic: the same scan, counting what it issues
          if (!ws->eval_groups) {
            NEEDLE_HIP_TRY(hipMalloc((void **)&ws->eval_groups, 2 * sizeof(unsigned long long)));
            NEEDLE_HIP_TRY(hipMemsetAsync(ws->eval_groups, 0, 2 * sizeof(unsigned long long), stream));
          }
          hipLaunchKernelGGL(sampled_kernel<true>(scan_shape(), sparse_max >= 0), dim3((uint32_t)blocks), dim3(256), lds_bytes, stream, d_hashes,
                             ws->problems.ptr, staged, threshold, d_runs, capacity, d_count, bands_per_wave, sparse_max,
                             ws->eval_groups);
        } else if (plan.mfma) {
          const uint32_t *images = nullptr;
          if (!plan.image_seqs.empty()) {   // the sources' window images, once per launch (the hashes are this job's)
            if (!(s = ws->image_upload.put(&ws->image_seqs, &ws->image_stage, plan.image_seqs, stream)).ok() ||
                !(s = ws->images.reserve((size_t)plan.image_windows * kM2ImageWords)).ok())
              return s;
            const uint32_t grid = (uint32_t)std::min<uint64_t>(8192, ((uint64_t)plan.image_windows * kM2Rows + 255) / 256);
            hipLaunchKernelGGL(m2_window_images_kernel<kSampleW>, dim3(grid), dim3(256), 0, stream, d_hashes, ws->image_seqs.ptr,
                               (int)plan.image_seqs.size(), plan.image_windows, plan.image_min_len, ws->images.ptr);
            images = ws->images.ptr;
          }
          hipLaunchKernelGGL(mfma2_kernel(plan.mfma_waves), dim3((uint32_t)blocks), dim3(64 * plan.mfma_waves),
                             lds_bytes, stream, d_hashes, ws->problems.ptr, staged, threshold, d_runs, capacity, d_count,
                             plan.mfma_splits, images);
#if NEEDLE_M2_LAB & 8
          {
            unsigned long long c[8], z[8] = {};
            (void)hipStreamSynchronize(stream);
            (void)hipMemcpyFromSymbol(c, HIP_SYMBOL(m2_dbg), sizeof c);
            (void)hipMemcpyToSymbol(HIP_SYMBOL(m2_dbg), z, sizeof z);
            fprintf(stderr, "m2 counts: resolve calls %llu, trips %llu, mismatches handled %llu, walks continued %llu, extra forward trips %llu, "
                            "windows past the tail rows %llu, items %llu, windows past the head rows %llu\n", c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7]);
          }
#endif
        } else {
          hipLaunchKernelGGL(sampled_kernel<false>(scan_shape(), sparse_max >= 0), dim3((uint32_t)blocks), dim3(256), lds_bytes, stream, d_hashes,
                             ws->problems.ptr, staged, threshold, d_runs, capacity, d_count, bands_per_wave, sparse_max,
                             (unsigned long long *)nullptr);
        }
      }
      else if (fast)
        hipLaunchKernelGGL((hamming_runs_band_kernel<kBandR, kBandU>), dim3((uint32_t)blocks), dim3(256), lds_bytes,
                           stream, d_hashes, ws->problems.ptr, staged, threshold, d_runs, capacity, d_count);
      else
        hipLaunchKernelGGL(hamming_runs_kernel<true>, dim3((uint32_t)blocks), dim3(256), lds_bytes, stream, d_hashes,
                           ws->problems.ptr, staged, 0u, threshold, d_runs, capacity, d_count);
    }
    if (oversize > 0) {  // pairs too long to stage: scanned from HBM, behind the others in the table
      KernelTimer timer("hamming_runs_unstaged");
      hipLaunchKernelGGL(hamming_runs_kernel<false>, dim3((uint32_t)plan.oversize_blocks), dim3(256), 0, stream, d_hashes,
                         ws->problems.ptr + staged, oversize, (uint32_t)staged, threshold, d_runs, capacity, d_count);
    }
    {
      KernelTimer timer("simhash_runs");
      hipLaunchKernelGGL(simhash_runs_kernel, dim3(2048), dim3(256), 0, stream, d_hashes, ws->problems.ptr, d_runs,
                         capacity, d_count);
    }
    NEEDLE_HIP_TRY(hipGetLastError());
  }
  if (sync) NEEDLE_HIP_TRY(hipStreamSynchronize(stream));
  return Status::Ok();
}

// Cell evaluations ISSUED by the sampled scan's counting launches (NEEDLE_HIP_SCAN_COUNT=1) since the last reset:
// groups of xor / popcount / compare over a wave's 64 lanes, times 64.
Status gpu_scan_issued_evaluations(uint64_t *lane_evaluations, bool reset, uint64_t *head_survivors) {
  std::lock_guard<std::recursive_mutex> gpu_lock(gpu_mutex());
  Status s = ensure_device();
  if (!s.ok()) return s;
  SearchWorkspace *ws = workspace();
  unsigned long long groups[2] = {0, 0};
  hipStream_t stream = library_stream();
  if (ws->eval_groups) {
    NEEDLE_HIP_TRY(hipMemcpyAsync(groups, ws->eval_groups, sizeof(groups), hipMemcpyDeviceToHost, stream));
    if (reset) NEEDLE_HIP_TRY(hipMemsetAsync(ws->eval_groups, 0, sizeof(groups), stream));
    NEEDLE_HIP_TRY(hipStreamSynchronize(stream));
  }
  *lane_evaluations = (uint64_t)groups[0] * 64u;
  if (head_survivors) *head_survivors = (uint64_t)groups[1];
  return Status::Ok();
}

namespace {
// grow-only device buffers kept per device (guarded by gpu_mutex()): a search-only call over a library pays for its
// copies and kernels, not for hipMalloc / hipFree
struct HostCallBuffers {
  DeviceBuffer<uint32_t> d_hashes, d_count;
  DeviceBuffer<NeedleHipRun> d_runs;
  void *pinned = nullptr;  // results of the device epilogue + its failure count
  size_t pinned_bytes = 0;
  // the caller's hash arena in pinned memory (gpu_pinned_arena_*), and what of it is already on its way to d_hashes
  uint32_t *arena = nullptr;
  size_t arena_words = 0;
  bool arena_out = false;
  const uint32_t *sent = nullptr;  // gpu_prefetch_hashes: this many words from here were enqueued for d_hashes
  size_t sent_words = 0;
};
HostCallBuffers *host_call_buffers() {
  static std::map<int, HostCallBuffers *> all;
  int dev = 0;
  (void)hipGetDevice(&dev);
  HostCallBuffers *&hb = all[dev];
  if (!hb) hb = new HostCallBuffers();
  return hb;
}
}  // namespace

// A search call's hash arena in PINNED host memory: the comparator fills it on host threads, and it goes up without the
// runtime's staging copy (3.4 MB from pageable memory: 0.3 ms of a 2.3 ms search-only call) and, enqueued right after the
// fill (gpu_prefetch_hashes), beside the building of the pair table.  One buffer per device, one call at a time: a second
// concurrent call gets nullptr and uses pageable memory as before.
uint32_t *gpu_pinned_arena_acquire(size_t words) {
  std::lock_guard<std::recursive_mutex> gpu_lock(gpu_mutex());
  if (!ensure_device().ok()) return nullptr;
  HostCallBuffers *hb = host_call_buffers();
  if (hb->arena_out || getenv("NEEDLE_HIP_PAGEABLE_ARENA")) return nullptr;  // (the switch: tests of the pageable path)
  if (words > hb->arena_words) {
    if (hb->arena) (void)hipHostFree(hb->arena);
    hb->arena = nullptr;
    hb->arena_words = 0;
    void *p = nullptr;
    const size_t want = words + words / 8 + 1024;
    if (hipHostMalloc(&p, want * sizeof(uint32_t), hipHostMallocDefault) != hipSuccess) {
      (void)hipGetLastError();
      return nullptr;
    }
    hb->arena = static_cast<uint32_t *>(p);
    hb->arena_words = want;
  }
  hb->arena_out = true;
  return hb->arena;
}
void gpu_pinned_arena_release(uint32_t *p) {
  if (!p) return;
  std::lock_guard<std::recursive_mutex> gpu_lock(gpu_mutex());
  HostCallBuffers *hb = host_call_buffers();
  if (hb->sent) (void)hipStreamSynchronize(library_stream());  // (a prefetch nobody picked up: it must not outlive the buffer's next use)
  hb->sent = nullptr;
  hb->arena_out = false;
}
// Enqueues the upload of a pinned arena into the search calls' device buffer; the next gpu_search_results_host /
// gpu_hamming_runs_host call with the same arena does not copy again.  Failure is not an error: that call copies.
void gpu_prefetch_hashes(const uint32_t *hashes, size_t num_hashes) {
  std::lock_guard<std::recursive_mutex> gpu_lock(gpu_mutex());
  if (!ensure_device().ok()) return;
  HostCallBuffers *hb = host_call_buffers();
  hb->sent = nullptr;
  if (hashes != hb->arena || !hb->arena_out || num_hashes == 0) return;
  if (!hb->d_hashes.reserve(num_hashes).ok()) return;
  if (hipMemcpyAsync(hb->d_hashes.ptr, hashes, num_hashes * sizeof(uint32_t), hipMemcpyHostToDevice, library_stream()) != hipSuccess) {
    (void)hipGetLastError();
    return;
  }
  hb->sent = hashes;
  hb->sent_words = num_hashes;
}
namespace {
// the arena goes up unless gpu_prefetch_hashes has sent exactly this one already
Status upload_hashes(HostCallBuffers *hb, const uint32_t *hashes, size_t num_hashes, hipStream_t stream) {
  const bool sent = hb->sent == hashes && hb->sent_words == num_hashes && num_hashes > 0;
  hb->sent = nullptr;
  if (!sent) NEEDLE_HIP_TRY(hipMemcpyAsync(hb->d_hashes.ptr, hashes, num_hashes * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
  return Status::Ok();
}
}  // namespace

Status gpu_search_results_host(const uint32_t *hashes, size_t num_hashes, const NeedleHipSeq *seqs, size_t num_seqs,
                               const NeedleHipProblem *problems, size_t num_problems, uint32_t threshold, EpilogueJob job,
                               std::vector<NeedleHipSearchResult> *results, uint32_t *failed, std::vector<NeedleHipRun> *runs,
                               size_t *num_runs) {
  std::lock_guard<std::recursive_mutex> gpu_lock(gpu_mutex());
  Status s = ensure_device();
  if (!s.ok()) return s;
  for (size_t i = 0; i < num_seqs; i++)
    if ((uint64_t)seqs[i].offset + seqs[i].len > num_hashes)
      return Status::Make(NeedleError_InvalidArgument, "hamming_runs: sequence outside the hash arena");
  hipStream_t stream = library_stream();
  HostCallBuffers *hb = host_call_buffers();
  if (!(s = hb->d_hashes.reserve(std::max<size_t>(num_hashes, 1))).ok() || !(s = hb->d_count.reserve(1)).ok()) return s;
  const size_t want = ((size_t)job.n + 1) * sizeof(NeedleHipSearchResult);
  if (want > hb->pinned_bytes) {
    if (hb->pinned) (void)hipHostFree(hb->pinned);
    hb->pinned = nullptr;
    hb->pinned_bytes = 0;
    NEEDLE_HIP_TRY(hipHostMalloc(&hb->pinned, want, hipHostMallocDefault));
    hb->pinned_bytes = want;
  }
  NeedleHipSearchResult *pinned = static_cast<NeedleHipSearchResult *>(hb->pinned);
  uint32_t *pinned_failed = reinterpret_cast<uint32_t *>(pinned + job.n);
  if (!(s = upload_hashes(hb, hashes, num_hashes, stream)).ok()) return s;
  uint32_t capacity = (uint32_t)std::min<uint64_t>(
      0x7fffffffu, std::max<uint64_t>({(uint64_t)1 << 16, (uint64_t)hb->d_runs.count, 3 * (uint64_t)num_problems}));
  for (int attempt = 0; attempt < 2; attempt++) {
    if (!(s = hb->d_runs.reserve(capacity)).ok()) return s;
    s = gpu_hamming_runs_device(hb->d_hashes.ptr, seqs, num_seqs, problems, num_problems, threshold, hb->d_runs.ptr, capacity,
                                hb->d_count.ptr, false, false);
    if (!s.ok()) return s;
    // the run count first (one small copy behind the scan): an attempt whose list overflowed is repeated without its
    // epilogue ever running, and the epilogue's workspaces (144 bytes per run, kept for the life of the process) are sized
    // by the runs there ARE, not by the capacity of the list (3 runs per pair of room: 770 MB at 2000 videos)
    uint32_t found = 0;
    NEEDLE_HIP_TRY(hipMemcpyAsync(&found, hb->d_count.ptr, sizeof(found), hipMemcpyDeviceToHost, stream));
    NEEDLE_HIP_TRY(hipStreamSynchronize(stream));
    if (found <= capacity) {
      job.num_segments = 1;
      job.segment_count[0] = hb->d_count.ptr;
      job.segment_runs[0] = hb->d_runs.ptr;
      job.segment_capacity = capacity;
      job.max_runs = (uint64_t)found + (found >> 3) + 1024;   // (grow-only workspaces: a margin so that the next call's few more runs fit)
      if (!(s = gpu_epilogue_enqueue(job, stream, pinned, pinned_failed)).ok()) return s;
      NEEDLE_HIP_TRY(hipStreamSynchronize(stream));
      *num_runs = found;
      *failed = *pinned_failed;
      results->assign(pinned, pinned + job.n);
      runs->clear();
      if (*failed && found) {  // the host epilogue names the failing video in the reference's order: it needs the list
        runs->resize(found);
        NEEDLE_HIP_TRY(hipMemcpy(runs->data(), hb->d_runs.ptr, found * sizeof(NeedleHipRun), hipMemcpyDeviceToHost));
      }
      return Status::Ok();
    }
    capacity = found;  // the scan is deterministic: a second pass with the exact size fits
  }
  return Status::Make(NeedleError_Unknown, "hamming_runs: run list did not fit after resize");
}

Status gpu_hamming_runs_host(const uint32_t *hashes, size_t num_hashes, const NeedleHipSeq *seqs, size_t num_seqs,
                             const NeedleHipProblem *problems, size_t num_problems, uint32_t threshold,
                             std::vector<NeedleHipRun> *runs) {
  std::lock_guard<std::recursive_mutex> gpu_lock(gpu_mutex());
  Status s = ensure_device();
  if (!s.ok()) return s;
  for (size_t i = 0; i < num_seqs; i++)
    if ((uint64_t)seqs[i].offset + seqs[i].len > num_hashes)
      return Status::Make(NeedleError_InvalidArgument, "hamming_runs: sequence outside the hash arena");
  hipStream_t stream = library_stream();
  HostCallBuffers *hb = host_call_buffers();
  DeviceBuffer<uint32_t> &d_hashes = hb->d_hashes, &d_count = hb->d_count;
  DeviceBuffer<NeedleHipRun> &d_runs = hb->d_runs;
  if (!(s = d_hashes.reserve(std::max<size_t>(num_hashes, 1))).ok()) return s;
  if (!(s = d_count.reserve(1)).ok()) return s;
  if (!(s = upload_hashes(hb, hashes, num_hashes, stream)).ok()) return s;
  // run-list capacity: what the buffer already holds from earlier calls, or a few runs per pair (a library whose
  // episodes share an intro has at least one per pair); a list that still does not fit costs a second, exact pass
  uint32_t capacity = (uint32_t)std::min<uint64_t>(
      0x7fffffffu, std::max<uint64_t>({(uint64_t)1 << 16, (uint64_t)d_runs.count, 3 * (uint64_t)num_problems}));
  for (int attempt = 0; attempt < 2; attempt++) {
    if (!(s = d_runs.reserve(capacity)).ok()) return s;
    s = gpu_hamming_runs_device(d_hashes.ptr, seqs, num_seqs, problems, num_problems, threshold, d_runs.ptr,
                                capacity, d_count.ptr, false, false);
    if (!s.ok()) return s;
    uint32_t found = 0;
    NEEDLE_HIP_TRY(hipMemcpyAsync(&found, d_count.ptr, sizeof(found), hipMemcpyDeviceToHost, stream));
    NEEDLE_HIP_TRY(hipStreamSynchronize(stream));
    if (found <= capacity) {
      runs->resize(found);
      if (found)
        NEEDLE_HIP_TRY(hipMemcpy(runs->data(), d_runs.ptr, found * sizeof(NeedleHipRun), hipMemcpyDeviceToHost));
      return Status::Ok();
    }
    capacity = found;  // the scan is deterministic: a second pass with the exact size fits
  }
  return Status::Make(NeedleError_Unknown, "hamming_runs: run list did not fit after resize");
}

// ---- diagnostic: the integer-VALU ceiling of the scan's per-cell work on THIS device ------------------------------
// "calibrate with a popcount micro-benchmark on the box and use the measured ceiling as the
// denominator".  A cell here is exactly the four instructions of the band scan's inner loop -- v_xor_b32,
// v_bcnt_u32_b32, v_cmp, v_cndmask -- on registers only, no memory, eight independent cells per lane and row, at the
// occupancy that gives the highest rate.
namespace {
// The sampled scan's own blend (its head rows): per cell v_xor, v_bcnt (+ bias), a share of a v_or3 and of one v_cmp per
// diagonal and three rows, the masks counted in the scalar unit -- on registers only.  Counted as 3 instructions per cell.
template <int R>
__global__ __launch_bounds__(256) void valu_ceiling_sampled_kernel(uint32_t *out, int rows, uint32_t threshold, uint32_t seed) {
  uint32_t Wd[R + 2];
#pragma unroll
  for (int r = 0; r < R + 2; r++) Wd[r] = seed * (threadIdx.x + 1) * (r + 3) + blockIdx.x;
  const uint32_t bias = 31u - threshold;
  uint32_t sv = __builtin_amdgcn_readfirstlane(seed ^ blockIdx.x);
  int total = 0;
  for (int i = 0; i < rows; i += 3) {
    uint32_t s0 = sv * 1664525u + 1013904223u, s1 = s0 * 1664525u + 1013904223u, s2 = s1 * 1664525u + 1013904223u;
    sv = s2;
    int survivors = 0;
#pragma unroll
    for (int r = 0; r < R; r++) {
      const uint32_t miss = ((uint32_t)__popc(s0 ^ Wd[r]) + bias) | ((uint32_t)__popc(s1 ^ Wd[r + 1]) + bias) |
                            ((uint32_t)__popc(s2 ^ Wd[r + 2]) + bias);
      survivors += __popcll(__ballot(miss < 32u));
    }
    total += survivors;
    if (survivors == (int)seed) Wd[0] ^= 1u;  // never (seed is large): a use the compiler cannot remove
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)total + Wd[0];
}

template <int R>
__global__ __launch_bounds__(256) void valu_ceiling_kernel(uint32_t *out, int rows, uint32_t threshold, uint32_t seed) {
  uint32_t W[R];
  int Z[R];
#pragma unroll
  for (int r = 0; r < R; r++) {
    W[r] = seed * (threadIdx.x + 1) * (r + 3) + blockIdx.x;
    Z[r] = -1;
  }
  uint32_t sv = __builtin_amdgcn_readfirstlane(seed ^ blockIdx.x);
  for (int i = 0; i < rows; i++) {
    sv = sv * 1664525u + 1013904223u;  // scalar update: stands in for the s_load'ed source hash of the row
#pragma unroll
    for (int r = 0; r < R; r++) {
      const uint32_t c = (uint32_t)__popc(sv ^ W[r]);
      Z[r] = (c <= threshold) ? Z[r] : i;
    }
  }
  uint32_t acc = 0;
#pragma unroll
  for (int r = 0; r < R; r++) acc += (uint32_t)Z[r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
}  // namespace

Status gpu_int_valu_ceiling(double *cells_per_second) {
  std::lock_guard<std::recursive_mutex> gpu_lock(gpu_mutex());
  Status s = ensure_device();
  if (!s.ok()) return s;
  constexpr int R = 8;
  const int rows = 20000;
  int cus = 256;
  int dev = 0;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  hipStream_t stream = library_stream();
  DeviceBuffer<uint32_t> out;
  if (!(s = out.reserve((size_t)cus * 8 * 256)).ok()) return s;
  hipEvent_t a = nullptr, b = nullptr;
  NEEDLE_HIP_TRY(hipEventCreate(&a));
  NEEDLE_HIP_TRY(hipEventCreate(&b));
  double best = 0.0;
  for (int blocks_per_cu : {2, 4, 8}) {
    const int grid = cus * blocks_per_cu;
    hipLaunchKernelGGL(valu_ceiling_kernel<R>, dim3(grid), dim3(256), 0, stream, out.ptr, 100, 10u, 12345u);  // warm-up
    (void)hipEventRecord(a, stream);
    hipLaunchKernelGGL(valu_ceiling_kernel<R>, dim3(grid), dim3(256), 0, stream, out.ptr, rows, 10u, 12345u);
    (void)hipEventRecord(b, stream);
    if (hipEventSynchronize(b) != hipSuccess) break;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, a, b) == hipSuccess && ms > 0.f)
      best = std::max(best, (double)grid * 256 * R * rows / (ms * 1e-3));
    // the sampled kernel's three-instruction cell, expressed in cells of four instructions so that 4 x the figure
    // returned stays "lane-instructions per second" for either blend (bench.py's search_roofline)
    constexpr int RS = 7;
    hipLaunchKernelGGL(valu_ceiling_sampled_kernel<RS>, dim3(grid), dim3(256), 0, stream, out.ptr, 99, 10u, 12345678u);
    (void)hipEventRecord(a, stream);
    hipLaunchKernelGGL(valu_ceiling_sampled_kernel<RS>, dim3(grid), dim3(256), 0, stream, out.ptr, rows - rows % 3, 10u, 12345678u);
    (void)hipEventRecord(b, stream);
    if (hipEventSynchronize(b) != hipSuccess) break;
    if (hipEventElapsedTime(&ms, a, b) == hipSuccess && ms > 0.f) {
      const double rate3 = (double)grid * 256 * RS * (rows - rows % 3) / (ms * 1e-3);  // cells of 3 instructions per second
      if (getenv("NEEDLE_HIP_TRACE"))
        std::fprintf(stderr, "[needle_hip] int-VALU ceiling, %d workgroups per CU: sampled blend %.3e cells/s (x3 = %.3e lane-instructions/s), "
                             "band blend so far %.3e cells/s (x4 = %.3e)\n", blocks_per_cu, rate3, 3.0 * rate3, best, 4.0 * best);
      best = std::max(best, rate3 * 3.0 / 4.0);
    }
  }
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  if (best <= 0.0) return Status::Make(NeedleError_Unknown, "integer-VALU ceiling measurement failed");
  *cells_per_second = best;
  return Status::Ok();
}

}  // namespace needle
