// stft_chroma_kernel: the dominant kernel of the analyze half, in a header of its own so that the product
// (fingerprint.hip) and the timing laboratory (tools/stft_lab.hip) compile the SAME source.  LAB = 0 is the product;
// non-zero LAB bits switch pieces off or change the schedule for measurements (results are then wrong on purpose).
#pragma once

#include <hip/hip_runtime.h>

#include <stdint.h>
#include <type_traits>

#include "fp_core.h"

namespace needle {
namespace stft {

using core::cd;
constexpr int kHop = 1365, kBands = 12;

struct FpStream {
  uint64_t pcm_off;     // s16 values
  uint64_t item_off;    // where this stream's kept items go in d_items
  uint32_t frames;
  uint32_t frame_base;  // prefix of frames
  uint32_t fir_rows;    // frames - 4 (or 0)
  uint32_t fir_base;
  uint32_t kept;
  uint32_t kept_base;
  uint32_t pair_base;   // prefix of ceil(frames / 2): the STFT kernel transforms two frames per FFT
  uint32_t tile_base;   // prefix of ceil(kept / items per tile): tiles of features_classify_kernel
};

// index of the stream whose [base, next base) range holds g; `base` is a field of FpStream
template <uint32_t FpStream::*BASE>
__device__ __forceinline__ int find_stream(const FpStream *streams, int n, uint32_t g) {
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    int mid = (lo + hi + 1) >> 1;
    if (streams[mid].*BASE <= g) lo = mid; else hi = mid - 1;
  }
  return lo;
}

// ---- kernel 1: one 256-thread workgroup per PAIR of consecutive frames ---------------------------------------
// z = frameA + i*frameB through one 4096-point complex FFT (fp_core.h, radix 16 x 3, padded LDS), split into
// the two real spectra, |X|^2 over bins 10..1307 folded into 12 pitch classes per frame.  A workgroup walks
// kPairsPerBlock CONSECUTIVE pairs of one region of the batch, so the 3x overlap between neighbouring frames
// (hop 1365 of 4096) is re-read from this XCD's L2 rather than from HBM.
constexpr int kPairsPerBlock = 16;  // default; NEEDLE_STFT_PAIRS overrides for tuning

// LDS-only workgroup barrier: waits for this wave's LDS traffic, not for its outstanding global loads
// (__syncthreads() would also drain vmcnt and with it the prefetch of the next pair's PCM).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Orders this wave's LDS writes before its later LDS reads for the compiler; the hardware executes one wave's LDS
// operations in order, so lanes of the same wave see each other's data without a workgroup barrier.
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// x from another lane of the row, by a DPP control word (no LDS round trip, unlike __shfl_xor)
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double x) {
  // every lane of a row has a source under these controls; bound_ctrl spares the "old value" register and its move
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}

struct PairSrc {
  const int16_t *a, *b;  // first value of frame A / frame B (B = A when the stream has an odd frame count)
  uint64_t row;          // chroma row of frame A
  bool has_b;
};

// LAB bits (timing experiments only; 0 = product; results are wrong with any of bits 1..8 set):
//   1  barrier 1 (previous pair's fold reads done -> stage-0 stores) -> none
//   2  barrier 2 (stage-0 stores -> stage-1 reads) -> wave fence
//   4  EXTRA workgroup barrier between the publish stores and the partner reads (the round-1 schedule had one)
//   8  barrier 3 (power stores -> fold reads) -> wave fence
//  16  round-1 order: all butterflies of a stage, then all its stores; barrier 1 after the first butterflies
//  32  the stores of a tail spaced out under the next tail (fp_core.h kStoreSpaced) instead of right behind their own
//  64  round-1 arithmetic in the 16-point transforms (168 instead of 144 instructions each)
enum : int { kLabNoB1 = 1, kLabNoB2 = 2, kLabExtraB = 4, kLabNoB3 = 8, kLabSerial = 16, kLabSpaced = 32, kLabPlainFft = 64 };

// LISTED: the workgroups walk a device-resident LIST of chunks (chunk c = pairs [c * pairs_per_block, + pairs_per_block)
// of the batch) instead of the whole batch: the f64 recomputation of the frames whose f32 first pass could not be
// certified (fingerprint.hip).  The list and its length are written by an earlier kernel of the same stream; the grid
// is fixed and a workgroup takes entries blockIdx.x, + gridDim.x, ... until the list is exhausted (every wave reads
// the same length, so every wave leaves).  A pair's result does not depend on which workgroup computes it, or in which
// company: the chroma rows written here are bit-identical to those of the plain launch.
struct ChunkList {
  const uint32_t *chunks;
  const uint32_t *count;
};

// WAVES: minimum waves per SIMD the kernel is compiled for (2: 226-229 VGPRs, the form that runs alone; 3: the LISTED
// recomputation held to 168 VGPRs -- 63 of them spilled, twice as slow alone -- so that its workgroup fits the hole ONE
// retiring first-pass workgroup leaves when it runs BESIDE the next job's first pass: fingerprint.hip, "shared-CU overlap")
template <int CH, int LAB = 0, bool LISTED = false, int WAVES = 2>
__global__ __launch_bounds__(256, WAVES) void stft_chroma_kernel(const int16_t *__restrict__ pcm,
                                                             const FpStream *__restrict__ streams, int num_streams,
                                                             const cd *__restrict__ tw,
                                                             const double *__restrict__ wcos, core::WindowConst wconst,
                                                             const uint16_t *__restrict__ bin_slot,
                                                             const uint32_t *__restrict__ fold_tab,
                                                             double *__restrict__ chroma, uint32_t total_pairs,
                                                             uint32_t pairs_per_block, ChunkList list = ChunkList{nullptr, nullptr}) {
  extern __shared__ cd lds[];  // core::kLds2Slots complex slots
  using raw_t = typename std::conditional<CH == 1, int16_t, int>::type;  // one sample, or one packed L|R pair
  const int t = threadIdx.x;
  if (LISTED) __builtin_amdgcn_s_setprio(3);  // the recomputation runs beside the next job's first pass: its waves issue first
  // Workgroups are dealt to the 8 XCDs round-robin (blockIdx.x & 7) and each XCD has its own L2.  Neighbouring
  // stretches of the timeline share 2731 of their samples (the frame overlap), so each XCD gets one contiguous
  // eighth of the timeline: the workgroups that run side by side on an XCD are then neighbours in time and the
  // overlap is re-read from that XCD's L2 (the grid is a multiple of 8).  Measured: fabric fetches per launch
  // 482 MB either way for 445 MB of PCM -- the boundary overlap of a plain mapping is only 28 MB and was mostly
  // caught by the memory-side cache already -- and no change in kernel time; kept because it is never worse.
  uint32_t first, last;
  uint32_t list_at = blockIdx.x, list_len = 0;
  if (LISTED) {
    list_len = *list.count;
    if (list_at >= list_len) return;
  } else {
    const uint32_t per_xcd = gridDim.x >> 3;
    const uint32_t logical = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    first = logical * pairs_per_block;
    last = min(total_pairs, first + pairs_per_block);
    if (first >= last) return;
  }
  const cd base0 = tw[t], base1 = tw[16 * (t & 15)];  // W_4096^t, W_4096^{16 n0}: loop-invariant twiddle bases

  // ---- loop invariants of this thread.  Where the powers of its six bins go and its share of the fold stay in
  // registers (four words; the compiler derives the ten LDS addresses from them once: 227 VGPRs, no scratch -- check
  // both after any change to this kernel, the budget is 256).  The window seeds do not fit as well (the compiler then
  // also keeps all sixteen window values of the thread and spills inside the loop): they live in the thread's private
  // pad slot of the LDS image and are read back once per pair. ------------------------------------------------------
  const bool folds = t < kBands * core::kClassLanes;
  core::Words4 inv;
#pragma unroll
  for (int j = 0; j < core::kBinsPerThread; j += 2) {  // where the powers of its six bins go in the power image
    uint32_t idx[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int kf = core::dif_bin_of(t, j + h);
      idx[h] = (kf >= core::kMinBin && kf < core::kMaxBin) ? bin_slot[kf - core::kMinBin] : (uint32_t)core::kPowerTrashSlot;
    }
    inv.w[j >> 1] = core::pack_slots(idx[0], idx[1]);
  }
  // its share of the pitch-class fold: 12 classes x 16 lanes (one DPP row per class), each lane both frames
  // fp_core.h PowerLayout: first slot | positions << 16; the fourth wave has no class: "no positions from slot 0"
  const uint32_t fold_entry = folds ? fold_tab[t] : 0u;
  // window recurrence (fp_core.h window_step), seeded per thread with cos(theta t) and cos(theta (t - 256))
  core::lds_put(lds, core::thread_pad_slot(t), cd{wcos[t + 256], wcos[t]});
  if (t == 0) core::lds_put(lds, core::kPowerZeroSlot, cd{0.0, 0.0});  // first read after the loop's barriers

  do {  // LISTED: once per listed chunk of this workgroup; otherwise once
  if (LISTED) {
    first = list.chunks[list_at] * pairs_per_block;
    last = min(total_pairs, first + pairs_per_block);
  }
  // ---- the stream (region of the batch) the current pair belongs to; consecutive pairs rarely change it ------
  int si = find_stream<&FpStream::pair_base>(streams, num_streams, first);
  FpStream st = streams[si];
  uint32_t st_end = st.pair_base + (st.frames + 1) / 2;
  auto locate = [&](uint32_t g) {  // g must not decrease between calls
    while (g >= st_end) {
      st = streams[++si];
      st_end = st.pair_base + (st.frames + 1) / 2;
    }
    const uint32_t fa = 2 * (g - st.pair_base);
    PairSrc p;
    p.has_b = fa + 1 < st.frames;
    p.a = pcm + st.pcm_off + (uint64_t)fa * kHop * CH;
    p.b = p.has_b ? p.a + kHop * CH : p.a;  // no frame B: read A again, zeroed after conversion
    p.row = (uint64_t)st.frame_base + fa;
    return p;
  };
  using reg_t = int;  // one 16-bit sample sign-extended by the load, or one packed L|R pair
  reg_t ra[16], rb[16];
  // PCM of both frames: issued one pair ahead, while the previous pair's powers are still being produced (the
  // spectrum registers are dead by then).
  auto issue_loads = [&](const PairSrc &p) {
    const raw_t *qa = reinterpret_cast<const raw_t *>(p.a), *qb = reinterpret_cast<const raw_t *>(p.b);
    // an opaque copy of the thread index keeps these loads (and their addresses) in the loop; laundering the
    // POINTER would do that too but loses its address space: flat loads, which count in lgkmcnt and so stall
    // every LDS-only barrier
    int tt = t;
    asm volatile("" : "+v"(tt));
#pragma unroll
    for (int k = 0; k < 16; k++) {
      ra[k] = (reg_t)qa[tt + 256 * k];
      rb[k] = (reg_t)qb[tt + 256 * k];
    }
  };
  // The pitch-class fold of a pair runs one pair late, interleaved with the next pair's sample conversion, in two
  // halves (five reads, eight samples, five reads, eight samples) so that the conversion hides the reads' latency and
  // the loaded power pairs never take more than 20 registers.  It is branch-free: the fourth wave, which owns no
  // class, reads spectrum values and the zero slot and adds them like the others -- divergent regions around the loads
  // and around the sums made the register allocator spill the loaded values between them.  Only the store is
  // conditional.
  constexpr int kFoldHalf = core::kClassLaneMax / 2;
  auto fold_tree_store = [&](cd acc, const PairSrc &p, int tt, bool store) {  // tt: the opaque copy of t
    // fixed-order tree over the class's 16 lanes (fp_core.h class_tree_partner)
    acc = cd{acc.x + dpp_f64<0xB1>(acc.x), acc.y + dpp_f64<0xB1>(acc.y)};    // quad_perm [1,0,3,2]
    acc = cd{acc.x + dpp_f64<0x4E>(acc.x), acc.y + dpp_f64<0x4E>(acc.y)};    // quad_perm [2,3,0,1]
    acc = cd{acc.x + dpp_f64<0x141>(acc.x), acc.y + dpp_f64<0x141>(acc.y)};  // row_half_mirror
    acc = cd{acc.x + dpp_f64<0x140>(acc.x), acc.y + dpp_f64<0x140>(acc.y)};  // row_mirror
    if (store && (tt & 15) == 0 && tt < kBands * core::kClassLanes) {
      double *out = chroma + p.row * kBands;  // uniform base + a 32-bit lane offset
      const uint32_t c = (uint32_t)tt >> 4;
      out[c] = acc.x;
      if (p.has_b) out[kBands + c] = acc.y;
    }
  };
  PairSrc cur = locate(first), prev = cur;
  issue_loads(cur);
  // The samples are carried around the loop as sign-extended 32-bit values.  If every value that enters the loop's phi
  // is "sext(load)", the optimiser moves the extension behind the phi, and each sample then costs a global_load_ushort
  // plus a v_bfe_i32 at its use; an opaque first set keeps the extension with the in-loop loads (global_load_sshort).
#pragma unroll
  for (int k = 0; k < 16; k++) asm volatile("" : "+v"(ra[k]), "+v"(rb[k]));

  for (uint32_t g = first; g < last; g++) {
    // all per-thread address arithmetic is redone per pair from this opaque copy of the thread index: kept
    // loop-invariant by the compiler it costs more registers than the kernel has (spills to scratch)
    int tt = t;
    asm volatile("" : "+v"(tt));
    const cd seeds = core::lds_get(lds, core::thread_pad_slot(tt));
    cd r[16];
    double wc = seeds.x, wc_prev = seeds.y;
    auto convert = [&](int k) {
      int sa, sb;
      if (CH == 1) {
        sa = ra[k];
        sb = rb[k];
      } else {  // AudioProcessor::LoadStereo: (L + R) / 2, C truncation
        sa = ((int)(int16_t)ra[k] + (ra[k] >> 16)) / 2;
        sb = ((int)(int16_t)rb[k] + (rb[k] >> 16)) / 2;
      }
      const double w = core::window_step(wconst, &wc, &wc_prev);
      r[k] = cd{(double)sa * w, (double)sb * w};
    };
    {
      cd fv[kFoldHalf];
      core::class_lane_load_part<0, kFoldHalf>(lds, fold_entry, fv);
#pragma unroll
      for (int k = 0; k < 8; k++) convert(k);
      cd acc = fv[0];
#pragma unroll
      for (int i = 1; i < kFoldHalf; i++) acc = core::cadd(acc, fv[i]);
      core::class_lane_load_part<kFoldHalf, core::kClassLaneMax>(lds, fold_entry, fv);
#pragma unroll
      for (int k = 8; k < 16; k++) convert(k);
#pragma unroll
      for (int i = 0; i < core::kClassLaneMax - kFoldHalf; i++) acc = core::cadd(acc, fv[i]);
      fold_tree_store(acc, prev, tt, g != first);
    }
    if (!cur.has_b) {  // odd frame count: the stream's last pair has no frame B (uniform branch)
#pragma unroll
      for (int k = 0; k < 16; k++) r[k].y = 0.0;
    }
    // In-place decimation-in-frequency stages; which exchanges need a workgroup barrier: fp_core.h.  The stores of
    // a stage are issued tail by tail (fp_core.h dif_tail_store) so that the LDS pipeline, whose 128-bit stores cost
    // 13 cycles each, drains them under the arithmetic that follows instead of in a burst in front of a barrier.
    if (LAB & kLabSerial) {
      core::fft16(r);
      if (!(LAB & kLabNoB1)) lds_barrier();
      core::dif0_store(tt, base0, lds, r);
      if (!(LAB & kLabNoB2)) lds_barrier(); else wave_lds_fence();
      core::dif1(tt, base1, lds, r);
      wave_lds_fence();
      core::dif2(tt, lds, r);
      core::dif2_publish(tt, lds, r);
    } else {
      if (!(LAB & kLabNoB1)) lds_barrier();  // every thread has read its share of the previous pair's powers
      constexpr int kMode = ((LAB & kLabSpaced) ? core::kStoreSpaced : 0) | ((LAB & kLabPlainFft) ? core::kPlainFft : 0);
      core::dif0_streamed<kMode>(tt, base0, lds, r);
      if (!(LAB & kLabNoB2)) lds_barrier(); else wave_lds_fence();
      core::dif1_streamed<kMode>(tt, base1, lds, r);
      wave_lds_fence();                // stage 1 -> 2 stays inside 16 consecutive lanes
      core::dif2_streamed<kMode>(tt, lds, r);  // r[out16(j)] = Z[bin b + 16 (t & 15) + 256 j]; publishes j = 10..15
    }
    // publish -> partner reads stays inside the wave (fp_core.h group_k0)
    if (LAB & kLabExtraB) lds_barrier(); else wave_lds_fence();

    // partners Z[N - k] of the six bins: one base + constants, all six reads in flight together
    cd yp[core::kBinsPerThread];
    core::dif_partner_load(tt, lds, yp);
    double pwa[core::kBinsPerThread], pwb[core::kBinsPerThread];
#pragma unroll
    for (int j = 0; j < core::kBinsPerThread; j++) core::dif_power_of(r[core::out16(j)], yp[j], &pwa[j], &pwb[j]);
    // power pairs into dead slots of this wave's own rows (fp_core.h PowerLayout): no barrier after the partner
    // reads; bins outside 10..1307 land in a pad slot nobody reads
#pragma unroll
    for (int j = 0; j < core::kBinsPerThread; j++)
      core::lds_put_bytes(lds, (j & 1) ? core::slot_bytes<1>(inv.w[j >> 1]) : core::slot_bytes<0>(inv.w[j >> 1]), cd{pwa[j], pwb[j]});
    const PairSrc nxt = locate(min(g + 1, last - 1));  // last pair: harmless re-read
    issue_loads(nxt);
    if (!(LAB & kLabNoB3)) lds_barrier(); else wave_lds_fence();  // the power image is complete
    prev = cur;
    cur = nxt;
  }
  {
    cd fv[core::kClassLaneMax];
    core::class_lane_load(lds, fold_entry, fv);
    fold_tree_store(core::class_lane_add(fv), prev, t, true);
  }
  // (LISTED: the next chunk's first stores into the image sit behind barrier 1 of its first pair, which every
  // thread reaches only after these reads)
  list_at += gridDim.x;
  } while (LISTED && list_at < list_len);
}

}  // namespace stft
}  // namespace needle
