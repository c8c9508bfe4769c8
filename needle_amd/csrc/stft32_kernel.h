// stft_chroma32_kernel: the f32 FIRST PASS of the analyze half.
//
// The u32 contract is defined on the f64 pipeline (oracle/ora_chromaprint.c), and stft_chroma_kernel (stft_kernel.h)
// is within ~1.3x of the floor of its f64 decomposition.  This kernel runs the SAME schedule -- two real frames per
// 4096-point complex transform, radix 16 x 3 in place in one padded LDS image, wave-local power image, DPP fold -- in
// f32: half the LDS bytes, half the registers (three to four workgroups per CU instead of two) and the f32 issue
// rate.  Its chroma is NOT the contract's; it is a first pass whose every consumer is certified:
// features_classify_cert_kernel (fingerprint.hip) accepts an item only if all 48 threshold comparisons clear a
// data-dependent radius, and every other item is recomputed from f64 chroma (stft_chroma_kernel over the listed
// chunks of frame pairs + fixup_items_kernel).  What the radius needs from here is the total energy of the frame PAIR (one transform)
// E = sum |x|^2 next to its 12 pitch-class sums: with a strong component outside chromaprint's band (a 5 kHz tone, a
// DC offset) the f32 transform's noise floor is set by E, not by the in-band energy the features are normalised by.
//
// Differences from the f64 kernel besides the width: the window and the twiddle powers are correctly rounded table
// values (an f32 recurrence / running product would add more error than the transform itself); every thread leaves
// the sum of squares of its 16 samples of each frame in the spare column 8 of its stage-2 row and the fourth wave --
// which owns no pitch class -- folds the 256 partials like a class (fp_core.h energy_slot / energy_fold_entry).
#pragma once

#include "stft32_schedule.h"
#include "stft_kernel.h"

namespace needle {
namespace stft {

using core::cf;

template <int CTRL>
__device__ __forceinline__ float dpp_f32(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}

constexpr int kEnergyParts = 4;  // partial sums per frame (one per DPP row of the fourth wave)

// Sixteen complex f32 values from LDS slots base + STRIDE k as sixteen SINGLE ds_read_b64.  Written with the compiler's
// own loads, the backend pairs them into ds_read2_b64, which the LDS serves at 8 cycles per pair (two passes of 4 x 16
// lanes, 32 banks) against 2 + 2 for two ds_read_b64 (2 x 32 lanes over 64 banks; MI355X_MICROARCH.md, LDS table) --
// and this kernel's time follows its LDS cycles one for one (profiles/r03_stft32_lab.log).  The asm loads are invisible
// to the compiler's wait-count bookkeeping, so the batch ends in its own s_waitcnt lgkmcnt(0), tied to every value
// ("+v") so that no use can be scheduled in front of it; "memory" keeps the surrounding LDS stores where they are.
template <int STRIDE>
__device__ __forceinline__ void lds_read16_single(const cf *lds, int base_slot, cf *out) {
  typedef float v2f __attribute__((ext_vector_type(2)));
  const uint32_t addr = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) cf *)lds + (uint32_t)base_slot * 8u;
  v2f v[16];
#pragma unroll
  for (int k = 0; k < 16; k++)
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v[k]) : "v"(addr), "n"(STRIDE * 8 * k) : "memory");
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]),
                 "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15])
               :
               : "memory");
#pragma unroll
  for (int k = 0; k < 16; k++) out[k] = cf{v[k].x, v[k].y};
}

// LAB bits (tools/stft32_lab.hip only; 0 = product; results are wrong or incomplete with any of them set)
enum : int {
  kLab32NoEnergy = 1,   // no energy partials (what the radius' E term costs)
  kLab32WinLoad = 2,    // window values re-read from the table for every pair instead of 16 loop-invariant registers
  kLab32NoB1 = 4, kLab32NoB2 = 8, kLab32NoB3 = 16,  // a workgroup barrier replaced by a wave fence
  kLab32NoFold = 32,    // no fold reads / tree / chroma store
  kLab32NoPower = 64,   // no partner reads, powers, power stores
  kLab32CompilerReads = 512, // the 2 x 16 stage inputs through the compiler's own loads (ds_read2_b64) instead of lds_read16_single
  kLab32NoConflict = 1024,  // power stores and fold reads on a trivially conflict-free (and wrong) slot pattern: the
                            // upper bound of what a conflict-free power image could buy
  kLab32ConsumerTw = 2048,  // twiddles on the consumer side in tan form + the window folded into stage 0's first layer
                            // (fp_core.h head_col_tw / head_col_win): 64 VALU instructions fewer per pair; rows built
                            // in-kernel from the f32 table here (timing only)
  kLab32Tw1Lds = 256,   // stage-1 twiddle powers from a [15][16] table behind the LDS image instead of 30 registers
  kLab32FormatLoad = 4096,  // mono: samples through tbuffer_load_format_x [16, SSCALED] -- the s16 -> f32 conversion happens in
                            // the load (exact: tools/format_load_probe.hip) and the 32 v_cvt_f32_i32 per pair go away
  kLab32Clock = 128,    // thread 0 stamps s_memtime / s_memrealtime around the pair loop into `energy` (as 4 x u64 per
                        // workgroup): the clock the kernel really ran at = d memtime / d memrealtime x 100 MHz
};

template <int CH, int WAVES_PER_SIMD = 3, int LAB = 0>
__global__ __launch_bounds__(256, WAVES_PER_SIMD) void stft_chroma32_kernel(
    const int16_t *__restrict__ pcm, const FpStream *__restrict__ streams, int num_streams, const cf *__restrict__ tw32,
    const float *__restrict__ win32, const uint16_t *__restrict__ bin_slot, const uint32_t *__restrict__ fold_tab,
    double *__restrict__ chroma, float *__restrict__ energy, uint32_t total_pairs, const Stft32Schedule sched,
    uint32_t *__restrict__ zero_words = nullptr, uint32_t num_zero_words = 0) {
  constexpr bool kAsmReads = !(LAB & (kLab32CompilerReads | kLab32ConsumerTw | kLab32Tw1Lds));
  extern __shared__ cf lds32[];  // core::kLds2Slots complex slots of 8 bytes
  cf *const lds = lds32;
  using raw_t = typename std::conditional<CH == 1, int16_t, int>::type;
  const int t = threadIdx.x;
  uint32_t first, last;  // one contiguous eighth of the timeline per XCD, smaller workgroups at its end (stft32_schedule.h)
  stft32_block_range(sched, blockIdx.x, total_pairs, &first, &last);
  // the certification control block (counters + chunk bitmap) of the kernels BEHIND this one starts at zero: cleared
  // here by one workgroup instead of by a memset dispatch of its own in front of them (6 us on a 600 us job)
  if (blockIdx.x == 0)
    for (uint32_t i = threadIdx.x; i < num_zero_words; i += blockDim.x) zero_words[i] = 0u;
  if (first >= last) return;

  // ---- loop invariants: the fifteen twiddle powers of both stages (table values, correctly rounded), the thread's
  // sixteen window values, where the powers of its six bins go, its share of the fold
  cf pw0[16], pw1[16];
  float win[16];
#pragma unroll
  for (int j = 1; j < 16; j++) {
    pw0[j] = tw32[(t * j) & 4095];
    pw1[j] = tw32[(16 * (t & 15) * j) & 4095];
  }
  if (!(LAB & kLab32WinLoad)) {
#pragma unroll
    for (int k = 0; k < 16; k++) win[k] = win32[t + 256 * k];
  }
  float row1[core::kTwRow], row2[core::kTwRow];
  if (LAB & kLab32ConsumerTw) {
    auto build = [&](int m, float *row) {
      int o = 0;
#pragma unroll
      for (int n2 = 0; n2 < 4; n2++) {
        const cf w0 = tw32[(m * n2) & 4095], w1 = tw32[(m * (4 + n2)) & 4095], w2 = tw32[(m * (8 + n2)) & 4095],
                 w3 = tw32[(m * (12 + n2)) & 4095];
        if (n2 != 0) {
          row[o++] = w0.x;
          row[o++] = w0.y;
        }
        row[o++] = w2.y / w2.x;
        row[o++] = w2.x;
        row[o++] = w1.y / w1.x;
        row[o++] = w3.y / w3.x;
        row[o++] = w3.x / w1.x;
        row[o++] = w1.x;
      }
    };
    build(16 * core::thread_k0(t), row1);
    build(core::thread_k0(t) + 16 * (t & 15), row2);
  }
  core::Words4 inv;
#pragma unroll
  for (int j = 0; j < core::kBinsPerThread; j += 2) {
    uint32_t idx[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int kf = core::dif_bin_of(t, j + h);
      idx[h] = (kf >= core::kMinBin && kf < core::kMaxBin) ? bin_slot[kf - core::kMinBin] : (uint32_t)core::kPowerTrashSlot;
    }
    inv.w[j >> 1] = core::pack_slots(idx[0], idx[1]);
  }
  uint32_t fold_entry = t < kBands * core::kClassLanes ? fold_tab[t] : core::energy_fold_entry(t - kBands * core::kClassLanes);
  if (LAB & kLab32NoConflict) fold_entry = (uint32_t)(17 * min(4 * (t & 63), 246) + (t >> 6)) | (10u << 16);
  if (t == 0) core::lds_put(lds, core::kPowerZeroSlot, cf{0.0f, 0.0f});  // first read after the loop's barriers
  // LAB: W^(16 n0 j) at kTw1Base + 16 (j - 1) + n0: the 16 lanes of a group read 16 consecutive slots (conflict-free),
  // the wave's four groups the same ones (broadcast)
  constexpr int kTw1Base = core::kLds2Slots;
  if (LAB & kLab32Tw1Lds) {
    if (t < 16) {
#pragma unroll
      for (int j = 1; j < 16; j++) core::lds_put(lds, kTw1Base + 16 * (j - 1) + t, pw1[j]);
    }
  }

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
    p.b = p.has_b ? p.a + kHop * CH : p.a;
    p.row = (uint64_t)st.frame_base + fa;
    return p;
  };
  using reg_t = int;
  reg_t ra[16], rb[16];
  constexpr bool kFormatLoad = (LAB & kLab32FormatLoad) != 0 && CH == 1;
  typedef int rsrc_t __attribute__((ext_vector_type(4)));
  const int voff = 2 * t;  // byte offset of this thread's first sample inside a frame
  auto issue_loads = [&](const PairSrc &p) {
    if (kFormatLoad) {
      // One buffer resource per pair, based at frame A (uniform: SGPRs), stride 0, no clamp that matters; frame B is the
      // same resource at soffset = one hop.  The instruction offset reaches 4095 bytes, so samples 8 x 256 and up take
      // 4096 more through soffset.  Values arrive as (float)s16; the registers are the integer ones, reinterpreted.
      const uint64_t a = (uint64_t)(uintptr_t)p.a;
      rsrc_t rs;
      rs.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)a);
      rs.y = __builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32)) & 0xffff;
      rs.z = 0x7fffffff;
      rs.w = 4 | (5 << 3) | (6 << 6) | (7 << 9) | (3 << 12) | (2 << 15);
      const int b_lo = __builtin_amdgcn_readfirstlane(p.has_b ? kHop * 2 : 0), b_hi = b_lo + 4096, a_hi = 4096;
#pragma unroll
      for (int k = 0; k < 8; k++) {
        asm volatile("tbuffer_load_format_x %0, %1, %2, 0 format:[BUF_DATA_FORMAT_16,BUF_NUM_FORMAT_SSCALED] offen offset:%3"
                     : "=v"(ra[k]) : "v"(voff), "s"(rs), "n"(512 * k) : "memory");
        asm volatile("tbuffer_load_format_x %0, %1, %2, %4 format:[BUF_DATA_FORMAT_16,BUF_NUM_FORMAT_SSCALED] offen offset:%3"
                     : "=v"(rb[k]) : "v"(voff), "s"(rs), "n"(512 * k), "s"(b_lo) : "memory");
      }
#pragma unroll
      for (int k = 0; k < 8; k++) {
        asm volatile("tbuffer_load_format_x %0, %1, %2, %4 format:[BUF_DATA_FORMAT_16,BUF_NUM_FORMAT_SSCALED] offen offset:%3"
                     : "=v"(ra[8 + k]) : "v"(voff), "s"(rs), "n"(512 * k), "s"(a_hi) : "memory");
        asm volatile("tbuffer_load_format_x %0, %1, %2, %4 format:[BUF_DATA_FORMAT_16,BUF_NUM_FORMAT_SSCALED] offen offset:%3"
                     : "=v"(rb[8 + k]) : "v"(voff), "s"(rs), "n"(512 * k), "s"(b_hi) : "memory");
      }
      return;
    }
    const raw_t *qa = reinterpret_cast<const raw_t *>(p.a), *qb = reinterpret_cast<const raw_t *>(p.b);
    int tt = t;
    asm volatile("" : "+v"(tt));  // an opaque INDEX keeps these global (not flat) loads inside the loop (stft_kernel.h)
#pragma unroll
    for (int k = 0; k < 16; k++) {
      ra[k] = (reg_t)qa[tt + 256 * k];
      rb[k] = (reg_t)qb[tt + 256 * k];
    }
  };
  // the asm loads are invisible to the compiler's wait-count bookkeeping: their consumer waits for them itself
  auto loads_landed = [&]() {
    if (!kFormatLoad) return;
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(ra[0]), "+v"(ra[1]), "+v"(ra[2]), "+v"(ra[3]), "+v"(ra[4]), "+v"(ra[5]), "+v"(ra[6]), "+v"(ra[7]), "+v"(ra[8]),
                   "+v"(ra[9]), "+v"(ra[10]), "+v"(ra[11]), "+v"(ra[12]), "+v"(ra[13]), "+v"(ra[14]), "+v"(ra[15])
                 :
                 : "memory");
    asm volatile(""
                 : "+v"(rb[0]), "+v"(rb[1]), "+v"(rb[2]), "+v"(rb[3]), "+v"(rb[4]), "+v"(rb[5]), "+v"(rb[6]), "+v"(rb[7]), "+v"(rb[8]),
                   "+v"(rb[9]), "+v"(rb[10]), "+v"(rb[11]), "+v"(rb[12]), "+v"(rb[13]), "+v"(rb[14]), "+v"(rb[15])
                 :
                 : "memory");
  };
  constexpr int kFoldHalf = core::kClassLaneMax / 2;
  auto fold_tree_store = [&](cf acc, const PairSrc &p, int tt, bool store) {
    // fixed-order tree over the 16 lanes of the row (fp_core.h class_tree_partner)
    acc = cf{acc.x + dpp_f32<0xB1>(acc.x), acc.y + dpp_f32<0xB1>(acc.y)};    // quad_perm [1,0,3,2]
    acc = cf{acc.x + dpp_f32<0x4E>(acc.x), acc.y + dpp_f32<0x4E>(acc.y)};    // quad_perm [2,3,0,1]
    acc = cf{acc.x + dpp_f32<0x141>(acc.x), acc.y + dpp_f32<0x141>(acc.y)};  // row_half_mirror
    acc = cf{acc.x + dpp_f32<0x140>(acc.x), acc.y + dpp_f32<0x140>(acc.y)};  // row_mirror
    if (store && (tt & 15) == 0) {
      const uint32_t c = (uint32_t)tt >> 4;
      if (c < (uint32_t)kBands) {
        double *out = chroma + p.row * kBands;
        out[c] = (double)acc.x;
        if (p.has_b) out[kBands + c] = (double)acc.y;
      } else if (!(LAB & kLab32Clock)) {  // the fourth wave: rows 12..15 hold the four partial sums of the frames' energy
        // BOTH frames get the energy of the PAIR: they went through one complex transform, and what separates them
        // afterwards (X_a = (Z_k + conj Z_{N-k}) / 2, X_b = (Z_k - conj Z_{N-k}) / 2i) leaves an error of u |Z| in each --
        // a frame of three samples of an onset beside a full-scale partner carries the PARTNER's noise floor.  Found by
        // tools/fuzz_cert_adversarial.py (round 5): with the frame's own energy such an item was accepted at 75 S.
        float *out = energy + p.row * kEnergyParts;
        const float e = p.has_b ? acc.x + acc.y : acc.x;
        out[c - kBands] = e;
        if (p.has_b) out[kEnergyParts + c - kBands] = e;
      }
    }
  };
  PairSrc cur = locate(first), prev = cur;
  issue_loads(cur);
#pragma unroll
  for (int k = 0; k < 16; k++) asm volatile("" : "+v"(ra[k]), "+v"(rb[k]));  // keep the sign extension with the loads
  uint64_t stamp_core = 0, stamp_real = 0;
  if (LAB & kLab32Clock) {
    stamp_core = __builtin_amdgcn_s_memtime();
    stamp_real = __builtin_amdgcn_s_memrealtime();
  }

  for (uint32_t g = first; g < last; g++) {
    int tt = t;
    asm volatile("" : "+v"(tt));
    cf r[16];
    float ea = 0.0f, eb = 0.0f;
    loads_landed();
    auto convert = [&](int k) {
      int sa, sb;
      if (CH == 1) {
        sa = ra[k];
        sb = rb[k];
      } else {  // AudioProcessor::LoadStereo: (L + R) / 2, C truncation
        sa = ((int)(int16_t)ra[k] + (ra[k] >> 16)) / 2;
        sb = ((int)(int16_t)rb[k] + (rb[k] >> 16)) / 2;
      }
      const float w = (LAB & kLab32WinLoad) ? win32[tt + 256 * k] : win[k];
      const float fa = kFormatLoad ? __int_as_float(sa) : (float)sa, fb = kFormatLoad ? __int_as_float(sb) : (float)sb;
      const cf x = (LAB & kLab32ConsumerTw) ? cf{fa, fb} : cf{fa * w, fb * w};
      r[k] = x;
      if (!(LAB & kLab32NoEnergy)) {
        ea = core::fmad(x.x, x.x, ea);
        eb = core::fmad(x.y, x.y, eb);
      }
    };
    if (LAB & kLab32NoFold) {
#pragma unroll
      for (int k = 0; k < 16; k++) convert(k);
    } else {
      cf fv[kFoldHalf];
      core::class_lane_load_part<0, kFoldHalf>(lds, fold_entry, fv);
#pragma unroll
      for (int k = 0; k < 8; k++) convert(k);
      cf acc = fv[0];
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
      for (int k = 0; k < 16; k++) r[k].y = 0.0f;
      eb = 0.0f;
    }
    if (!(LAB & kLab32NoB1)) lds_barrier(); else wave_lds_fence();  // every thread has read its share of the previous pair's powers and energy partials
    if (LAB & kLab32ConsumerTw) {
      cf x[16];
#pragma unroll
      for (int k = 0; k < 16; k++) x[k] = r[k];
      float w4[4];
#pragma unroll
      for (int n2 = 0; n2 < 4; n2++) {
#pragma unroll
        for (int q = 0; q < 4; q++) w4[q] = win[4 * q + n2];
        if (n2 == 0) core::head_col_win<0>(r, w4, x[0], x[4], x[8], x[12]);
        if (n2 == 1) core::head_col_win<1>(r, w4, x[1], x[5], x[9], x[13]);
        if (n2 == 2) core::head_col_win<2>(r, w4, x[2], x[6], x[10], x[14]);
        if (n2 == 3) core::head_col_win<3>(r, w4, x[3], x[7], x[11], x[15]);
      }
      core::dif_tails_store_raw(core::dif0_base(tt), 272, lds, r);
    } else {
      core::dif0_streamed_pw<0>(tt, pw0, lds, r);
    }
    if (!(LAB & kLab32NoB2)) lds_barrier(); else wave_lds_fence();  // stage 0 -> 1 crosses waves
    if (LAB & kLab32ConsumerTw) {
      core::dif1_consumer(tt, row1, lds, r);
      wave_lds_fence();
      core::dif2_consumer(tt, row2, lds, r);
    } else if (kAsmReads) {  // the product: single ds_read_b64 (0.4549 against 0.4621 ms in tools/stft32_lab, same bits)
      lds_read16_single<17>(lds, core::dif1_base(tt), r);
      core::fft16_head(r);
      core::dif_tails_store<0>(core::dif1_base(tt), 17, pw1, lds, r);
      wave_lds_fence();
      lds_read16_single<1>(lds, core::dif2_base(tt), r);
      core::dif2_from_registers<0>(tt, lds, r);
    } else if (LAB & kLab32Tw1Lds) {
      cf tw1[16];
#pragma unroll
      for (int j = 1; j < 16; j++) tw1[j] = core::lds_get(lds, kTw1Base + 16 * (j - 1) + (tt & 15));
      core::dif1_streamed_pw<0>(tt, tw1, lds, r);
    } else {
      core::dif1_streamed_pw<0>(tt, pw1, lds, r);
    }
    if (!kAsmReads && !(LAB & kLab32ConsumerTw)) {
      wave_lds_fence();  // stage 1 -> 2 stays inside 16 consecutive lanes
      core::dif2_streamed<0>(tt, lds, r);
    }
    wave_lds_fence();  // publish -> partner reads stays inside the wave (fp_core.h group_k0)

    if (!(LAB & kLab32NoPower)) {
      cf yp[core::kBinsPerThread];
      core::dif_partner_load(tt, lds, yp);
      float pwa[core::kBinsPerThread], pwb[core::kBinsPerThread];
#pragma unroll
      for (int j = 0; j < core::kBinsPerThread; j++) core::dif_power_of(r[core::out16(j)], yp[j], &pwa[j], &pwb[j]);
#pragma unroll
      for (int j = 0; j < core::kBinsPerThread; j++)
        if (LAB & kLab32NoConflict)
          core::lds_put(lds, core::dif2_base(tt) + j, cf{pwa[j], pwb[j]});
        else
          core::lds_put_bytes(lds, (j & 1) ? core::slot_bytes<1>(inv.w[j >> 1]) : core::slot_bytes<0>(inv.w[j >> 1]), cf{pwa[j], pwb[j]});
    } else {
      asm volatile("" ::"v"(r[0].x), "v"(r[5].y), "v"(r[10].x), "v"(r[15].y));
    }
    if (!(LAB & kLab32NoEnergy)) core::lds_put(lds, core::energy_slot(tt), cf{ea, eb});  // spare column 8 of this thread's own stage-2 row
    const PairSrc nxt = locate(min(g + 1, last - 1));  // last pair: harmless re-read
    issue_loads(nxt);
    if (!(LAB & kLab32NoB3)) lds_barrier(); else wave_lds_fence();  // the power image is complete
    prev = cur;
    cur = nxt;
  }
  if (LAB & kLab32Clock) {  // the stamps replace the energy output of this diagnostic build
    if (t == 0) {
      uint64_t *o = reinterpret_cast<uint64_t *>(energy) + 2 * (size_t)blockIdx.x;
      o[0] = __builtin_amdgcn_s_memtime() - stamp_core;
      o[1] = __builtin_amdgcn_s_memrealtime() - stamp_real;
    }
    return;
  }
  {
    cf fv[core::kClassLaneMax];
    core::class_lane_load(lds, fold_entry, fv);
    fold_tree_store(core::class_lane_add(fv), prev, t, true);
  }
}

}  // namespace stft
}  // namespace needle
