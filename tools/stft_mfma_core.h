// The transform core of tools/stft_mfma_lab.hip (laboratory, not part of the product): one 4096-point complex transform per
// frame pair (z = frame A + i frame B) as three radix-16 stages on the MATRIX pipe.
//
//   n = 256 n0 + 16 n1 + n2,  k = k0 + 16 k1 + 256 k2,  W = exp(-2 pi i / 4096)
//   stage 0 over n0 -> twiddle W^((16 n1 + n2) k0) / 16 -> stage 1 over n1 -> twiddle W^(16 n2 k1) / 16 -> stage 2 over n2
//
// A stage is Y[32 x 256] = F[32 x 32] X[32 x 256]: the 16 complex inputs of a butterfly as 32 reals down a column, 256
// butterflies side by side, F = [[C, S], [-S, C]] of the 16-point DFT.  v_mfma_f32_32x32x16_f16 multiplies f16 operands
// exactly and accumulates in f32, so an f32 value goes in as x = hi + lo (two f16, 22 bits) and F as Fh + Fl:
// Y = Fh lo + Fl hi + Fh hi (Fl lo is 2^-22 of the result and dropped): 6 instructions per block of 32 columns and stage.
// A wave owns two blocks; F is the A operand (rows = output components), the data the B operand (a lane holds its column's
// K = 8 (l >> 5) + j), so a lane ends with whole complex outputs of its column in its accumulator registers:
//   row r = 8 g + 4 (l >> 5) + 2 p + part  <->  output point m = 4 g + 2 (l >> 5) + p;  K index 16 q + 8 (l >> 5) + j  <->
//   input point 8 q + 4 (l >> 5) + (j >> 1), part j & 1.
// Samples enter as s16 x window WITHOUT the 1 / 32767 (|x| <= 32768 stays inside f16's range); the 2^-4 per stage keeps the
// modulus under 46 341 all the way (the last stage's output is f32 and unscaled: spectrum x 2^-8).
//
// Exchange image in LDS (36 864 bytes): 256 column slots of 36 words; a slot holds its column's 16 K-points as four groups
// of [hi x 4 | lo x 4] words (a word = the f16 pair re, im), so a consumer reads an operand half with ONE ds_read_b128 and
// consecutive lanes (pitch 36) tile the 64 banks.  Stage 0 -> 1 crosses waves (barrier); stage 1 -> 2 stays inside the
// wave (slot = 16 k0 + ..., k0 = 4 wave + ...: a wave rewrites its own 64 slots in place).
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace mfmalab {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
struct cf {
  float x, y;
};
struct Stream {
  uint64_t pcm_off;
  uint32_t frames, pair_base;
};
struct Tables {
  const u32x4 *afrag;  // [hi, lo][chunk][lane]
  const cf *tw0;       // [column 0..255][m]  W^(b m) / 16
  const cf *tw1;       // [n2][m]            W^(16 n2 m) / 16
  const float *win;    // Hamming, unscaled
};

constexpr int kN = 4096, kHop = 1365;
constexpr int kPitch = 36;
constexpr size_t kLdsBytes = 256 * kPitch * 4;
enum : int { kVerify = 1, kNoMfma = 2, kNoLds = 4, kNoEpilogue = 8, kNoBarrier = 16, kNoLoads = 32, kStagger = 64 };

struct HostTables {
  std::vector<uint32_t> afrag;
  std::vector<cf> tw0, tw1;
  std::vector<float> win;
};

inline uint16_t half_bits(double v) {
  _Float16 h = (_Float16)v;
  uint16_t u;
  std::memcpy(&u, &h, 2);
  return u;
}

inline void build_tables(HostTables *t) {
  const long double pi = 3.14159265358979323846264338327950288L;
  t->afrag.assign(2 * 2 * 64 * 4, 0u);
  for (int q = 0; q < 2; q++)
    for (int l = 0; l < 64; l++)
      for (int j = 0; j < 8; j++) {
        const int r = l & 31, hh = l >> 5;
        const int n = 8 * q + 4 * hh + (j >> 1), part = j & 1;
        const int m = 4 * (r >> 3) + 2 * ((r >> 2) & 1) + ((r >> 1) & 1), po = r & 1;
        const int e = (m * n) & 15;
        long double c = cosl(2 * pi * e / 16), s = sinl(2 * pi * e / 16);
        if (e % 4 == 0) {  // exact 0 / +-1
          c = (e == 0) ? 1 : (e == 8) ? -1 : 0;
          s = (e == 4) ? 1 : (e == 12) ? -1 : 0;
        }
        const double val = (double)(po == 0 ? (part == 0 ? c : s) : (part == 0 ? -s : c));
        const uint16_t hi = half_bits(val);
        _Float16 hf;
        std::memcpy(&hf, &hi, 2);
        const uint16_t lo = half_bits(val - (double)hf);
        const int shift = 16 * (j & 1);
        t->afrag[((0 * 2 + q) * 64 + l) * 4 + (j >> 1)] |= (uint32_t)hi << shift;
        t->afrag[((1 * 2 + q) * 64 + l) * 4 + (j >> 1)] |= (uint32_t)lo << shift;
      }
  t->tw0.resize(256 * 16);
  t->tw1.resize(16 * 16);
  for (int b = 0; b < 256; b++)
    for (int m = 0; m < 16; m++) {
      const long double a = -2 * pi * ((b * m) & 4095) / 4096;
      t->tw0[b * 16 + m] = cf{(float)(cosl(a) / 16), (float)(sinl(a) / 16)};
    }
  for (int n2 = 0; n2 < 16; n2++)
    for (int m = 0; m < 16; m++) {
      const long double a = -2 * pi * ((16 * n2 * m) & 4095) / 4096;
      t->tw1[n2 * 16 + m] = cf{(float)(cosl(a) / 16), (float)(sinl(a) / 16)};
    }
  t->win.resize(kN);
  for (int i = 0; i < kN; i++) t->win[i] = (float)(0.54L - 0.46L * cosl(2 * pi * i / (kN - 1)));
}

// x = hi + lo as two f16 pairs (re in the low half): v_cvt_pk_f16_f32 rounds to nearest even; v_fma_mix{lo,hi}_f16 take the
// f16 half back as f32, subtract it from x in f32 (exact) and round the difference to f16 -- three instructions per complex
// value.  hipcc neither schedules nor pads what is inside an asm statement (cdna_hip_programming.md section 5.7): a VGPR
// written here and read as an MFMA operand needs two wait states in between, which the LAST split in front of a group of
// MFMAs carries itself (the statements are volatile, so that one stays last).
template <bool LAST>
__device__ __forceinline__ void split(float re, float im, uint32_t &hi, uint32_t &lo) {
  if (LAST)
    asm volatile("v_cvt_pk_f16_f32 %0, %2, %3\n\tv_fma_mixlo_f16 %1, %0, -1.0, %2 op_sel_hi:[1,0,0]\n\t"
                 "v_fma_mixhi_f16 %1, %0, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\ts_nop 1"
                 : "=&v"(hi), "=&v"(lo) : "v"(re), "v"(im));
  else
    asm volatile("v_cvt_pk_f16_f32 %0, %2, %3\n\tv_fma_mixlo_f16 %1, %0, -1.0, %2 op_sel_hi:[1,0,0]\n\t"
                 "v_fma_mixhi_f16 %1, %0, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
                 : "=&v"(hi), "=&v"(lo) : "v"(re), "v"(im));
}

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ half8 as_half8(u32x4 v) { return __builtin_bit_cast(half8, v); }

struct Consts {
  half8 ahi[2], alo[2];
};

// the six products of one block of 32 columns: lo terms first, the large one last
template <int LAB>
__device__ __forceinline__ f32x16 stage_products(const Consts &k, const u32x4 bh[2], const u32x4 bl[2]) {
  f32x16 acc = {};
  if (LAB & kNoMfma) {  // timing only: keeps the operands alive and the accumulator defined
#pragma unroll
    for (int i = 0; i < 4; i++) {
      acc[i] = __builtin_bit_cast(float, bh[0][i]);
      acc[4 + i] = __builtin_bit_cast(float, bh[1][i]);
      acc[8 + i] = __builtin_bit_cast(float, bl[0][i]);
      acc[12 + i] = __builtin_bit_cast(float, bl[1][i]);
    }
    return acc;
  }
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(k.ahi[0], as_half8(bl[0]), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(k.ahi[1], as_half8(bl[1]), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(k.alo[0], as_half8(bh[0]), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(k.alo[1], as_half8(bh[1]), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(k.ahi[0], as_half8(bh[0]), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(k.ahi[1], as_half8(bh[1]), acc, 0, 0, 0);
  return acc;
}

// a stage's outputs of one block: twiddle, split, into the exchange image at base + OFF(g, p) (words)
template <int LAB, int STRIDE_G, int STRIDE_P>
__device__ __forceinline__ void stage_outputs(const f32x16 &acc, const cf *tw, uint32_t *lds, int base) {
#pragma unroll
  for (int g = 0; g < 4; g++)
#pragma unroll
    for (int p = 0; p < 2; p++) {
      const float yr = acc[4 * g + 2 * p], yi = acc[4 * g + 2 * p + 1];
      uint32_t hi, lo;
      if (LAB & kNoEpilogue) {
        hi = __builtin_bit_cast(uint32_t, yr);
        lo = __builtin_bit_cast(uint32_t, yi);
      } else {
        const cf t = tw[2 * g + p];
        const float re = __builtin_fmaf(yr, t.x, -(yi * t.y)), im = __builtin_fmaf(yr, t.y, yi * t.x);
        split<false>(re, im, hi, lo);
      }
      if (!(LAB & kNoLds)) {
        lds[base + STRIDE_G * g + STRIDE_P * p] = hi;
        lds[base + STRIDE_G * g + STRIDE_P * p + 4] = lo;
      } else {
        asm volatile("" ::"v"(hi), "v"(lo));
      }
    }
}

template <int LAB>
__device__ __forceinline__ void stage_inputs(const uint32_t *lds, int base, u32x4 bh[2], u32x4 bl[2]) {
  if (LAB & kNoLds) {
    const uint32_t v = (uint32_t)base;
    bh[0] = u32x4{v, v, v, v};
    bh[1] = bl[0] = bl[1] = bh[0];
    asm volatile("" : "+v"(bh[0]), "+v"(bh[1]), "+v"(bl[0]), "+v"(bl[1]));
    return;
  }
#pragma unroll
  for (int q = 0; q < 2; q++) {
    bh[q] = *reinterpret_cast<const u32x4 *>(lds + base + 16 * q);
    bl[q] = *reinterpret_cast<const u32x4 *>(lds + base + 16 * q + 4);
  }
}

template <int WAVES_PER_SIMD, int LAB>
__global__ __launch_bounds__(256, WAVES_PER_SIMD) void stft_mfma_core_kernel(const int16_t *__restrict__ pcm, const Stream *__restrict__ streams,
                                                                            int num_streams, const Tables tab, uint32_t first_pair,
                                                                            uint32_t pairs, uint32_t pairs_per_block, float *__restrict__ out,
                                                                            cf *__restrict__ zout) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  const int t = threadIdx.x, w = t >> 6, l = t & 63, hh = l >> 5, c = l & 31;
  const uint32_t first = first_pair + blockIdx.x * pairs_per_block, last = min(first + pairs_per_block, first_pair + pairs);
  if (first >= last) return;

  Consts k;
  k.ahi[0] = as_half8(tab.afrag[(0 * 2 + 0) * 64 + l]);
  k.ahi[1] = as_half8(tab.afrag[(0 * 2 + 1) * 64 + l]);
  k.alo[0] = as_half8(tab.afrag[(1 * 2 + 0) * 64 + l]);
  k.alo[1] = as_half8(tab.afrag[(1 * 2 + 1) * 64 + l]);
  cf tw0[2][8], tw1[8];
  float win[2][8];
#pragma unroll
  for (int g = 0; g < 4; g++)
#pragma unroll
    for (int p = 0; p < 2; p++) {
      const int m = 4 * g + 2 * hh + p;
      tw1[2 * g + p] = tab.tw1[(c & 15) * 16 + m];
#pragma unroll
      for (int blk = 0; blk < 2; blk++) tw0[blk][2 * g + p] = tab.tw0[(64 * w + 32 * blk + c) * 16 + m];
    }
#pragma unroll
  for (int blk = 0; blk < 2; blk++)
#pragma unroll
    for (int i = 0; i < 8; i++) win[blk][i] = tab.win[256 * (8 * (i >> 2) + 4 * hh + (i & 3)) + 64 * w + 32 * blk + c];

  // word addresses in the exchange image (header comment)
  const int base0 = (c & 15) * kPitch + 8 * w + (c >> 4) + 32 * kPitch * hh;                       // + 2 blk + 64 pitch g + 16 pitch p
  const int base1 = (64 * w + c) * kPitch + 8 * hh;                                                 // + 32 pitch blk + 16 q
  const int base2 = (64 * w + 16 * (c >> 4) + 2 * hh) * kPitch + 8 * ((c >> 2) & 3) + (c & 3);      // + pitch (32 blk + 4 g + p)

  // streams
  int si = 0;
  {
    int lo = 0, hi = num_streams - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (streams[mid].pair_base <= first) lo = mid; else hi = mid - 1;
    }
    si = lo;
  }
  Stream st = streams[si];
  uint32_t st_end = st.pair_base + (st.frames + 1) / 2;
  struct Src {
    const int16_t *a, *b;
    bool has_b;
  };
  auto locate = [&](uint32_t g) {
    while (g >= st_end) {
      st = streams[++si];
      st_end = st.pair_base + (st.frames + 1) / 2;
    }
    const uint32_t fa = 2 * (g - st.pair_base);
    Src s;
    s.has_b = fa + 1 < st.frames;
    s.a = pcm + st.pcm_off + (uint64_t)fa * kHop;
    s.b = s.has_b ? s.a + kHop : s.a;
    return s;
  };
  // A lane's 32 samples of a pair at a wave-uniform base + lane offset + immediate (global_load_sshort: sign-extended).
  // (Packing frame A / frame B into the halves of one register with global_load_short_d16 / _d16_hi does not work here:
  // with SRAM ECC on -- gfx950:sramecc+ -- a d16 load writes the whole register and ZEROES the other half.)
  // The loads are invisible to hipcc's wait counting: loads_landed() before the first use (section 5.7 form ii).
  int sa[2][8], sb[2][8];
  const uint32_t voff = 2u * (uint32_t)(64 * w + c + 1024 * hh);  // bytes; + 64 blk + 512 i (q = 0), + 4096 (q = 1)
  const uint32_t voff1 = voff + 4096u;
  auto issue_loads = [&](const Src &s) {
    if (LAB & kNoLoads) return;
    const int16_t *pa = s.a, *pb = s.b;
#pragma unroll
    for (int blk = 0; blk < 2; blk++)
#pragma unroll
      for (int i = 0; i < 8; i++) {
        if (i < 4) {
          asm volatile("global_load_sshort %0, %1, %2 offset:%3" : "=v"(sa[blk][i]) : "v"(voff), "s"(pa), "n"(64 * blk + 512 * (i & 3)) : "memory");
          asm volatile("global_load_sshort %0, %1, %2 offset:%3" : "=v"(sb[blk][i]) : "v"(voff), "s"(pb), "n"(64 * blk + 512 * (i & 3)) : "memory");
        } else {
          asm volatile("global_load_sshort %0, %1, %2 offset:%3" : "=v"(sa[blk][i]) : "v"(voff1), "s"(pa), "n"(64 * blk + 512 * (i & 3)) : "memory");
          asm volatile("global_load_sshort %0, %1, %2 offset:%3" : "=v"(sb[blk][i]) : "v"(voff1), "s"(pb), "n"(64 * blk + 512 * (i & 3)) : "memory");
        }
      }
  };
  auto loads_landed = [&]() {
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(sa[0][0]), "+v"(sa[0][1]), "+v"(sa[0][2]), "+v"(sa[0][3]), "+v"(sa[0][4]), "+v"(sa[0][5]), "+v"(sa[0][6]), "+v"(sa[0][7]),
                   "+v"(sa[1][0]), "+v"(sa[1][1]), "+v"(sa[1][2]), "+v"(sa[1][3]), "+v"(sa[1][4]), "+v"(sa[1][5]), "+v"(sa[1][6]), "+v"(sa[1][7])
                 :
                 : "memory");
    asm volatile(""
                 : "+v"(sb[0][0]), "+v"(sb[0][1]), "+v"(sb[0][2]), "+v"(sb[0][3]), "+v"(sb[0][4]), "+v"(sb[0][5]), "+v"(sb[0][6]), "+v"(sb[0][7]),
                   "+v"(sb[1][0]), "+v"(sb[1][1]), "+v"(sb[1][2]), "+v"(sb[1][3]), "+v"(sb[1][4]), "+v"(sb[1][5]), "+v"(sb[1][6]), "+v"(sb[1][7])
                 :
                 : "memory");
  };
#pragma unroll
  for (int blk = 0; blk < 2; blk++)
#pragma unroll
    for (int i = 0; i < 8; i++) sa[blk][i] = t + i, sb[blk][i] = t - i;
  Src cur = locate(first);
  issue_loads(cur);
  if (LAB & kStagger) {  // the first round of workgroups starts a third of a pair apart on each CU (blocks b, b + 256, b + 512 share one)
    const int third = (blockIdx.x >> 8) % 3;
    if (blockIdx.x < 768 && third != 0) {
      for (int i = 0; i < third; i++) __builtin_amdgcn_s_sleep(16);  // 16 x 64 cycles each
    }
  }
  float check = 0.0f;

  for (uint32_t g = first; g < last; g++) {
    // ---- stage 0: operands from registers
    f32x16 acc[2];
    loads_landed();
    if (!cur.has_b) {  // the stream's last pair has no frame B (uniform, rare): its samples become zeros
#pragma unroll
      for (int blk = 0; blk < 2; blk++)
#pragma unroll
        for (int i = 0; i < 8; i++) sb[blk][i] = 0;
    }
#pragma unroll
    for (int blk = 0; blk < 2; blk++) {
      uint32_t h[8], lw[8];
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const float fa = (float)sa[blk][i], fb = (float)sb[blk][i];
        const float re = fa * win[blk][i], im = fb * win[blk][i];
        if (i == 7) split<true>(re, im, h[i], lw[i]); else split<false>(re, im, h[i], lw[i]);
      }
      const u32x4 bh[2] = {u32x4{h[0], h[1], h[2], h[3]}, u32x4{h[4], h[5], h[6], h[7]}};
      const u32x4 bl[2] = {u32x4{lw[0], lw[1], lw[2], lw[3]}, u32x4{lw[4], lw[5], lw[6], lw[7]}};
      acc[blk] = stage_products<LAB>(k, bh, bl);
    }
    const Src nxt = locate(min(g + 1, last - 1));  // last pair: harmless re-read
    issue_loads(nxt);
    if (!(LAB & kNoBarrier)) lds_barrier(); else wave_lds_fence();  // every wave has read the previous pair's image to the end
#pragma unroll
    for (int blk = 0; blk < 2; blk++) stage_outputs<LAB, 64 * kPitch, 16 * kPitch>(acc[blk], tw0[blk], lds, base0 + 2 * blk);
    if (!(LAB & kNoBarrier)) lds_barrier(); else wave_lds_fence();  // stage 0 -> 1 crosses waves

    // ---- stage 1
    {
      u32x4 bh[2][2], bl[2][2];
#pragma unroll
      for (int blk = 0; blk < 2; blk++) stage_inputs<LAB>(lds, base1 + 32 * kPitch * blk, bh[blk], bl[blk]);
      if (!(LAB & kNoLds)) wave_lds_fence();
#pragma unroll
      for (int blk = 0; blk < 2; blk++) acc[blk] = stage_products<LAB>(k, bh[blk], bl[blk]);
    }
#pragma unroll
    for (int blk = 0; blk < 2; blk++) stage_outputs<LAB, 4 * kPitch, kPitch>(acc[blk], tw1, lds, base2 + 32 * kPitch * blk);
    wave_lds_fence();  // stage 1 -> 2 stays inside the wave

    // ---- stage 2
    {
      u32x4 bh[2][2], bl[2][2];
#pragma unroll
      for (int blk = 0; blk < 2; blk++) stage_inputs<LAB>(lds, base1 + 32 * kPitch * blk, bh[blk], bl[blk]);
#pragma unroll
      for (int blk = 0; blk < 2; blk++) acc[blk] = stage_products<LAB>(k, bh[blk], bl[blk]);
    }
    if (LAB & kVerify) {
#pragma unroll
      for (int blk = 0; blk < 2; blk++)
#pragma unroll
        for (int gg = 0; gg < 4; gg++)
#pragma unroll
          for (int p = 0; p < 2; p++) {
            const int m = 4 * gg + 2 * hh + p;
            const int kk = 256 * m + 16 * (c & 15) + 4 * w + 2 * blk + (c >> 4);
            zout[(size_t)(g - first_pair) * kN + kk] = cf{acc[blk][4 * gg + 2 * p], acc[blk][4 * gg + 2 * p + 1]};
          }
    } else {
      check += acc[0][0] + acc[1][5];
      asm volatile("" ::"v"(acc[0]), "v"(acc[1]));
    }
    cur = nxt;
  }
  if (!(LAB & kVerify)) out[(size_t)(blockIdx.x & 4095) * 256 + t] = check;
}

}  // namespace mfmalab
