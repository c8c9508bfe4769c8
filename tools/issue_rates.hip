// Diagnostic (not part of the product): what one SIMD of gfx950 sustains, in shader cycles per wave-instruction, on
// each instruction class of stft_chroma32_kernel's loop body, at the kernel's own occupancy (3 workgroups of 256
// threads per CU = 3 waves per SIMD) and at 1 wave per SIMD; plus mixes in the kernel's proportions, to see which costs
// ADD on a SIMD and which overlap.  Cycles come from s_memtime around the loop (the shader clock itself, whatever
// frequency the device runs at); a workgroup's figure is that of its first wave, the table prints the median workgroup.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/issue_rates.hip -o tools/issue_rates
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

#define REP4(x) x x x x
#define REP8(x) REP4(x) REP4(x)
#define REP16(x) REP8(x) REP8(x)

enum Op {
  kFmac, kAdd, kMul, kFmamk, kMulE64Neg, kPkAdd, kPkFma, kCvt, kAddDpp, kFmacDep1, kFmacDep2, kFmacBank,
  kFmacSame, kFmacSgpr, kAddNoAcc, kXor, kBcnt, kCmpSgpr, kMax3, kCell3, kCndmask,
  kDsRead, kDsRead2, kDsWrite, kDsWrite2, kLoadShort, kMixValuLds, kMixValuLdsVmem, kMixValuOnly, kNumOps
};
static const char *kNames[kNumOps] = {
    "v_fmac_f32_e32 (8 independent)", "v_add_f32_e32", "v_mul_f32_e32", "v_fmamk_f32 (32-bit literal)",
    "v_mul_f32_e64 (neg modifier)", "v_pk_add_f32", "v_pk_fma_f32", "v_cvt_f32_i32_e32", "v_add_f32_dpp row_mirror",
    "v_fmac_f32 dependent, distance 1", "v_fmac_f32 dependent, distance 2", "v_fmac_f32, three sources in one VGPR bank",
    "v_fmac_f32 a_i, x, x (one VGPR read twice)", "v_fmac_f32 a_i, s0, x (SGPR multiplicand)", "v_add_f32 a_i, x, y (no accumulator read)",
    "v_xor_b32", "v_bcnt_u32_b32", "v_cmp_le_u32_e64 -> SGPR pair", "v_max3_u32", "scan cell: v_xor + v_bcnt + v_cmp (+ s_and)", "v_cndmask_b32 (VCC)",
    "ds_read_b64 (conflict-free)", "ds_read2_b64", "ds_write_b64", "ds_write2_b64", "global_load_sshort (L2 hits)",
    "mix: 16 VALU + 1 ds_write_b64 + 1 ds_read_b64", "mix: + 0.7 global_load_sshort per 16 VALU", "mix: the 16 VALU alone"};
// wave-instructions per iteration of each loop below
static const int kPerIter[kNumOps] = {128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 384, 128, 64, 64, 64, 64, 64, 144, 150, 128};

template <int OP>
__global__ __launch_bounds__(256) void rate_kernel(unsigned long long *cycles, float *sink, const short *pcm, int iters) {
  extern __shared__ float lds[];
  const int t = threadIdx.x;
  float a0 = t, a1 = t + 1, a2 = t + 2, a3 = t + 3, a4 = t + 4, a5 = t + 5, a6 = t + 6, a7 = t + 7;
  float x = 1.0000001f, y = 0.9999999f;
  typedef float v2f __attribute__((ext_vector_type(2)));
  v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, px = {x, y};
  typedef float v4f __attribute__((ext_vector_type(4)));
  v4f q0 = {0, 0, 0, 0}, q1 = q0, q2 = q0, q3 = q0;
  int i0 = t;
  asm volatile("" : "+v"(x), "+v"(y), "+v"(i0), "+v"(px));
  const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float *)lds + (unsigned)t * 8u;  // b64 per lane: conflict-free
  const short *q = pcm + t;
  for (int i = t; i < 8192; i += 256) lds[i] = (float)i;
  __syncthreads();
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
    if (OP == kFmac) {
      REP16(asm volatile("v_fmac_f32_e32 %0, %8, %9\n v_fmac_f32_e32 %1, %8, %9\n v_fmac_f32_e32 %2, %8, %9\n v_fmac_f32_e32 %3, %8, %9\n"
                         "v_fmac_f32_e32 %4, %8, %9\n v_fmac_f32_e32 %5, %8, %9\n v_fmac_f32_e32 %6, %8, %9\n v_fmac_f32_e32 %7, %8, %9"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));)
    } else if (OP == kAdd) {
      REP16(asm volatile("v_add_f32_e32 %0, %8, %0\n v_add_f32_e32 %1, %8, %1\n v_add_f32_e32 %2, %8, %2\n v_add_f32_e32 %3, %8, %3\n"
                         "v_add_f32_e32 %4, %8, %4\n v_add_f32_e32 %5, %8, %5\n v_add_f32_e32 %6, %8, %6\n v_add_f32_e32 %7, %8, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x));)
    } else if (OP == kMul) {
      REP16(asm volatile("v_mul_f32_e32 %0, %8, %0\n v_mul_f32_e32 %1, %8, %1\n v_mul_f32_e32 %2, %8, %2\n v_mul_f32_e32 %3, %8, %3\n"
                         "v_mul_f32_e32 %4, %8, %4\n v_mul_f32_e32 %5, %8, %5\n v_mul_f32_e32 %6, %8, %6\n v_mul_f32_e32 %7, %8, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x));)
    } else if (OP == kFmamk) {
      REP16(asm volatile("v_fmamk_f32 %0, %8, 0x3f6c835e, %0\n v_fmamk_f32 %1, %8, 0x3f6c835e, %1\n v_fmamk_f32 %2, %8, 0x3f6c835e, %2\n"
                         "v_fmamk_f32 %3, %8, 0x3f6c835e, %3\n v_fmamk_f32 %4, %8, 0x3f6c835e, %4\n v_fmamk_f32 %5, %8, 0x3f6c835e, %5\n"
                         "v_fmamk_f32 %6, %8, 0x3f6c835e, %6\n v_fmamk_f32 %7, %8, 0x3f6c835e, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x));)
    } else if (OP == kMulE64Neg) {
      REP16(asm volatile("v_mul_f32_e64 %0, -%8, %0\n v_mul_f32_e64 %1, -%8, %1\n v_mul_f32_e64 %2, -%8, %2\n v_mul_f32_e64 %3, -%8, %3\n"
                         "v_mul_f32_e64 %4, -%8, %4\n v_mul_f32_e64 %5, -%8, %5\n v_mul_f32_e64 %6, -%8, %6\n v_mul_f32_e64 %7, -%8, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x));)
    } else if (OP == kPkAdd) {
      REP16(asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                         "v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(px));)
    } else if (OP == kPkFma) {
      REP16(asm volatile("v_pk_fma_f32 %0, %4, %4, %0\n v_pk_fma_f32 %1, %4, %4, %1\n v_pk_fma_f32 %2, %4, %4, %2\n v_pk_fma_f32 %3, %4, %4, %3\n"
                         "v_pk_fma_f32 %0, %4, %4, %0\n v_pk_fma_f32 %1, %4, %4, %1\n v_pk_fma_f32 %2, %4, %4, %2\n v_pk_fma_f32 %3, %4, %4, %3"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(px));)
    } else if (OP == kCvt) {
      REP16(asm volatile("v_cvt_f32_i32_e32 %0, %8\n v_cvt_f32_i32_e32 %1, %8\n v_cvt_f32_i32_e32 %2, %8\n v_cvt_f32_i32_e32 %3, %8\n"
                         "v_cvt_f32_i32_e32 %4, %8\n v_cvt_f32_i32_e32 %5, %8\n v_cvt_f32_i32_e32 %6, %8\n v_cvt_f32_i32_e32 %7, %8"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(i0));)
    } else if (OP == kAddDpp) {
      REP16(asm volatile("v_add_f32_dpp %0, %8, %0 row_mirror row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %8, %1 row_mirror row_mask:0xf bank_mask:0xf\n"
                         "v_add_f32_dpp %2, %8, %2 row_mirror row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %8, %3 row_mirror row_mask:0xf bank_mask:0xf\n"
                         "v_add_f32_dpp %4, %8, %4 row_mirror row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %5, %8, %5 row_mirror row_mask:0xf bank_mask:0xf\n"
                         "v_add_f32_dpp %6, %8, %6 row_mirror row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %7, %8, %7 row_mirror row_mask:0xf bank_mask:0xf"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x));)
    } else if (OP == kFmacDep1) {
      REP16(asm volatile("v_fmac_f32_e32 %0, %1, %2\n v_fmac_f32_e32 %0, %1, %2\n v_fmac_f32_e32 %0, %1, %2\n v_fmac_f32_e32 %0, %1, %2\n"
                         "v_fmac_f32_e32 %0, %1, %2\n v_fmac_f32_e32 %0, %1, %2\n v_fmac_f32_e32 %0, %1, %2\n v_fmac_f32_e32 %0, %1, %2"
                         : "+v"(a0) : "v"(x), "v"(y));)
    } else if (OP == kFmacDep2) {
      REP16(asm volatile("v_fmac_f32_e32 %0, %2, %3\n v_fmac_f32_e32 %1, %2, %3\n v_fmac_f32_e32 %0, %2, %3\n v_fmac_f32_e32 %1, %2, %3\n"
                         "v_fmac_f32_e32 %0, %2, %3\n v_fmac_f32_e32 %1, %2, %3\n v_fmac_f32_e32 %0, %2, %3\n v_fmac_f32_e32 %1, %2, %3"
                         : "+v"(a0), "+v"(a1) : "v"(x), "v"(y));)
    } else if (OP == kFmacBank) {  // v4 += v8 * v12: all three in bank 0 (register number mod 4)
      REP16(asm volatile("v_fmac_f32_e32 v4, v8, v12\n v_fmac_f32_e32 v16, v8, v12\n v_fmac_f32_e32 v20, v8, v12\n v_fmac_f32_e32 v24, v8, v12\n"
                         "v_fmac_f32_e32 v28, v8, v12\n v_fmac_f32_e32 v32, v8, v12\n v_fmac_f32_e32 v36, v8, v12\n v_fmac_f32_e32 v40, v8, v12"
                         ::: "v4", "v8", "v12", "v16", "v20", "v24", "v28", "v32", "v36", "v40");)
    } else if (OP == kFmacSame) {
      REP16(asm volatile("v_fmac_f32_e32 %0, %8, %8\n v_fmac_f32_e32 %1, %8, %8\n v_fmac_f32_e32 %2, %8, %8\n v_fmac_f32_e32 %3, %8, %8\n"
                         "v_fmac_f32_e32 %4, %8, %8\n v_fmac_f32_e32 %5, %8, %8\n v_fmac_f32_e32 %6, %8, %8\n v_fmac_f32_e32 %7, %8, %8"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x));)
    } else if (OP == kFmacSgpr) {
      REP16(asm volatile("v_fmac_f32_e32 %0, %9, %8\n v_fmac_f32_e32 %1, %9, %8\n v_fmac_f32_e32 %2, %9, %8\n v_fmac_f32_e32 %3, %9, %8\n"
                         "v_fmac_f32_e32 %4, %9, %8\n v_fmac_f32_e32 %5, %9, %8\n v_fmac_f32_e32 %6, %9, %8\n v_fmac_f32_e32 %7, %9, %8"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "s"(iters));)
    } else if (OP == kAddNoAcc) {
      REP16(asm volatile("v_add_f32_e32 %0, %8, %9\n v_add_f32_e32 %1, %8, %9\n v_add_f32_e32 %2, %8, %9\n v_add_f32_e32 %3, %8, %9\n"
                         "v_add_f32_e32 %4, %8, %9\n v_add_f32_e32 %5, %8, %9\n v_add_f32_e32 %6, %8, %9\n v_add_f32_e32 %7, %8, %9"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));)
    } else if (OP == kXor) {
      REP16(asm volatile("v_xor_b32 %0, %8, %0\n v_xor_b32 %1, %8, %1\n v_xor_b32 %2, %8, %2\n v_xor_b32 %3, %8, %3\n"
                         "v_xor_b32 %4, %8, %4\n v_xor_b32 %5, %8, %5\n v_xor_b32 %6, %8, %6\n v_xor_b32 %7, %8, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x));)
    } else if (OP == kBcnt) {
      REP16(asm volatile("v_bcnt_u32_b32 %0, %8, %0\n v_bcnt_u32_b32 %1, %8, %1\n v_bcnt_u32_b32 %2, %8, %2\n v_bcnt_u32_b32 %3, %8, %3\n"
                         "v_bcnt_u32_b32 %4, %8, %4\n v_bcnt_u32_b32 %5, %8, %5\n v_bcnt_u32_b32 %6, %8, %6\n v_bcnt_u32_b32 %7, %8, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x));)
    } else if (OP == kCmpSgpr) {
      unsigned long long m0, m1, m2, m3;
      REP16(asm volatile("v_cmp_le_u32_e64 %0, %4, %5\n v_cmp_le_u32_e64 %1, %4, %6\n v_cmp_le_u32_e64 %2, %4, %7\n v_cmp_le_u32_e64 %3, %4, %8\n"
                         "v_cmp_le_u32_e64 %0, %4, %9\n v_cmp_le_u32_e64 %1, %4, %10\n v_cmp_le_u32_e64 %2, %4, %11\n v_cmp_le_u32_e64 %3, %4, %12"
                         : "=s"(m0), "=s"(m1), "=s"(m2), "=s"(m3) : "v"(x), "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7));)
      i0 += (int)(m0 ^ m1 ^ m2 ^ m3);
    } else if (OP == kMax3) {
      REP16(asm volatile("v_max3_u32 %0, %8, %9, %0\n v_max3_u32 %1, %8, %9, %1\n v_max3_u32 %2, %8, %9, %2\n v_max3_u32 %3, %8, %9, %3\n"
                         "v_max3_u32 %4, %8, %9, %4\n v_max3_u32 %5, %8, %9, %5\n v_max3_u32 %6, %8, %9, %6\n v_max3_u32 %7, %8, %9, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));)
    } else if (OP == kCell3) {  // eight cells: xor, popcount, compare to an SGPR mask, masks ANDed in the scalar unit
      unsigned long long m0 = ~0ull;
      REP16(asm volatile("v_xor_b32 %1, %9, %1\n v_bcnt_u32_b32 %1, %1, 0\n v_cmp_le_u32_e64 vcc, %1, %10\n s_and_b64 %0, %0, vcc\n"
                         "v_xor_b32 %2, %9, %2\n v_bcnt_u32_b32 %2, %2, 0\n v_cmp_le_u32_e64 vcc, %2, %10\n s_and_b64 %0, %0, vcc\n"
                         "v_xor_b32 %3, %9, %3\n v_bcnt_u32_b32 %3, %3, 0\n v_cmp_le_u32_e64 vcc, %3, %10\n s_and_b64 %0, %0, vcc\n"
                         "v_xor_b32 %4, %9, %4\n v_bcnt_u32_b32 %4, %4, 0\n v_cmp_le_u32_e64 vcc, %4, %10\n s_and_b64 %0, %0, vcc\n"
                         "v_xor_b32 %5, %9, %5\n v_bcnt_u32_b32 %5, %5, 0\n v_cmp_le_u32_e64 vcc, %5, %10\n s_and_b64 %0, %0, vcc\n"
                         "v_xor_b32 %6, %9, %6\n v_bcnt_u32_b32 %6, %6, 0\n v_cmp_le_u32_e64 vcc, %6, %10\n s_and_b64 %0, %0, vcc\n"
                         "v_xor_b32 %7, %9, %7\n v_bcnt_u32_b32 %7, %7, 0\n v_cmp_le_u32_e64 vcc, %7, %10\n s_and_b64 %0, %0, vcc\n"
                         "v_xor_b32 %8, %9, %8\n v_bcnt_u32_b32 %8, %8, 0\n v_cmp_le_u32_e64 vcc, %8, %10\n s_and_b64 %0, %0, vcc"
                         : "+s"(m0), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(i0) : "vcc");)
      i0 += (int)m0;
    } else if (OP == kCndmask) {
      REP16(asm volatile("v_cndmask_b32_e32 %0, %8, %0, vcc\n v_cndmask_b32_e32 %1, %8, %1, vcc\n v_cndmask_b32_e32 %2, %8, %2, vcc\n v_cndmask_b32_e32 %3, %8, %3, vcc\n"
                         "v_cndmask_b32_e32 %4, %8, %4, vcc\n v_cndmask_b32_e32 %5, %8, %5, vcc\n v_cndmask_b32_e32 %6, %8, %6, vcc\n v_cndmask_b32_e32 %7, %8, %7, vcc"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x) : "vcc");)
    } else if (OP == kDsRead) {
      REP8(asm volatile("ds_read_b64 %0, %4\n ds_read_b64 %1, %4 offset:2048\n ds_read_b64 %2, %4 offset:4096\n ds_read_b64 %3, %4 offset:6144\n"
                        "ds_read_b64 %0, %4 offset:8192\n ds_read_b64 %1, %4 offset:10240\n ds_read_b64 %2, %4 offset:12288\n ds_read_b64 %3, %4 offset:14336\n"
                        "s_waitcnt lgkmcnt(0)" : "=v"(p0), "=v"(p1), "=v"(p2), "=v"(p3) : "v"(addr) : "memory");)
    } else if (OP == kDsRead2) {
      REP8(asm volatile("ds_read2_b64 %0, %4 offset1:17\n ds_read2_b64 %1, %4 offset0:34 offset1:51\n ds_read2_b64 %2, %4 offset0:68 offset1:85\n ds_read2_b64 %3, %4 offset0:102 offset1:119\n"
                        "ds_read2_b64 %0, %4 offset0:136 offset1:153\n ds_read2_b64 %1, %4 offset0:170 offset1:187\n ds_read2_b64 %2, %4 offset0:204 offset1:221\n ds_read2_b64 %3, %4 offset0:238 offset1:255\n"
                        "s_waitcnt lgkmcnt(0)" : "=v"(q0), "=v"(q1), "=v"(q2), "=v"(q3) : "v"(addr) : "memory");)
    } else if (OP == kDsWrite) {
      REP8(asm volatile("ds_write_b64 %4, %0\n ds_write_b64 %4, %1 offset:2048\n ds_write_b64 %4, %2 offset:4096\n ds_write_b64 %4, %3 offset:6144\n"
                        "ds_write_b64 %4, %0 offset:8192\n ds_write_b64 %4, %1 offset:10240\n ds_write_b64 %4, %2 offset:12288\n ds_write_b64 %4, %3 offset:14336\n"
                        "s_waitcnt lgkmcnt(0)" :: "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(addr) : "memory");)
    } else if (OP == kDsWrite2) {
      REP8(asm volatile("ds_write2_b64 %4, %0, %1 offset1:17\n ds_write2_b64 %4, %2, %3 offset0:34 offset1:51\n ds_write2_b64 %4, %0, %1 offset0:68 offset1:85\n ds_write2_b64 %4, %2, %3 offset0:102 offset1:119\n"
                        "ds_write2_b64 %4, %0, %1 offset0:136 offset1:153\n ds_write2_b64 %4, %2, %3 offset0:170 offset1:187\n ds_write2_b64 %4, %0, %1 offset0:204 offset1:221\n ds_write2_b64 %4, %2, %3 offset0:238 offset1:255\n"
                        "s_waitcnt lgkmcnt(0)" :: "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(addr) : "memory");)
    } else if (OP == kLoadShort) {
      int r0, r1, r2, r3;
      REP8(asm volatile("global_load_sshort %0, %4, off\n global_load_sshort %1, %4, off offset:512\n global_load_sshort %2, %4, off offset:1024\n global_load_sshort %3, %4, off offset:1536\n"
                        "global_load_sshort %0, %4, off offset:2048\n global_load_sshort %1, %4, off offset:2560\n global_load_sshort %2, %4, off offset:3072\n global_load_sshort %3, %4, off offset:3584\n"
                        "s_waitcnt vmcnt(0)" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(q) : "memory");)
      i0 += r0 + r1 + r2 + r3;
    } else if (OP == kMixValuLds || OP == kMixValuLdsVmem || OP == kMixValuOnly) {
      int r0 = 0;
      // 16 VALU in the kernel's blend (6 fmac, 4 add, 2 sub, 2 mul, 1 fmamk, 1 pk_add), then one LDS store and one LDS read
#define VALU16                                                                                                                  \
  "v_fmac_f32_e32 %0, %8, %9\n v_add_f32_e32 %1, %8, %1\n v_fmac_f32_e32 %2, %8, %9\n v_sub_f32_e32 %3, %8, %3\n"               \
  "v_mul_f32_e32 %4, %8, %4\n v_fmac_f32_e32 %5, %8, %9\n v_add_f32_e32 %6, %8, %6\n v_fmamk_f32 %7, %8, 0x3f6c835e, %7\n"      \
  "v_fmac_f32_e32 %0, %8, %9\n v_add_f32_e32 %1, %8, %1\n v_fmac_f32_e32 %2, %8, %9\n v_sub_f32_e32 %3, %8, %3\n"               \
  "v_mul_f32_e32 %4, %8, %4\n v_fmac_f32_e32 %5, %8, %9\n v_add_f32_e32 %6, %8, %6\n v_pk_add_f32 %10, %10, %11\n"
      REP8(asm volatile(VALU16 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y), "v"(p0), "v"(px));
           if (OP != kMixValuOnly) asm volatile("ds_write_b64 %1, %0 offset:2048\n ds_read_b64 %2, %1 offset:4096" : : "v"(p1), "v"(addr), "v"(p2) : "memory");)
      if (OP == kMixValuLdsVmem) {
        REP4(asm volatile("global_load_sshort %0, %1, off offset:512" : "=v"(r0) : "v"(q) : "memory");)
        asm volatile("global_load_sshort %0, %1, off offset:1024\n global_load_sshort %0, %1, off offset:1536" : "=v"(r0) : "v"(q) : "memory");
      }
      if (OP != kMixValuOnly) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      i0 += r0;
    }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  if (t == 0) cycles[blockIdx.x] = c1 - c0;
  sink[blockIdx.x * 256 + t] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + (float)i0 + q0.x + q1.y + q2.z + q3.w;
}

template <int OP>
static void run(unsigned long long *d_cycles, float *d_sink, const short *d_pcm) {
  const int iters = 400;
  for (int wgs_per_cu : {1, 2, 3, 4, 5, 8}) {
    const int grid = 256 * wgs_per_cu;
    // 160 KiB of LDS per CU: 40 KiB per workgroup admits four, and the grid is 256 x the count wanted
    hipFuncSetAttribute(reinterpret_cast<const void *>(rate_kernel<OP>), hipFuncAttributeMaxDynamicSharedMemorySize, 40 * 1024);
    std::vector<unsigned long long> h(grid);
    double best = 1e30;
    for (int rep = 0; rep < 3; rep++) {
      rate_kernel<OP><<<grid, 256, (wgs_per_cu >= 8 ? 18 : wgs_per_cu == 5 ? 30 : wgs_per_cu == 4 ? 36 : 40) * 1024>>>(d_cycles, d_sink, d_pcm, iters);
      hipMemcpy(h.data(), d_cycles, grid * sizeof(unsigned long long), hipMemcpyDeviceToHost);
      std::sort(h.begin(), h.end());
      best = std::min(best, (double)h[grid / 2]);
    }
    // all waves of a SIMD share it: cycles per wave-instruction ISSUED ON THAT SIMD = loop cycles / (instructions of one wave x waves per SIMD)
    std::printf("%-52s %d wave(s)/SIMD: %7.2f cycles per wave-instruction of one wave, %6.2f per instruction on the SIMD\n", kNames[OP],
                wgs_per_cu, best / ((double)iters * kPerIter[OP]), best / ((double)iters * kPerIter[OP] * wgs_per_cu));
  }
}

int main() {
  unsigned long long *d_cycles;
  float *d_sink;
  short *d_pcm;
  hipMalloc(&d_cycles, 4096 * sizeof(unsigned long long));
  hipMalloc(&d_sink, 4096 * 256 * sizeof(float));
  hipMalloc(&d_pcm, 1 << 20);
  hipMemset(d_pcm, 1, 1 << 20);
  run<kFmac>(d_cycles, d_sink, d_pcm);
  run<kAdd>(d_cycles, d_sink, d_pcm);
  run<kMul>(d_cycles, d_sink, d_pcm);
  run<kFmamk>(d_cycles, d_sink, d_pcm);
  run<kMulE64Neg>(d_cycles, d_sink, d_pcm);
  run<kPkAdd>(d_cycles, d_sink, d_pcm);
  run<kPkFma>(d_cycles, d_sink, d_pcm);
  run<kCvt>(d_cycles, d_sink, d_pcm);
  run<kAddDpp>(d_cycles, d_sink, d_pcm);
  run<kFmacDep1>(d_cycles, d_sink, d_pcm);
  run<kFmacDep2>(d_cycles, d_sink, d_pcm);
  run<kFmacBank>(d_cycles, d_sink, d_pcm);
  run<kFmacSame>(d_cycles, d_sink, d_pcm);
  run<kFmacSgpr>(d_cycles, d_sink, d_pcm);
  run<kAddNoAcc>(d_cycles, d_sink, d_pcm);
  run<kXor>(d_cycles, d_sink, d_pcm);
  run<kBcnt>(d_cycles, d_sink, d_pcm);
  run<kCmpSgpr>(d_cycles, d_sink, d_pcm);
  run<kMax3>(d_cycles, d_sink, d_pcm);
  run<kCell3>(d_cycles, d_sink, d_pcm);
  run<kCndmask>(d_cycles, d_sink, d_pcm);
  run<kDsRead>(d_cycles, d_sink, d_pcm);
  run<kDsRead2>(d_cycles, d_sink, d_pcm);
  run<kDsWrite>(d_cycles, d_sink, d_pcm);
  run<kDsWrite2>(d_cycles, d_sink, d_pcm);
  run<kLoadShort>(d_cycles, d_sink, d_pcm);
  run<kMixValuOnly>(d_cycles, d_sink, d_pcm);
  run<kMixValuLds>(d_cycles, d_sink, d_pcm);
  run<kMixValuLdsVmem>(d_cycles, d_sink, d_pcm);
  return 0;
}
