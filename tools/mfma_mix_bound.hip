// What could a PERFECTLY interleaved instruction stream of the matrix-pipe first pass reach?  (round 6 laboratory, not part
// of the product.)  One loop trip = one frame pair's instruction mix of tools/stft_mfma_core.h per wave -- 36
// v_mfma_f32_32x32x16_f16, ~330 vector instructions in the kernel's classes (v_mul, v_fmac, v_cvt_pk_f16_f32,
// v_fma_mix{lo,hi}_f16, v_cvt_f32_i32), 32 ds_write2_b32, 16 ds_read_b128 -- on registers only, hand-interleaved (one matrix
// instruction, then its share of the vector and LDS instructions), no barriers, no dependences between groups beyond the
// accumulator chains.  Variants: the mix; matrix instructions only; vector instructions only; without the LDS instructions;
// the matrix instructions in bursts of 12 (what the compiler's order amounts to).  3 workgroups of 4 waves per CU.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/mfma_mix_bound.hip -o tools/mfma_mix_bound
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

enum : int { kMfma = 1, kValu = 2, kLds = 4, kBurst = 8 };

// the vector share of one matrix instruction: 9 instructions in the kernel's proportions (of 330: 96 v_mul, 64 v_fmac,
// 48 v_cvt_pk, 96 v_fma_mix, 32 v_cvt_f32_i32 -> per 9: 3 mul, 2 fmac, 1 cvt_pk, 2-3 mix, 1 cvt_i32 every other group)
#define VALU_GROUP(x0, x1, x2, x3, t0, t1, h, lo, s)                                                                       \
  asm volatile("v_mul_f32 %4, %0, %2\n\tv_mul_f32 %5, %1, %3\n\tv_fmac_f32 %4, %1, %2\n\tv_fmac_f32 %5, %0, %3\n\t"      \
               "v_cvt_pk_f16_f32 %6, %4, %5\n\tv_fma_mixlo_f16 %7, %6, -1.0, %4 op_sel_hi:[1,0,0]\n\t"                   \
               "v_fma_mixhi_f16 %7, %6, -1.0, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\tv_mul_f32 %0, %4, %3\n\t"            \
               "v_cvt_f32_i32 %1, %8"                                                                                      \
               : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(t0), "+v"(t1), "+v"(h), "+v"(lo) : "v"(s))

template <int MODE>
__global__ __launch_bounds__(256, 3) void mix_kernel(const half8 *a, float *out, int trips) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  const int t = threadIdx.x;
  const half8 a0 = a[t & 63], a1 = a[64 + (t & 63)];
  half8 b0 = a[128 + (t & 63)], b1 = a[192 + (t & 63)];
  f32x16 acc0 = {}, acc1 = {};
  float x0 = t, x1 = t + 1, x2 = 1.0001f, x3 = 0.9999f, t0 = 0, t1 = 0;
  float y0 = t + 2, y1 = t + 3, u0 = 0, u1 = 0;
  uint32_t h = 0, lo = 0, h2 = 0, lo2 = 0;
  int s = t;
  const int wbase = (t * 9) & 8191, rbase = (t * 36) & 8188;
  u32x4 r0 = {}, r1 = {};
  for (int it = 0; it < trips; it++) {
#pragma unroll
    for (int m = 0; m < 36; m++) {
      if (MODE & kMfma) {
        if (!(MODE & kBurst) || true) {
          if (m & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, acc1, 0, 0, 0);
          else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc0, 0, 0, 0);
        }
      }
      if (MODE & kBurst) {
        if (m % 12 != 11) continue;  // the vector / LDS share of twelve matrix instructions behind the twelfth
#pragma unroll
        for (int k = 0; k < 12; k++) {
          if (MODE & kValu) {
            if (k & 1) VALU_GROUP(y0, y1, x2, x3, u0, u1, h2, lo2, s); else VALU_GROUP(x0, x1, x2, x3, t0, t1, h, lo, s);
          }
          if (MODE & kLds) {
            if ((m - 11 + k) % 9 != 8) asm volatile("ds_write2_b32 %0, %1, %2 offset1:4" ::"v"(4 * wbase + 16 * ((m + k) & 7)), "v"(h), "v"(lo) : "memory");
            if ((m - 11 + k) % 9 < 4) asm volatile("ds_read_b128 %0, %1" : "=v"(r0) : "v"(4 * rbase) : "memory");
          }
        }
        continue;
      }
      if (MODE & kValu) {
        if (m & 1) VALU_GROUP(y0, y1, x2, x3, u0, u1, h2, lo2, s); else VALU_GROUP(x0, x1, x2, x3, t0, t1, h, lo, s);
      }
      if (MODE & kLds) {
        if (m % 9 != 8) asm volatile("ds_write2_b32 %0, %1, %2 offset1:4" ::"v"(4 * wbase + 16 * (m & 7)), "v"(h), "v"(lo) : "memory");
        if (m % 9 < 4) asm volatile("ds_read_b128 %0, %1" : "=v"(m & 1 ? r1 : r0) : "v"(4 * rbase + 64 * (m & 3)) : "memory");
      }
    }
    if (MODE & kLds) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r0), "+v"(r1)::"memory");
    b0 = __builtin_bit_cast(half8, r0) + b0;  // (keeps the reads alive; never a NaN concern for timing)
  }
  out[blockIdx.x * 256 + t] = acc0[0] + acc1[3] + x0 + x1 + y0 + y1 + (float)h + (float)lo2 + (float)b0[0];
}

template <typename K>
static void run(const char *name, K kernel, const half8 *a, float *out) {
  const int grid = 768, trips = 106;  // 3 workgroups per CU, each a slot's ~106 frame pairs of a 28 x 24 min launch
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 53 * 1024));
  std::vector<float> ms;
  for (int r = 0; r < 7; r++) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 53 * 1024, 0, a, out, trips);  // 53 KB of LDS each: exactly three workgroups fit a CU
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float v;
    CK(hipEventElapsedTime(&v, e0, e1));
    if (r >= 2) ms.push_back(v);
  }
  std::sort(ms.begin(), ms.end());
  std::printf("%-72s median %.4f ms  best %.4f  (= %.0f cycles per trip and wave at 2.4 GHz)\n", name, ms[ms.size() / 2], ms[0], ms[ms.size() / 2] * 2.4e6 / trips);
}

int main() {
  half8 *a;
  float *out;
  CK(hipMalloc(&a, 256 * 16));
  CK(hipMemset(a, 0x3c, 256 * 16));  // 0x3c3c = 1.06 in f16: finite everywhere
  CK(hipMalloc(&out, 768 * 256 * 4));
  for (int round = 0; round < 2; round++) {
    run("the mix, one matrix instruction : 9 vector : ~1.3 LDS, interleaved", mix_kernel<kMfma | kValu | kLds>, a, out);
    run("  matrix instructions only (36 per trip)", mix_kernel<kMfma>, a, out);
    run("  vector instructions only (324 per trip)", mix_kernel<kValu>, a, out);
    run("  vector + LDS", mix_kernel<kValu | kLds>, a, out);
    run("  matrix + vector, no LDS", mix_kernel<kMfma | kValu>, a, out);
    run("  the mix with the matrix instructions in bursts of 12", mix_kernel<kMfma | kValu | kLds | kBurst>, a, out);
  }
  return 0;
}
