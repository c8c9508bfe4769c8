// Round 6 laboratory (not part of the product): shader cycles per wave-instruction (s_memtime) of the instructions a split-f16
// matrix-pipe transform lives on -- v_cvt_pk_f16_f32, v_fma_mixlo/hi_f16, v_mfma_f32_32x32x16_f16 alone and beside vector
// instructions -- at 1, 2 and 3 waves per SIMD, and the clock the chip holds meanwhile (cycles / wall time).
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/mfma_split_rates.hip -o tools/mfma_split_rates
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

enum Op { kCvtPk, kMixLo, kMixHi, kSplit3, kMul, kFmac, kMfma, kMfma2Acc, kMfmaPlus6Mul, kMfmaPlus3Split, kMfmaPlus9Mix, kNumOps };
static const char *kNames[kNumOps] = {"v_cvt_pk_f16_f32 (8 independent)", "v_fma_mixlo_f16 (8 independent)", "v_fma_mixhi_f16 (8 independent)",
                                      "split: cvt_pk + mixlo + mixhi (dependent triple, 8 triples)", "v_mul_f32 (8 independent)", "v_fmac_f32 (8 independent)",
                                      "v_mfma_f32_32x32x16_f16, one accumulator chain", "v_mfma_f32_32x32x16_f16, two accumulators",
                                      "mfma + 6 v_mul_f32 per mfma", "mfma + 3 splits (9 instructions) per mfma", "mfma + the kernel's 9 (3 mul, 2 fmac, cvt_pk, 2 mix, cvt_i32)"};
static const int kPerIter[kNumOps] = {128, 128, 128, 384, 128, 128, 16, 16, 16 * 7, 16 * 10, 16 * 10};

template <int OP>
__global__ __launch_bounds__(256) void rate_kernel(unsigned long long *cycles, float *sink, const half8 *ab, int iters) {
  extern __shared__ float lds_unused[];
  const int t = threadIdx.x;
  if (iters < 0) lds_unused[t] = 0;
  float a0 = t, a1 = t + 1, a2 = t + 2, a3 = t + 3, a4 = t + 4, a5 = t + 5, a6 = t + 6, a7 = t + 7;
  float x = 1.0000001f, y = 0.9999999f;
  uint32_t h0 = 0, h1 = 0, h2 = 0, h3 = 0, h4 = 0, h5 = 0, h6 = 0, h7 = 0;
  int i0 = t;
  asm volatile("" : "+v"(x), "+v"(y), "+v"(i0));
  const half8 fa = ab[t & 63], fb = ab[64 + (t & 63)];
  f32x16 acc0 = {}, acc1 = {};
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
    if (OP == kCvtPk) {
      REP16(asm volatile("v_cvt_pk_f16_f32 %0, %8, %9\n v_cvt_pk_f16_f32 %1, %8, %9\n v_cvt_pk_f16_f32 %2, %8, %9\n v_cvt_pk_f16_f32 %3, %8, %9\n"
                         "v_cvt_pk_f16_f32 %4, %8, %9\n v_cvt_pk_f16_f32 %5, %8, %9\n v_cvt_pk_f16_f32 %6, %8, %9\n v_cvt_pk_f16_f32 %7, %8, %9"
                         : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3), "+v"(h4), "+v"(h5), "+v"(h6), "+v"(h7) : "v"(x), "v"(y));)
    } else if (OP == kMixLo || OP == kMixHi) {
#define MIX(op, sel) op " %0, %8, -1.0, %9 " sel "\n " op " %1, %8, -1.0, %9 " sel "\n " op " %2, %8, -1.0, %9 " sel "\n " op " %3, %8, -1.0, %9 " sel "\n " \
                     op " %4, %8, -1.0, %9 " sel "\n " op " %5, %8, -1.0, %9 " sel "\n " op " %6, %8, -1.0, %9 " sel "\n " op " %7, %8, -1.0, %9 " sel
      if (OP == kMixLo) {
        REP16(asm volatile(MIX("v_fma_mixlo_f16", "op_sel_hi:[1,0,0]")
                           : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3), "+v"(h4), "+v"(h5), "+v"(h6), "+v"(h7) : "v"(i0), "v"(y));)
      } else {
        REP16(asm volatile(MIX("v_fma_mixhi_f16", "op_sel:[1,0,0] op_sel_hi:[1,0,0]")
                           : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3), "+v"(h4), "+v"(h5), "+v"(h6), "+v"(h7) : "v"(i0), "v"(y));)
      }
    } else if (OP == kSplit3) {
#define SPLIT(H, L) "v_cvt_pk_f16_f32 " H ", %16, %17\n v_fma_mixlo_f16 " L ", " H ", -1.0, %16 op_sel_hi:[1,0,0]\n v_fma_mixhi_f16 " L ", " H ", -1.0, %17 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n "
      REP16(asm volatile(SPLIT("%0", "%8") SPLIT("%1", "%9") SPLIT("%2", "%10") SPLIT("%3", "%11") SPLIT("%4", "%12") SPLIT("%5", "%13") SPLIT("%6", "%14") SPLIT("%7", "%15")
                         : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3), "+v"(h4), "+v"(h5), "+v"(h6), "+v"(h7), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5),
                           "+v"(a6), "+v"(a7) : "v"(x), "v"(y));)
    } else if (OP == kMul) {
      REP16(asm volatile("v_mul_f32_e32 %0, %8, %0\n v_mul_f32_e32 %1, %8, %1\n v_mul_f32_e32 %2, %8, %2\n v_mul_f32_e32 %3, %8, %3\n"
                         "v_mul_f32_e32 %4, %8, %4\n v_mul_f32_e32 %5, %8, %5\n v_mul_f32_e32 %6, %8, %6\n v_mul_f32_e32 %7, %8, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x));)
    } else if (OP == kFmac) {
      REP16(asm volatile("v_fmac_f32_e32 %0, %8, %9\n v_fmac_f32_e32 %1, %8, %9\n v_fmac_f32_e32 %2, %8, %9\n v_fmac_f32_e32 %3, %8, %9\n"
                         "v_fmac_f32_e32 %4, %8, %9\n v_fmac_f32_e32 %5, %8, %9\n v_fmac_f32_e32 %6, %8, %9\n v_fmac_f32_e32 %7, %8, %9"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));)
    } else {
#pragma unroll
      for (int m = 0; m < 16; m++) {
        if (OP == kMfma || (m & 1) == 0) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc0, 0, 0, 0);
        else acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc1, 0, 0, 0);
        if (OP == kMfmaPlus6Mul)
          asm volatile("v_mul_f32_e32 %0, %6, %0\n v_mul_f32_e32 %1, %6, %1\n v_mul_f32_e32 %2, %6, %2\n v_mul_f32_e32 %3, %6, %3\n v_mul_f32_e32 %4, %6, %4\n v_mul_f32_e32 %5, %6, %5"
                       : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5) : "v"(x));
        if (OP == kMfmaPlus3Split)
          asm volatile(SPLIT("%0", "%8") SPLIT("%1", "%9") SPLIT("%2", "%10")
                       : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3), "+v"(h4), "+v"(h5), "+v"(h6), "+v"(h7), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5),
                         "+v"(a6), "+v"(a7) : "v"(x), "v"(y));
        if (OP == kMfmaPlus9Mix)
          asm volatile("v_mul_f32 %4, %0, %2\n v_mul_f32 %5, %1, %3\n v_fmac_f32 %4, %1, %2\n v_fmac_f32 %5, %0, %3\n v_cvt_pk_f16_f32 %6, %4, %5\n"
                       "v_fma_mixlo_f16 %7, %6, -1.0, %4 op_sel_hi:[1,0,0]\n v_fma_mixhi_f16 %7, %6, -1.0, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_mul_f32 %0, %4, %3\n v_cvt_f32_i32 %1, %8"
                       : "+v"(a0), "+v"(a1), "+v"(x), "+v"(y), "+v"(a2), "+v"(a3), "+v"(h0), "+v"(h1) : "v"(i0));
      }
    }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  if (t == 0) cycles[blockIdx.x] = c1 - c0;
  sink[blockIdx.x * 256 + t] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(h0 + h1 + h2 + h3 + h4 + h5 + h6 + h7) + acc0[0] + acc1[1] + x + y;
}

template <int OP>
static void run(unsigned long long *d_cycles, float *sink, const half8 *ab) {
  for (int waves : {1, 2, 3}) {
    const int grid = 256 * waves, iters = OP >= kMfma ? 2000 : 400;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    // LDS sized so that exactly `waves` workgroups fit a CU (160 KB): the dispatcher cannot stack them unevenly
    const size_t lds = waves == 1 ? 100 * 1024 : waves == 2 ? 70 * 1024 : 50 * 1024;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(rate_kernel<OP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(rate_kernel<OP>, dim3(grid), dim3(256), lds, 0, d_cycles, sink, ab, iters);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(rate_kernel<OP>, dim3(grid), dim3(256), lds, 0, d_cycles, sink, ab, iters);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> c(grid);
    CK(hipMemcpy(c.data(), d_cycles, grid * 8, hipMemcpyDeviceToHost));
    std::sort(c.begin(), c.end());
    const double per = (double)c[grid / 2] / ((double)iters * kPerIter[OP]);
    std::printf("%-72s %d wave(s)/SIMD: %7.2f cycles per wave-instruction of one wave, %6.2f per instruction on the SIMD; clock ~%.2f GHz\n", kNames[OP], waves, per,
                per / waves, (double)c[grid / 2] / (ms * 1e6));
  }
}

int main() {
  unsigned long long *d_cycles;
  float *sink;
  half8 *ab;
  CK(hipMalloc(&d_cycles, 1024 * 8));
  CK(hipMalloc(&sink, 1024 * 256 * 4));
  CK(hipMalloc(&ab, 128 * 16));
  CK(hipMemset(ab, 0x3c, 128 * 16));
  run<kCvtPk>(d_cycles, sink, ab);
  run<kMixLo>(d_cycles, sink, ab);
  run<kMixHi>(d_cycles, sink, ab);
  run<kSplit3>(d_cycles, sink, ab);
  run<kMul>(d_cycles, sink, ab);
  run<kFmac>(d_cycles, sink, ab);
  run<kMfma>(d_cycles, sink, ab);
  run<kMfma2Acc>(d_cycles, sink, ab);
  run<kMfmaPlus6Mul>(d_cycles, sink, ab);
  run<kMfmaPlus3Split>(d_cycles, sink, ab);
  run<kMfmaPlus9Mix>(d_cycles, sink, ab);
  return 0;
}
