// Diagnostic (not part of the product): what bounds stft_chroma_kernel?  Measures, at the kernel's own occupancy
// (256 threads, 2 workgroups/CU, 69.6 KB LDS each), (1) the f64 VALU issue rate, (2) one frame pair's arithmetic
// with no LDS traffic, (3) one frame pair's LDS traffic with no arithmetic, (4) both together without global loads.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <vector>

#include "../needle_amd/csrc/fp_core.h"

using needle::core::cd;
namespace core = needle::core;

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int OP, int VREG>
__global__ __launch_bounds__(256, 2) void rate_kernel(double *out, int iters, double x) {
  extern __shared__ cd lds[];
  double a[16];
  if (VREG) asm volatile("" : "+v"(x));  // operand from a VGPR instead of an SGPR pair
#pragma unroll
  for (int i = 0; i < 16; i++) a[i] = x * (threadIdx.x + i);
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 16; i++) {
      if (OP == 0) a[i] = a[i] + x;
      if (OP == 1) a[i] = a[i] * x;
      if (OP == 2) a[i] = __builtin_fma(a[i], x, x);
      if (OP == 3) a[i] = (double)(int)__double_as_longlong(a[i]) ;  // v_cvt_f64_i32 on the low dword
    }
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) s += a[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (x == 123.0) lds[threadIdx.x] = cd{s, s};
}

// MODE 0: arithmetic only; 1: LDS traffic only; 2: both (the kernel minus global loads, power fold and its barriers)
template <int MODE>
__global__ __launch_bounds__(256, 2) void pair_kernel(const cd *__restrict__ tw, double *out, int pairs) {
  extern __shared__ cd lds[];
  int t = threadIdx.x;
  const cd base0 = tw[t], base1 = tw[16 * (t & 15)];
  cd r[16];
#pragma unroll
  for (int k = 0; k < 16; k++) r[k] = cd{1.0 / (t + k + 1), 0.5 / (t + 2 * k + 1)};
  double keep = 0;
  for (int g = 0; g < pairs; g++) {
    int tt = t;
    asm volatile("" : "+v"(tt));
    const int mode = MODE == 3 ? (blockIdx.x < gridDim.x / 2 ? 0 : 1) : MODE;
    if (mode == 0) {
      for (int st = 0; st < 3; st++) {
        core::fft16(r);
        if (st < 2) {
          const cd b = st ? base1 : base0;
          cd w = b;
#pragma unroll
          for (int j = 1; j < 16; j++) {
            r[j] = core::cmulf(r[j], w);
            if (j < 15) w = core::cmulf(w, b);
          }
        }
      }
#pragma unroll
      for (int k = 0; k < 16; k++) { r[k].x *= 1e-3; r[k].y *= 1e-3; }
    } else if (mode == 1) {
#pragma unroll
      for (int j = 0; j < 16; j++) lds[core::pidx(tt + 256 * j)] = r[j];
      lds_barrier();
      const int o = 256 * (tt >> 4) + (tt & 15);
#pragma unroll
      for (int k = 0; k < 16; k++) r[k] = core::lds_get(lds, core::pidx(o + 16 * k));
#pragma unroll
      for (int j = 0; j < 16; j++) lds[core::pidx(o + 16 * j)] = r[(j + 1) & 15];
#pragma unroll
      for (int k = 0; k < 16; k++) r[k] = core::lds_get(lds, core::pidx(16 * tt + k));
#pragma unroll
      for (int j = 10; j < 16; j++) lds[core::pidx(16 * tt + j)] = r[j - 3];
      lds_barrier();
#pragma unroll
      for (int j = 0; j < 6; j++) {
        const cd y = core::lds_get(lds, core::pidx(core::dif_slot_of_bin(core::kFft2N - core::dif_bin_of(tt, j) - 1)));
        r[j].x += y.x;
        r[j].y += y.y;
      }
      lds_barrier();
    } else {
      core::dif0(tt, base0, lds, r);
      lds_barrier();
      core::dif1(tt, base1, lds, r);
      core::dif2(tt, lds, r);
      core::dif2_publish(tt, lds, r);
      lds_barrier();
#pragma unroll
      for (int j = 0; j < 6; j++) {
        double pa = 0, pb = 0;
        int kf;
        core::dif_bin_power(tt, j, lds, r, &kf, &pa, &pb);
        keep += pa + pb;
      }
      lds_barrier();
#pragma unroll
      for (int k = 0; k < 16; k++) { r[k].x = r[k].x * 1e-3 + 1.0; r[k].y = r[k].y * 1e-3 + 0.5; }
    }
  }
#pragma unroll
  for (int k = 0; k < 16; k++) keep += r[k].x + r[k].y;
  out[blockIdx.x * 256 + t] = keep;
}

int main() {
  const int blocks = 512;
  std::vector<cd> tw(4096);
  for (int k = 0; k < 4096; k++) tw[k] = cd{std::cos(-2 * M_PI * k / 4096), std::sin(-2 * M_PI * k / 4096)};
  cd *d_tw; double *d_out;
  hipMalloc(&d_tw, 4096 * sizeof(cd)); hipMalloc(&d_out, blocks * 256 * 8);
  hipMemcpy(d_tw, tw.data(), 4096 * sizeof(cd), hipMemcpyHostToDevice);
  const size_t lds = core::kLds2Slots * sizeof(cd);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  auto time = [&](auto launch) {
    launch(2);
    hipDeviceSynchronize();
    hipEventRecord(a);
    launch(0);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms;
  };
  const char *ops[4] = {"v_add_f64", "v_mul_f64", "v_fma_f64", "v_cvt_f64_i32"};
  const int iters = 4000;
#define RATE(OP, V)                                                                                              \
  {                                                                                                           \
    hipFuncSetAttribute(reinterpret_cast<const void *>(rate_kernel<OP, V>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    float ms = time([&](int small) { rate_kernel<OP, V><<<blocks, 256, lds>>>(d_out, small ? 10 : iters, 1.0000001); });           \
    double instr = (double)blocks * 4 * iters * 16;                                                           \
    printf("%-14s %s %.3f ms  %.2f cycles/wave-instr/SIMD at 2.4 GHz (2 waves/SIMD)\n", ops[OP], V ? "vgpr operand" : "sgpr operand", ms,            \
           ms * 1e-3 * 2.4e9 / (instr / (256 * 4)));                                                          \
  }
  RATE(0, 0) RATE(0, 1) RATE(0, 1) RATE(2, 1)
  const int pairs = 160;  // per block; the product's 28 x 24 min batch is 81 382 pairs = 159 per block at 512 blocks
  const char *modes[3] = {"arithmetic only (3 fft16 + 2 twiddle passes)", "LDS traffic only", "arithmetic + LDS (no global loads, no fold)"};
#define PAIR(M)                                                                                               \
  {                                                                                                           \
    hipFuncSetAttribute(reinterpret_cast<const void *>(pair_kernel<M>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    float ms = time([&](int small) { pair_kernel<M><<<blocks, 256, lds>>>(d_tw, d_out, small ? 2 : pairs); });  \
    printf("%-48s %.3f ms for %d pairs  (%.0f cycles/pair/CU at 2.4 GHz)\n", modes[M], ms, blocks * pairs,      \
           ms * 1e-3 * 2.4e9 / (blocks * pairs / 256.0));                                                     \
  }
  PAIR(0) PAIR(1) PAIR(2)
  {
    const char *names[3] = {"256 workgroups arithmetic only (1/CU)", "256 workgroups LDS only (1/CU)", "256 arithmetic + 256 LDS workgroups together"};
    hipFuncSetAttribute(reinterpret_cast<const void *>(pair_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    float m0 = time([&](int small) { pair_kernel<0><<<256, 256, lds>>>(d_tw, d_out, small ? 2 : pairs); });
    float m1 = time([&](int small) { pair_kernel<1><<<256, 256, lds>>>(d_tw, d_out, small ? 2 : pairs); });
    float m3 = time([&](int small) { pair_kernel<3><<<512, 256, lds>>>(d_tw, d_out, small ? 2 : pairs); });
    printf("%s: %.3f ms\n%s: %.3f ms\n%s: %.3f ms\n", names[0], m0, names[1], m1, names[2], m3);
  }
  return 0;
}
