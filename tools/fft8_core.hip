// Diagnostic (not part of the product): would a radix-8 x 4 schedule with 512 threads per frame pair -- half the
// registers per thread, so FOUR waves per SIMD instead of two -- beat the product's radix-16 x 3 / 256 threads?
// Times the FFT core only (butterflies, twiddles, LDS exchanges, partner exchange) of both, same launch shape as the
// product (2 workgroups per CU), fake inputs; outputs are not checked here (the index maps are the real ones, but
// this file exists to decide whether the variant is worth building).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <vector>

#include "../needle_amd/csrc/fp_core.h"

using needle::core::cd;
namespace core = needle::core;

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void wave_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ int p8(int i) { return i + (i >> 3); }  // row pitch 9

// 8-point DFT in registers, output k in a[rev3(k)] order handled by out8()
__device__ __forceinline__ int out8(int k) { return ((k & 1) << 2) | (k & 2) | ((k >> 2) & 1); }
__device__ __forceinline__ void fft8(cd *a) {
  const double h = 0.70710678118654752440;
  // stage A: pairs (i, i+4)
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const cd u = a[i], v = a[i + 4];
    a[i] = core::cadd(u, v);
    a[i + 4] = core::csub(u, v);
  }
  // twiddles W8^0..3 on a[4..7]
  { cd v = a[5]; a[5] = cd{(v.x + v.y) * h, (v.y - v.x) * h}; }
  { cd v = a[6]; a[6] = cd{v.y, -v.x}; }
  { cd v = a[7]; a[7] = cd{(v.y - v.x) * h, -(v.x + v.y) * h}; }
  // stage B: within halves, pairs (i, i+2)
#pragma unroll
  for (int hh = 0; hh < 8; hh += 4)
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const cd u = a[hh + i], v = a[hh + i + 2];
      a[hh + i] = core::cadd(u, v);
      a[hh + i + 2] = core::csub(u, v);
    }
  { cd v = a[3]; a[3] = cd{v.y, -v.x}; }
  { cd v = a[7]; a[7] = cd{v.y, -v.x}; }
  // stage C: pairs (i, i+1)
#pragma unroll
  for (int i = 0; i < 8; i += 2) {
    const cd u = a[i], v = a[i + 1];
    a[i] = core::cadd(u, v);
    a[i + 1] = core::csub(u, v);
  }
}

__global__ __launch_bounds__(512, 4) void core8(const cd *__restrict__ tw, double *out, int pairs) {
  extern __shared__ cd lds[];
  const int t = threadIdx.x;
  const cd b0_ = tw[t], b1_ = tw[8 * (t & 63)], b2_ = tw[64 * (t & 7)];
  cd r[8];
#pragma unroll
  for (int k = 0; k < 8; k++) r[k] = cd{1.0 / (t + k + 1), 0.5 / (t + 2 * k + 1)};
  double keep = 0;
  for (int g = 0; g < pairs; g++) {
    int tt = t;
    asm volatile("" : "+v"(tt));
    cd b0 = b0_, b1 = b1_, b2 = b2_;
    asm volatile("" : "+v"(b0.x), "+v"(b0.y), "+v"(b1.x), "+v"(b1.y), "+v"(b2.x), "+v"(b2.y));  // powers recomputed per pair
    fft8(r);
    lds_barrier();
    {  // stage 0 store: slots t + 512 j, twiddle W^{t j}
      lds[p8(tt)] = r[out8(0)];
      cd w = b0;
#pragma unroll
      for (int j = 1; j < 8; j++) {
        lds[p8(tt + 512 * j)] = core::cmulf(r[out8(j)], w);
        if (j < 7) w = core::cmulf(w, b0);
      }
    }
    lds_barrier();
    {  // stage 1 (one wave = one b): slots 512 b + r + 64 k
      const int o = 512 * (tt >> 6) + (tt & 63);
#pragma unroll
      for (int k = 0; k < 8; k++) r[k] = core::lds_get(lds, p8(o + 64 * k));
      fft8(r);
      lds[p8(o)] = r[out8(0)];
      cd w = b1;
#pragma unroll
      for (int j = 1; j < 8; j++) {
        lds[p8(o + 64 * j)] = core::cmulf(r[out8(j)], w);
        if (j < 7) w = core::cmulf(w, b1);
      }
    }
    wave_fence();
    {  // stage 2: slots 64 c + r' + 8 k
      const int o = 64 * (tt >> 3) + (tt & 7);
#pragma unroll
      for (int k = 0; k < 8; k++) r[k] = core::lds_get(lds, p8(o + 8 * k));
      fft8(r);
      lds[p8(o)] = r[out8(0)];
      cd w = b2;
#pragma unroll
      for (int j = 1; j < 8; j++) {
        lds[p8(o + 8 * j)] = core::cmulf(r[out8(j)], w);
        if (j < 7) w = core::cmulf(w, b2);
      }
    }
    wave_fence();
    {  // stage 3: slots 8 t + k; then publish j3 = 5..7
#pragma unroll
      for (int k = 0; k < 8; k++) r[k] = core::lds_get(lds, p8(8 * tt + k));
      fft8(r);
#pragma unroll
      for (int j = 5; j < 8; j++) lds[p8(8 * tt + j)] = r[out8(j)];
    }
    lds_barrier();
#pragma unroll
    for (int j = 0; j < 3; j++) {  // partner of bin j0 + 8 j1 + 64 j2 + 512 j: some other thread's register 7 - j
      const int kf = (tt >> 6) + 8 * ((tt >> 3) & 7) + 64 * (tt & 7) + 512 * j;
      const int kp = (4096 - kf) & 4095;
      const int slot = 512 * (kp & 7) + 64 * ((kp >> 3) & 7) + 8 * ((kp >> 6) & 7) + (kp >> 9);
      const cd y = core::lds_get(lds, p8(slot));
      double pa, pb;
      {
        const cd z = r[out8(j)];
        const double ar = z.x + y.x, ai = z.y - y.y, br = z.y + y.y, bi = y.x - z.x;
        pa = ar * ar + ai * ai;
        pb = br * br + bi * bi;
      }
      keep += pa + pb;
    }
#pragma unroll
    for (int k = 0; k < 8; k++) { r[k].x = r[k].x * 1e-3 + 1.0; r[k].y = r[k].y * 1e-3 + 0.5; }
  }
#pragma unroll
  for (int k = 0; k < 8; k++) keep += r[k].x + r[k].y;
  out[blockIdx.x * 512 + t] = keep;
}

__global__ __launch_bounds__(256, 2) void core16(const cd *__restrict__ tw, double *out, int pairs) {
  extern __shared__ cd lds[];
  const int t = threadIdx.x;
  const cd base0 = tw[t], base1 = tw[16 * (t & 15)];
  cd r[16];
#pragma unroll
  for (int k = 0; k < 16; k++) r[k] = cd{1.0 / (t + k + 1), 0.5 / (t + 2 * k + 1)};
  double keep = 0;
  for (int g = 0; g < pairs; g++) {
    int tt = t;
    asm volatile("" : "+v"(tt));
    core::fft16(r);
    lds_barrier();
    core::dif0_store(tt, base0, lds, r);
    lds_barrier();
    core::dif1(tt, base1, lds, r);
    wave_fence();
    core::dif2(tt, lds, r);
    core::dif2_publish(tt, lds, r);
    lds_barrier();
#pragma unroll
    for (int j = 0; j < 6; j++) {
      double pa, pb;
      core::dif_bin_power_any(tt, j, lds, r, &pa, &pb);
      keep += pa + pb;
    }
#pragma unroll
    for (int k = 0; k < 16; k++) { r[k].x = r[k].x * 1e-3 + 1.0; r[k].y = r[k].y * 1e-3 + 0.5; }
  }
#pragma unroll
  for (int k = 0; k < 16; k++) keep += r[k].x + r[k].y;
  out[blockIdx.x * 256 + t] = keep;
}

int main() {
  const int blocks = 512, pairs = 160;
  std::vector<cd> tw(4096);
  for (int k = 0; k < 4096; k++) tw[k] = cd{std::cos(-2 * M_PI * k / 4096), std::sin(-2 * M_PI * k / 4096)};
  cd *d_tw; double *d_out;
  (void)hipMalloc(&d_tw, 4096 * sizeof(cd)); (void)hipMalloc(&d_out, blocks * 512 * 8);
  (void)hipMemcpy(d_tw, tw.data(), 4096 * sizeof(cd), hipMemcpyHostToDevice);
  const size_t lds8 = (4096 + 512 + 8) * sizeof(cd), lds16 = core::kLds2Slots * sizeof(cd);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(core8), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds8);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(core16), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds16);
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  for (int rep = 0; rep < 3; rep++) {
    float m8, m16;
    (void)hipEventRecord(a); core16<<<blocks, 256, lds16>>>(d_tw, d_out, pairs); (void)hipEventRecord(b); (void)hipEventSynchronize(b); (void)hipEventElapsedTime(&m16, a, b);
    (void)hipEventRecord(a); core8<<<blocks, 512, lds8>>>(d_tw, d_out, pairs); (void)hipEventRecord(b); (void)hipEventSynchronize(b); (void)hipEventElapsedTime(&m8, a, b);
    printf("FFT core of %d pairs: radix 16 x 3, 256 threads: %.3f ms   radix 8 x 4, 512 threads: %.3f ms\n", blocks * pairs, m16, m8);
  }
  return 0;
}
