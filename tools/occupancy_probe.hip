// What would more resident workgroups per CU buy the STFT kernel?  The product's radix-16 x 3 FFT core needs 68 KB
// of LDS per frame pair (4096 complex f64, padded), so two workgroups fit a CU.  This probe runs the same stage
// structure with the LDS image stored as double (NOT a valid fingerprint: a timing probe only), which fits four
// workgroups, at 2, 3 and 4 workgroups per CU (VGPR caps 256 / 168 / 128), next to the f64 image at 2.
//   hipcc -w -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 tools/occupancy_probe.hip -o tools/occupancy_probe
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <vector>

#include "../needle_amd/csrc/fp_core.h"

using needle::core::cd;
namespace core = needle::core;

__device__ __forceinline__ void lds_barrier() { __syncthreads(); }
__device__ __forceinline__ void wave_fence() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); }

template <class S> struct Store;
template <> struct Store<double2> {
  static __device__ __forceinline__ void put(double2 *l, int i, cd v) { l[i] = double2{v.x, v.y}; }
  static __device__ __forceinline__ cd get(const double2 *l, int i) { double2 v = l[i]; return cd{v.x, v.y}; }
};
template <> struct Store<double> {  // half the bytes without conversion instructions: the real part only
  static __device__ __forceinline__ void put(double *l, int i, cd v) { l[i] = v.x + v.y; }
  static __device__ __forceinline__ cd get(const double *l, int i) { double v = l[i]; return cd{v, -v}; }
};

template <class S, int WGS, bool RECOMPUTE>
__global__ __launch_bounds__(256, WGS) void probe(const cd *__restrict__ tw, double *out, int pairs) {
  extern __shared__ char raw[];
  S *lds = reinterpret_cast<S *>(raw);
  const int t = threadIdx.x;
  cd base0 = tw[t], base1 = tw[16 * (t & 15)];
  cd r[16];
#pragma unroll
  for (int k = 0; k < 16; k++) r[k] = cd{1.0 / (t + k + 1), 0.5 / (t + 2 * k + 1)};
  double keep = 0;
  for (int g = 0; g < pairs; g++) {
    int tt = t;
    asm volatile("" : "+v"(tt));
    if (RECOMPUTE)  // too few registers to keep the 30 twiddle powers: recompute them for every pair
      asm volatile("" : "+v"(base0.x), "+v"(base0.y), "+v"(base1.x), "+v"(base1.y));
    core::fft16(r);
    lds_barrier();
    {  // stage 0 store
      Store<S>::put(lds, core::pidx(tt), r[core::out16(0)]);
      cd w = base0;
#pragma unroll
      for (int j = 1; j < 16; j++) {
        Store<S>::put(lds, core::pidx(tt + 256 * j), core::cmulf(r[core::out16(j)], w));
        if (j < 15) w = core::cmulf(w, base0);
      }
    }
    lds_barrier();
    {  // stage 1
      const int o = 256 * (tt >> 4) + (tt & 15);
#pragma unroll
      for (int k = 0; k < 16; k++) r[k] = Store<S>::get(lds, core::pidx(o + 16 * k));
      core::fft16(r);
      Store<S>::put(lds, core::pidx(o), r[core::out16(0)]);
      cd w = base1;
#pragma unroll
      for (int j = 1; j < 16; j++) {
        Store<S>::put(lds, core::pidx(o + 16 * j), core::cmulf(r[core::out16(j)], w));
        if (j < 15) w = core::cmulf(w, base1);
      }
    }
    wave_fence();
#pragma unroll
    for (int k = 0; k < 16; k++) r[k] = Store<S>::get(lds, core::pidx(16 * tt + k));
    core::fft16(r);
#pragma unroll
    for (int j = 10; j < 16; j++) Store<S>::put(lds, core::pidx(16 * tt + j), r[core::out16(j)]);
    lds_barrier();
#pragma unroll
    for (int j = 0; j < 6; j++) {
      const int kf = core::dif_bin_of(tt, j);
      const cd z = r[core::out16(j)], y = Store<S>::get(lds, core::pidx(core::dif_slot_of_bin((core::kFft2N - kf) & 4095)));
      const double ar = z.x + y.x, ai = z.y - y.y, br = z.y + y.y, bi = y.x - z.x;
      keep += ar * ar + ai * ai + br * br + bi * bi;
    }
#pragma unroll
    for (int k = 0; k < 16; k++) { r[k].x = r[k].x * 1e-3 + 1.0; r[k].y = r[k].y * 1e-3 + 0.5; }
  }
#pragma unroll
  for (int k = 0; k < 16; k++) keep += r[k].x + r[k].y;
  out[blockIdx.x * 256 + t] = keep;
}

template <class S, int WGS, bool RECOMPUTE>
float run(const cd *d_tw, double *d_out, int blocks, int pairs) {
  const size_t lds = core::kLds2Slots * sizeof(S);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(probe<S, WGS, RECOMPUTE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  int resident = 0;
  (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&resident, probe<S, WGS, RECOMPUTE>, 256, lds);
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  float best = 1e9f;
  for (int rep = 0; rep < 4; rep++) {
    (void)hipEventRecord(a);
    probe<S, WGS, RECOMPUTE><<<blocks, 256, lds>>>(d_tw, d_out, pairs);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    if (rep && ms < best) best = ms;
  }
  printf("  %s image, launch_bounds(256,%d), twiddle powers %s: %d workgroups resident per CU, %.3f ms\n",
         sizeof(S) == 16 ? "16-byte" : "8-byte", WGS, RECOMPUTE ? "recomputed per pair" : "kept in registers", resident, best);
  return best;
}

int main() {
  const int blocks = 3072, pairs = 27;  // 82 944 frame pairs; 3072 = 256 CUs x 12: whole waves of 2, 3 and 4 workgroups per CU
  std::vector<cd> tw(4096);
  for (int k = 0; k < 4096; k++) tw[k] = cd{std::cos(-2 * M_PI * k / 4096), std::sin(-2 * M_PI * k / 4096)};
  cd *d_tw; double *d_out;
  (void)hipMalloc(&d_tw, 4096 * sizeof(cd)); (void)hipMalloc(&d_out, (size_t)blocks * 256 * 8);
  (void)hipMemcpy(d_tw, tw.data(), 4096 * sizeof(cd), hipMemcpyHostToDevice);
  printf("FFT core of %d frame pairs\n", blocks * pairs);
  run<double2, 2, false>(d_tw, d_out, blocks, pairs);
  run<double2, 2, true>(d_tw, d_out, blocks, pairs);
  run<double, 2, false>(d_tw, d_out, blocks, pairs);
  run<double, 2, true>(d_tw, d_out, blocks, pairs);
  run<double, 3, true>(d_tw, d_out, blocks, pairs);
  run<double, 4, true>(d_tw, d_out, blocks, pairs);
  return 0;
}
