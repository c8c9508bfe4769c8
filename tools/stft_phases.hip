// Diagnostic (not part of the product): where does one frame pair of stft_chroma_kernel spend its cycles?
// Re-runs the kernel's phases (same fp_core.h code) on random PCM with s_memtime stamps between phases and
// prints average cycles per phase per pair for wave 0 of each workgroup.  Stamps fence the phases, so read the
// SHARES, not the total (cdna_hip_programming.md §7 "In-kernel stamps").
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <vector>

#include "../needle_amd/csrc/fp_core.h"

using needle::core::cd;
namespace core = needle::core;

constexpr int kPhases = 10;

__device__ __forceinline__ unsigned long long stamp() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}

__global__ __launch_bounds__(256, 2) void phases_kernel(const int16_t *__restrict__ pcm, const cd *__restrict__ tw,
                                                        const double *__restrict__ window, int pairs,
                                                        unsigned long long *out, double *sink) {
  extern __shared__ cd lds[];
  const int t = threadIdx.x;
  const cd base0 = tw[t], base1 = tw[16 * (t & 15)];
  unsigned long long acc[kPhases] = {0};
  double keep = 0.0;
  for (int g = 0; g < pairs; g++) {
    const int16_t *src_a = pcm + ((size_t)blockIdx.x * pairs + g) * 2 * 1365;
    const int16_t *src_b = src_a + 1365;
    unsigned long long t0 = stamp();
    const double *wptr = window;
    asm volatile("" : "+s"(wptr));
    cd r[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const int n = t + 256 * k;
      const double w = wptr[n];
      r[k] = cd{(double)src_a[n] * w, (double)src_b[n] * w};
    }
    __builtin_amdgcn_s_waitcnt(0);
    unsigned long long t1 = stamp();
    core::dif0(t, base0, lds, r);
    unsigned long long t2 = stamp();
    __syncthreads();
    unsigned long long t3 = stamp();
    core::dif1(t, base1, lds, r);
    __syncthreads();
    unsigned long long t4 = stamp();
    core::dif2(t, lds, r);
    __syncthreads();
    unsigned long long t5 = stamp();
    core::dif2_publish(t, lds, r);
    __syncthreads();
    unsigned long long t6 = stamp();
    double pa[6], pb[6];
    int kf[6];
#pragma unroll
    for (int j = 0; j < 6; j++) {
      pa[j] = pb[j] = 0.0;
      core::dif_bin_power(t, j, lds, r, &kf[j], &pa[j], &pb[j]);
    }
    __syncthreads();
    unsigned long long t7 = stamp();
    unsigned long long t8 = t7;
#pragma unroll
    for (int i = 0; i < 6; i++) keep += pa[i] + pb[i] + kf[i];
    acc[0] += t1 - t0;  // input loads + window
    acc[1] += t2 - t1;  // stage 0 butterfly + twiddle + LDS write
    acc[2] += t3 - t2;  // barrier
    acc[3] += t4 - t3;  // stage 1 in place + barrier
    acc[4] += t5 - t4;  // stage 2 read + fft16 + barrier
    acc[5] += t6 - t5;  // publish + barrier
    acc[6] += t7 - t6;  // split + power + barrier
    acc[7] += t8 - t7;  // (unused)
  }
  if ((t & 63) == 0)
    for (int p = 0; p < kPhases; p++) out[(blockIdx.x * 4 + (t >> 6)) * kPhases + p] = acc[p];
  sink[blockIdx.x * 256 + t] = keep;
}

int main() {
  const int blocks = 512, pairs = 32;
  std::vector<int16_t> pcm((size_t)blocks * pairs * 2 * 1365 + 8192);
  for (size_t i = 0; i < pcm.size(); i++) pcm[i] = (int16_t)((i * 2654435761u) >> 17);
  std::vector<cd> tw(4096);
  std::vector<double> win(4096);
  for (int k = 0; k < 4096; k++) {
    tw[k] = cd{std::cos(-2 * M_PI * k / 4096), std::sin(-2 * M_PI * k / 4096)};
    win[k] = (0.54 - 0.46 * std::cos(2 * M_PI * k / 4095)) / 32767;
  }
  int16_t *d_pcm; cd *d_tw; double *d_win, *d_sink; unsigned long long *d_out;
  hipMalloc(&d_pcm, pcm.size() * 2); hipMalloc(&d_tw, 4096 * sizeof(cd)); hipMalloc(&d_win, 4096 * 8);
  hipMalloc(&d_sink, blocks * 256 * 8); hipMalloc(&d_out, blocks * 4 * kPhases * 8);
  hipMemcpy(d_pcm, pcm.data(), pcm.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(d_tw, tw.data(), 4096 * sizeof(cd), hipMemcpyHostToDevice);
  hipMemcpy(d_win, win.data(), 4096 * 8, hipMemcpyHostToDevice);
  const size_t lds = core::kLds2Slots * sizeof(cd);
  hipFuncSetAttribute(reinterpret_cast<const void *>(phases_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  phases_kernel<<<blocks, 256, lds>>>(d_pcm, d_tw, d_win, 2, d_out, d_sink);
  hipDeviceSynchronize();
  hipEventRecord(a);
  phases_kernel<<<blocks, 256, lds>>>(d_pcm, d_tw, d_win, pairs, d_out, d_sink);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  std::vector<unsigned long long> out(blocks * 4 * kPhases);
  hipMemcpy(out.data(), d_out, out.size() * 8, hipMemcpyDeviceToHost);
  const char *names[8] = {"input", "stage0+write", "barrier", "stage1 (in place)+barrier", "stage2 read+fft+barrier",
                          "publish+barrier", "split+power+barrier", "-"};
  double tot = 0, sums[8] = {0};
  for (int w = 0; w < blocks * 4; w++)
    for (int p = 0; p < 8; p++) sums[p] += (double)out[w * kPhases + p] / pairs;
  for (int p = 0; p < 8; p++) tot += sums[p] / (blocks * 4);
  printf("stamped kernel: %.3f ms for %d pairs (%.2f us/pair/block)\n", ms, blocks * pairs, 1e3 * ms / pairs);
  for (int p = 0; p < 8; p++) printf("  %-22s %8.0f ticks/pair  %5.1f %%\n", names[p], sums[p] / (blocks * 4), 100 * sums[p] / (blocks * 4) / tot);
  printf("  total %.0f ticks/pair (s_memtime ticks = shader cycles)\n", tot);
  return 0;
}
