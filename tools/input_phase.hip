// Diagnostic: cost of the STFT input phase variants (cycles per pair per wave, s_memtime), MI355X.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>

__device__ __forceinline__ unsigned long long stamp() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}

// MODE 0: 32 short loads + 16 window loads (as in the kernel)   1: shorts only (window = const)
// MODE 2: window only (samples = const)                          3: aligned dword loads of sample pairs + window
// MODE 4: shorts with 16-bit loads but frames 4-byte aligned      5: shorts only, no (double) conversion (int sum)
template <int MODE>
__global__ __launch_bounds__(256, 2) void k(const int16_t *__restrict__ pcm, const double *__restrict__ window,
                                            int pairs, unsigned long long *out, double *sink, int hop) {
  const int t = threadIdx.x;
  unsigned long long acc = 0;
  double keep = 0;
  for (int g = 0; g < pairs; g++) {
    const int16_t *src_a = pcm + ((size_t)blockIdx.x * pairs + g) * 2 * hop;
    const int16_t *src_b = src_a + hop;
    unsigned long long t0 = stamp();
    double s = 0;
    if (MODE == 3) {
      const int *pa = reinterpret_cast<const int *>(src_a), *pb = reinterpret_cast<const int *>(src_b);
#pragma unroll
      for (int k2 = 0; k2 < 8; k2++) {
        const int n = t + 256 * k2;
        const int va = pa[n], vb = pb[n];
        s += (double)(short)va * window[2 * n] + (double)(va >> 16) * window[2 * n + 1] + (double)(short)vb + (double)(vb >> 16);
      }
    } else {
#pragma unroll
      for (int k2 = 0; k2 < 16; k2++) {
        const int n = t + 256 * k2;
        const int sa = (MODE == 2) ? n : src_a[n], sb = (MODE == 2) ? n + 1 : src_b[n];
        if (MODE == 5) { s += (double)(sa + sb); continue; }
        const double w = (MODE == 1) ? 0.5 : window[n];
        s += (double)sa * w + (double)sb * w;
      }
    }
    __builtin_amdgcn_s_waitcnt(0);
    keep += s;
    unsigned long long t1 = stamp();
    acc += t1 - t0;
    __syncthreads();
  }
  if ((t & 63) == 0) out[blockIdx.x * 4 + (t >> 6)] = acc;
  sink[blockIdx.x * 256 + t] = keep;
}

template <int MODE>
void run(const char *name, int16_t *d_pcm, double *d_win, unsigned long long *d_out, double *d_sink, int hop) {
  const int blocks = 512, pairs = 32;
  k<MODE><<<blocks, 256>>>(d_pcm, d_win, 2, d_out, d_sink, hop);
  hipDeviceSynchronize();
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipEventRecord(a);
  k<MODE><<<blocks, 256>>>(d_pcm, d_win, pairs, d_out, d_sink, hop);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  std::vector<unsigned long long> out(blocks * 4);
  hipMemcpy(out.data(), d_out, out.size() * 8, hipMemcpyDeviceToHost);
  double sum = 0; for (auto v : out) sum += (double)v / pairs;
  printf("%-48s hop=%d  %8.0f cycles/pair/wave   kernel %.3f ms\n", name, hop, sum / out.size(), ms);
}

int main() {
  const size_t n = (size_t)512 * 32 * 2 * 1366 + 16384;
  std::vector<int16_t> pcm(n);
  for (size_t i = 0; i < n; i++) pcm[i] = (int16_t)((i * 2654435761u) >> 17);
  std::vector<double> win(4096, 0.25);
  int16_t *d_pcm; double *d_win, *d_sink; unsigned long long *d_out;
  hipMalloc(&d_pcm, n * 2); hipMalloc(&d_win, 4096 * 8); hipMalloc(&d_sink, 512 * 256 * 8); hipMalloc(&d_out, 512 * 4 * 8);
  hipMemcpy(d_pcm, pcm.data(), n * 2, hipMemcpyHostToDevice);
  hipMemcpy(d_win, win.data(), 4096 * 8, hipMemcpyHostToDevice);
  run<0>("shorts + window (kernel as is)", d_pcm, d_win, d_out, d_sink, 1365);
  run<1>("shorts only", d_pcm, d_win, d_out, d_sink, 1365);
  run<2>("window only", d_pcm, d_win, d_out, d_sink, 1365);
  run<5>("shorts only, integer sum", d_pcm, d_win, d_out, d_sink, 1365);
  run<1>("shorts only, even hop (4-byte aligned frames)", d_pcm, d_win, d_out, d_sink, 1366);
  run<3>("dword loads (2 samples), aligned", d_pcm, d_win, d_out, d_sink, 1366);
  return 0;
}
