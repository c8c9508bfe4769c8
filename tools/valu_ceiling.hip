// Micro-benchmark: the integer-VALU ceiling of the search kernel's per-cell work on this GPU
// ("calibrate with a popcount micro-benchmark on the box").  Each "cell" is exactly the
// four instructions of hamming_runs_band_kernel's inner loop: v_xor_b32, v_bcnt_u32_b32, v_cmp_lt_u32,
// v_cndmask_b32 — on registers only, no memory.  Prints cells/s for several occupancies.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

template <int R>
__global__ __launch_bounds__(256) void cells_kernel(uint32_t *out, int rows, uint32_t threshold, uint32_t seed) {
  uint32_t W[R];
  int Z[R];
#pragma unroll
  for (int r = 0; r < R; r++) {
    W[r] = seed * (threadIdx.x + 1) * (r + 3) + blockIdx.x;
    Z[r] = -1;
  }
  uint32_t sv = __builtin_amdgcn_readfirstlane(seed ^ blockIdx.x);
  for (int i = 0; i < rows; i++) {
    sv = sv * 1664525u + 1013904223u;  // scalar (SALU) update: stands in for the s_load'ed src[i]
#pragma unroll
    for (int r = 0; r < R; r++) {
      const uint32_t c = (uint32_t)__popc(sv ^ W[r]);
      Z[r] = (c <= threshold) ? Z[r] : i;
    }
  }
  uint32_t acc = 0;
#pragma unroll
  for (int r = 0; r < R; r++) acc += (uint32_t)Z[r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main() {
  constexpr int R = 8;
  const int rows = 20000;
  uint32_t *out;
  hipMalloc(&out, 256 * 64 * 256 * sizeof(uint32_t));
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int blocks_per_cu : {1, 2, 4, 8}) {
    const int grid = 256 * blocks_per_cu;
    cells_kernel<R><<<grid, 256>>>(out, 100, 10, 12345u);
    hipDeviceSynchronize();
    hipEventRecord(a);
    cells_kernel<R><<<grid, 256>>>(out, rows, 10, 12345u);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double cells = (double)grid * 256 * R * rows;
    printf("waves/SIMD=%d  %.3f ms  %.3e cells/s  (%.2f cells/clk/CU at 2.4 GHz)\n", blocks_per_cu, ms,
           cells / (ms * 1e-3), cells / (ms * 1e-3) / 256 / 2.4e9);
  }
  return 0;
}
