// Diagnostic (not part of the product): issue cost of the f64 vector instructions stft_chroma_kernel is made of, at its
// occupancy (256 threads, 2 workgroups per CU = 2 waves per SIMD) and at 1 and 4 waves per SIMD: 16 independent
// accumulators per lane, every operand in VGPRs.  Prints cycles per wave-instruction per SIMD at the measured time
// (2.4 GHz nominal), so a clock held below nominal shows up as more "cycles".
//   hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 tools/f64_rates.hip -o tools/f64_rates
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

template <int OP>
__global__ __launch_bounds__(256) void rate_kernel(double *out, int iters, double x, double y) {
  double a[16];
  int ib[16];
  asm volatile("" : "+v"(x), "+v"(y));
#pragma unroll
  for (int i = 0; i < 16; i++) a[i] = x * (threadIdx.x + i) + 1.0;
#pragma unroll
  for (int i = 0; i < 16; i++) ib[i] = threadIdx.x * i;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 16; i++) {
      if (OP == 7) { ib[i] += it; a[i] = a[i] + (double)ib[i]; }                    // v_add_u32 + v_cvt_f64_i32 + v_add_f64
      if (OP == 8) { ib[i] += it; a[i] = a[i] + x; }                                // v_add_u32 + v_add_f64
      if (OP == 9) { ib[i] = __builtin_amdgcn_sbfe(ib[i] + it, 0, 16); a[i] = a[i] + x; }  // v_add_u32 + v_bfe_i32 + v_add_f64
      if (OP == 0) a[i] = a[i] + x;
      if (OP == 1) a[i] = a[i] * y;
      if (OP == 2) a[i] = __builtin_fma(a[i], y, x);
      if (OP == 3) a[i] = (i & 1) ? __builtin_fma(a[i], y, x) : a[i] + x;          // half fma, half add
      if (OP == 4) a[i] = (i & 1) ? a[i] * y : a[i] + x;                           // half mul, half add
      if (OP == 5) { float f = (float)a[i]; f = __builtin_fmaf(f, 1.0001f, 0.5f); a[i] = (double)f; }  // f32 fma between two converts
      if (OP == 6) a[i] = __builtin_fma(a[i], a[(i + 1) & 15], a[(i + 2) & 15]);   // three different VGPR pairs
    }
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) s += a[i] + ((OP >= 7) ? (double)ib[i] : 0.0);
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// dependent chains: ACC independent accumulators per lane, each instruction depends on the one ACC earlier
template <int ACC, int OP>
__global__ __launch_bounds__(256) void chain_kernel(double *out, int iters, double x, double y) {
  double a[ACC];
  asm volatile("" : "+v"(x), "+v"(y));
#pragma unroll
  for (int i = 0; i < ACC; i++) a[i] = x * (threadIdx.x + i) + 1.0;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int rep = 0; rep < 16 / ACC; rep++)
#pragma unroll
      for (int i = 0; i < ACC; i++) a[i] = OP ? __builtin_fma(a[i], y, x) : a[i] + x;
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < ACC; i++) s += a[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int ACC, int OP>
static void run_chain(double *d_out) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  const int iters = 4000;
  for (int waves_per_simd : {1, 2}) {
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
      hipEventRecord(a);
      chain_kernel<ACC, OP><<<256 * waves_per_simd, 256>>>(d_out, iters, 1.0000001, 0.9999999);
      hipEventRecord(b);
      hipEventSynchronize(b);
      float ms;
      hipEventElapsedTime(&ms, a, b);
      if (rep && ms < best) best = ms;
    }
    std::printf("%s, %2d independent chains per lane       %d wave(s)/SIMD: %.2f cycles per wave-instruction\n", OP ? "v_fma_f64" : "v_add_f64", ACC,
                waves_per_simd, best * 1e-3 * 2.4e9 / ((double)waves_per_simd * iters * 16));
  }
}

template <int OP>
static void run(const char *name, int instr_per_iter, double *d_out) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  const int iters = 4000;
  for (int waves_per_simd : {1, 2, 4}) {
    const int grid = 256 * waves_per_simd;  // 256-thread workgroups: 4 waves, one per SIMD of a CU
    float best = 1e30f;
    for (int rep = 0; rep < 5; rep++) {
      hipEventRecord(a);
      rate_kernel<OP><<<grid, 256>>>(d_out, iters, 1.0000001, 0.9999999);
      hipEventRecord(b);
      hipEventSynchronize(b);
      float ms;
      hipEventElapsedTime(&ms, a, b);
      if (rep && ms < best) best = ms;
    }
    const double instr_per_simd = (double)waves_per_simd * iters * instr_per_iter;
    std::printf("%-44s %d wave(s)/SIMD: %.2f cycles per wave-instruction\n", name, waves_per_simd, best * 1e-3 * 2.4e9 / instr_per_simd);
  }
}

int main() {
  double *d_out;
  hipMalloc(&d_out, 1024 * 256 * sizeof(double));
  run<0>("v_add_f64", 16, d_out);
  run<1>("v_mul_f64", 16, d_out);
  run<2>("v_fma_f64 (one accumulator operand)", 16, d_out);
  run<6>("v_fma_f64 (three VGPR-pair operands)", 16, d_out);
  run<3>("half v_fma_f64, half v_add_f64", 16, d_out);
  run<4>("half v_mul_f64, half v_add_f64", 16, d_out);
  run<8>("v_add_u32 + v_add_f64 (cycles per PAIR)", 16, d_out);
  run<7>("v_add_u32 + v_cvt_f64_i32 + v_add_f64 (cycles per TRIPLE)", 16, d_out);
  run<9>("v_add_u32 + v_bfe_i32 + v_add_f64 (cycles per TRIPLE)", 16, d_out);
  run_chain<1, 1>(d_out);
  run_chain<2, 1>(d_out);
  run_chain<4, 1>(d_out);
  run_chain<8, 1>(d_out);
  run_chain<1, 0>(d_out);
  run_chain<2, 0>(d_out);
  run_chain<4, 0>(d_out);
  return 0;
}
