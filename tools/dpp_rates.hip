// Diagnostic (not part of the product): issue cost of v_fmac_f32 with a DPP row-broadcast source operand against the
// plain instruction, 4 independent accumulators per lane as in resample_quad_kernel, at 1 / 2 / 4 waves per SIMD; and of
// v_pk_fma_f32 (two FMAs per lane and instruction, no DPP form: the coefficients then have to sit in every lane).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/dpp_rates.hip -o tools/dpp_rates
#include <hip/hip_runtime.h>

#include <cstdio>

template <int OP>
__global__ __launch_bounds__(256) void rate_kernel(float *out, int iters, float x, float y) {
  float a[4] = {x, x + 1, x + 2, x + 3};
  float c = y * (threadIdx.x & 15);
  asm volatile("" : "+v"(c), "+v"(x));
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 16; r++)
#pragma unroll
      for (int q = 0; q < 4; q++) {
        if (OP == 0) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(a[q]) : "v"(c), "v"(x));
        if (OP == 1) asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(a[q]) : "v"(c), "v"(x));
        if (OP == 2) asm volatile("v_fmac_f32_dpp %0, %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[q]) : "v"(c), "v"(x));
        if (OP == 3) asm volatile("v_fmac_f32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[q]) : "v"(c), "v"(x));
      }
  }
  out[blockIdx.x * 256 + threadIdx.x] = a[0] + a[1] + a[2] + a[3];
}

typedef float v2f __attribute__((ext_vector_type(2)));
// OP 0: acc.lo += c.lo x.lo, acc.hi += c.hi x.hi; OP 1: both halves take x.lo (op_sel_hi), what a filter needs: two
// outputs, one sample; OP 2: v_fma_f32 in its VOP3 encoding (three VGPRs), for the cost of the 64-bit encoding alone
template <int OP>
__global__ __launch_bounds__(256) void pk_rate_kernel(float *out, int iters, float x, float y) {
  v2f a[4] = {{x, x + 1}, {x + 2, x + 3}, {x + 4, x + 5}, {x + 6, x + 7}};
  v2f c = {y * (threadIdx.x & 15), y}, xx = {x, x * 0.5f};
  asm volatile("" : "+v"(c), "+v"(xx));
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 16; r++)
#pragma unroll
      for (int q = 0; q < 4; q++) {
        if (OP == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[q]) : "v"(c), "v"(xx));
        if (OP == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(a[q]) : "v"(c), "v"(xx));
        if (OP == 2) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[q].x) : "v"(c.x), "v"(xx.x));
      }
  }
  out[blockIdx.x * 256 + threadIdx.x] = a[0].x + a[1].x + a[2].x + a[3].x + a[0].y + a[1].y + a[2].y + a[3].y;
}

template <int OP>
static void run(const char *name, float *d_out) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  const int iters = 4000;
  for (int waves_per_simd : {1, 2, 4}) {
    float best = 1e30f;
    for (int rep = 0; rep < 5; rep++) {
      (void)hipEventRecord(a);
      rate_kernel<OP><<<256 * waves_per_simd, 256>>>(d_out, iters, 1.0000001f, 0.9999999f);
      (void)hipEventRecord(b);
      (void)hipEventSynchronize(b);
      float ms;
      (void)hipEventElapsedTime(&ms, a, b);
      if (rep && ms < best) best = ms;
    }
    std::printf("%-40s %d wave(s)/SIMD: %.2f cycles per wave-instruction\n", name, waves_per_simd,
                best * 1e-3 * 2.4e9 / ((double)waves_per_simd * iters * 64));
  }
}

template <int OP>
static void run_pk(const char *name, float *d_out) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  const int iters = 4000;
  for (int waves_per_simd : {1, 2, 4}) {
    float best = 1e30f;
    for (int rep = 0; rep < 5; rep++) {
      (void)hipEventRecord(a);
      pk_rate_kernel<OP><<<256 * waves_per_simd, 256>>>(d_out, iters, 1.0000001f, 0.9999999f);
      (void)hipEventRecord(b);
      (void)hipEventSynchronize(b);
      float ms;
      (void)hipEventElapsedTime(&ms, a, b);
      if (rep && ms < best) best = ms;
    }
    std::printf("%-40s %d wave(s)/SIMD: %.2f cycles per wave-instruction\n", name, waves_per_simd,
                best * 1e-3 * 2.4e9 / ((double)waves_per_simd * iters * 64));
  }
}

int main() {
  float *d_out;
  (void)hipMalloc(&d_out, 1024 * 256 * sizeof(float));
  run<0>("v_fmac_f32", d_out);
  run<1>("v_fmac_f32_dpp row_newbcast", d_out);
  run<2>("v_fmac_f32_dpp quad_perm", d_out);
  run<3>("v_fmac_f32_dpp row_shr", d_out);
  run_pk<2>("v_fma_f32 (VOP3, three VGPRs)", d_out);
  run_pk<0>("v_pk_fma_f32 (2 FMAs per lane)", d_out);
  run_pk<1>("v_pk_fma_f32, one sample for both", d_out);
  return 0;
}
