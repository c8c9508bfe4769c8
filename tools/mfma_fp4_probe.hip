// Diagnostic (not product): can the FP4 form of gfx950's block-scaled matrix instruction count differing BITS?
//
// The scan's matrix-pipe first stage (needle_amd/csrc/scan_mfma_kernel.h) multiplies hashes expanded to 32 bytes of +-1 with
// v_mfma_i32_32x32x32_i8: one hash row per instruction, 32 cycles.  v_mfma_scale_f32_32x32x64_f8f6f4 with both operands
// FP4 (e2m1: +1 = 0x2, -1 = 0xA, exact) takes K = 64 in the cycles of the bf16 form (MI355X_MICROARCH.md, matrix cores):
// TWO hash rows per instruction, 16 bytes per lane and operand, f32 accumulators (exact: |sum| <= 64 + preset).
//
// Checked here with exact data, one wave:
//   1. value: lane l (r = l & 31, h = l >> 5) holds nibbles j = 0 .. 31 of A[r][32 h + j] / B[32 h + j][r] -- for a dot
//      product over k any permutation of j inside a lane is the same on both sides; what matters is that the lanes of
//      half h of A meet the lanes of half h of B.  Random +-1 (and 0) nibbles, asymmetric, against the host's sums;
//      scale operands 0x7F (2^0) and 0 (the compiler then picks the unscaled instruction).
//   2. D layout: col = l & 31, row = (reg & 3) + 8 (reg >> 2) + 4 h (the guide: shape-determined).
//   3. time: cycles per instruction (s_memtime), back-to-back on independent accumulators, one wave per SIMD, against the
//      i8 instruction.
//
// build: hipcc -O2 -std=c++17 --offload-arch=gfx950 tools/mfma_fp4_probe.hip -o tools/mfma_fp4_probe
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                              \
  do {                                                                                     \
    hipError_t e_ = (x);                                                                   \
    if (e_ != hipSuccess) {                                                                \
      std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));      \
      std::exit(1);                                                                        \
    }                                                                                      \
  } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v16f __attribute__((ext_vector_type(16)));

template <int SCALE>
__device__ inline v16f fp4_product(v4i a, v4i b, v16f c) {
  const v8i a8 = {a[0], a[1], a[2], a[3], 0, 0, 0, 0};
  const v8i b8 = {b[0], b[1], b[2], b[3], 0, 0, 0, 0};
  return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c, 4, 4, 0, SCALE, 0, SCALE);
}

template <int SCALE>
__global__ __launch_bounds__(64) void one_product(const v4i *a, const v4i *b, float preset, float *d) {
  const int l = threadIdx.x;
  v16f c;
  for (int r = 0; r < 16; r++) c[r] = preset;
  c = fp4_product<SCALE>(a[l], b[l], c);
  for (int r = 0; r < 16; r++) d[l * 16 + r] = c[r];
}

template <int KIND>   // 0: i8 32x32x32, 1: fp4 scaled by 2^0, 2: fp4 with zero scale operands
__global__ __launch_bounds__(256) void timing(const v4i *a, const v4i *b, long long *cycles, float *sink, int reps) {
  const int l = threadIdx.x & 63;
  const v4i fa = a[l], fb = b[l];
  v16f c0, c1, c2, c3;
  v16i i0, i1, i2, i3;
  for (int r = 0; r < 16; r++) c0[r] = c1[r] = c2[r] = c3[r] = 1.0f, i0[r] = i1[r] = i2[r] = i3[r] = 1;
  const long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < reps; i++) {
    if (KIND == 0) {
      i0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa, fb, i0, 0, 0, 0);
      i1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa, fb, i1, 0, 0, 0);
      i2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa, fb, i2, 0, 0, 0);
      i3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa, fb, i3, 0, 0, 0);
    } else if (KIND == 1) {
      c0 = fp4_product<0x7F7F7F7F>(fa, fb, c0);
      c1 = fp4_product<0x7F7F7F7F>(fa, fb, c1);
      c2 = fp4_product<0x7F7F7F7F>(fa, fb, c2);
      c3 = fp4_product<0x7F7F7F7F>(fa, fb, c3);
    } else {
      c0 = fp4_product<0>(fa, fb, c0);
      c1 = fp4_product<0>(fa, fb, c1);
      c2 = fp4_product<0>(fa, fb, c2);
      c3 = fp4_product<0>(fa, fb, c3);
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int r = 0; r < 16; r++) s += c0[r] + c1[r] + c2[r] + c3[r] + (float)(i0[r] + i1[r] + i2[r] + i3[r]);
  sink[blockIdx.x * 256 + threadIdx.x] = s;
  if (l == 0) cycles[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

static float nibble_value(uint32_t n) {
  static const float mag[8] = {0.f, 0.5f, 1.f, 1.5f, 2.f, 3.f, 4.f, 6.f};
  return (n & 8 ? -1.f : 1.f) * mag[n & 7];
}

template <int SCALE>
static int check_value(const char *name, bool with_zeros) {
  // A [32][64], B [64][32] as nibbles
  std::vector<uint32_t> A(32 * 64), B(64 * 32);
  uint32_t s = 777u + (with_zeros ? 5u : 0u);
  auto rnd = [&] { s = s * 1664525u + 1013904223u; return s >> 16; };
  for (auto &x : A) { const uint32_t r = rnd() % (with_zeros ? 3 : 2); x = r == 0 ? 0x2u : r == 1 ? 0xAu : 0x0u; }
  for (auto &x : B) { const uint32_t r = rnd() % (with_zeros ? 3 : 2); x = r == 0 ? 0x2u : r == 1 ? 0xAu : 0x0u; }
  std::vector<uint32_t> fa(64 * 4, 0), fb(64 * 4, 0);
  for (int l = 0; l < 64; l++) {
    const int r = l & 31, h = l >> 5;
    for (int j = 0; j < 32; j++) {
      fa[l * 4 + j / 8] |= A[r * 64 + 32 * h + j] << (4 * (j % 8));
      fb[l * 4 + j / 8] |= B[(32 * h + j) * 32 + r] << (4 * (j % 8));
    }
  }
  uint32_t *da, *db;
  float *dd;
  CK(hipMalloc(&da, fa.size() * 4));
  CK(hipMalloc(&db, fb.size() * 4));
  CK(hipMalloc(&dd, 64 * 16 * 4));
  CK(hipMemcpy(da, fa.data(), fa.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(db, fb.data(), fb.size() * 4, hipMemcpyHostToDevice));
  const float preset = 63.f - 4.f * 5.f;
  one_product<SCALE><<<1, 64>>>((const v4i *)da, (const v4i *)db, preset, dd);
  CK(hipDeviceSynchronize());
  std::vector<float> D(64 * 16);
  CK(hipMemcpy(D.data(), dd, D.size() * 4, hipMemcpyDeviceToHost));
  int bad = 0;
  for (int l = 0; l < 64; l++)
    for (int reg = 0; reg < 16; reg++) {
      const int col = l & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (l >> 5);
      float want = preset;
      for (int k = 0; k < 64; k++) want += nibble_value(A[row * 64 + k]) * nibble_value(B[k * 32 + col]);
      if (D[l * 16 + reg] != want) {
        if (bad < 4) std::printf("  %s: lane %d reg %d got %g want %g\n", name, l, reg, D[l * 16 + reg], want);
        bad++;
      }
    }
  std::printf("%-44s %s (%d of 1024 wrong)\n", name, bad ? "WRONG" : "exact", bad);
  CK(hipFree(da)); CK(hipFree(db)); CK(hipFree(dd));
  return bad;
}

template <int KIND>
static void time_kind(const char *name) {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int blocks = prop.multiProcessorCount, reps = 4096;
  uint32_t *da, *db;
  long long *dc;
  float *ds;
  std::vector<uint32_t> f(64 * 4, 0x2A2A2A2Au);
  CK(hipMalloc(&da, f.size() * 4));
  CK(hipMalloc(&db, f.size() * 4));
  CK(hipMalloc(&dc, blocks * 4 * 8));
  CK(hipMalloc(&ds, blocks * 256 * 4));
  CK(hipMemcpy(da, f.data(), f.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(db, f.data(), f.size() * 4, hipMemcpyHostToDevice));
  for (int pass = 0; pass < 2; pass++) {
    timing<KIND><<<blocks, 256>>>((const v4i *)da, (const v4i *)db, dc, ds, reps);
    CK(hipDeviceSynchronize());
  }
  std::vector<long long> c(blocks * 4);
  CK(hipMemcpy(c.data(), dc, c.size() * 8, hipMemcpyDeviceToHost));
  double sum = 0;
  for (auto v : c) sum += (double)v;
  // s_memtime counts at 100 MHz on this part; __builtin_readcyclecounter is s_memtime: report both raw ticks and the
  // ratio to the i8 instruction (the caller prints the ratio)
  std::printf("%-44s %.3f ticks per instruction (one wave per SIMD, every CU busy)\n", name, sum / c.size() / (4.0 * reps));
  CK(hipFree(da)); CK(hipFree(db)); CK(hipFree(dc)); CK(hipFree(ds));
}

int main() {
  int bad = 0;
  bad += check_value<0x7F7F7F7F>("fp4 32x32x64, scale 2^0, +-1", false);
  bad += check_value<0x7F7F7F7F>("fp4 32x32x64, scale 2^0, +-1 and 0", true);
  const int unscaled = check_value<0>("fp4 32x32x64, scale operands 0, +-1 and 0", true);
  std::printf("(zero scale operands %s)\n", unscaled ? "do NOT mean 2^0" : "mean 2^0 as well");
  time_kind<0>("v_mfma_i32_32x32x32_i8");
  time_kind<1>("v_mfma_scale_f32_32x32x64_f8f6f4 fp4 (0x7F)");
  time_kind<2>("v_mfma_f32_32x32x64_f8f6f4 fp4 (0)");
  return bad ? 1 : 0;
}
