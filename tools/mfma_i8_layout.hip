// Diagnostic (not product): operand layouts of the int8 matrix instructions on gfx950, checked with exact integer data
// as the guide asks ("other dtypes: check the map with exact integer data before relying on it").  Preparation for a
// matrix-pipe first stage of the sampled scan (tools/mfma_filter_model.py).
//
// Hypotheses (the bf16 maps with twice the elements per lane):
//   v_mfma_i32_32x32x32_i8: lane l, r = l & 31, h = l >> 5: byte j (0..15) of its 16-byte fragment is A[r][16 h + j] / B[16 h + j][r];
//                           D: col = l & 31, row = (reg & 3) + 8 (reg >> 2) + 4 h, 16 registers
//   v_mfma_i32_16x16x64_i8: lane l, r = l & 15, g = l >> 4: byte j is A[r][16 g + j] / B[16 g + j][r];
//                           D: col = l & 15, row = 4 g + reg, 4 registers
// One wave, one instruction each, random int8 matrices (asymmetric), compared with the product computed on the host.
// If a hypothesis fails, the same data is used to FIND the k index of every (lane, byte) by brute force: one-hot A
// elements against one-hot B elements (D is non-zero exactly where the two k indices agree).
//
// build: hipcc -O2 -std=c++17 --offload-arch=gfx950 tools/mfma_i8_layout.hip -o tools/mfma_i8_layout
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
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
typedef int v16i __attribute__((ext_vector_type(16)));

// a, b: 64 lanes x 16 bytes; d: 64 lanes x 16 (or 4) ints
__global__ __launch_bounds__(64) void mfma_32x32x32(const v4i *a, const v4i *b, int *d) {
  const int l = threadIdx.x;
  v16i c = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[l], b[l], c, 0, 0, 0);
  for (int r = 0; r < 16; r++) d[l * 16 + r] = c[r];
}
__global__ __launch_bounds__(64) void mfma_16x16x64(const v4i *a, const v4i *b, int *d) {
  const int l = threadIdx.x;
  v4i c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[l], b[l], c, 0, 0, 0);
  for (int r = 0; r < 4; r++) d[l * 4 + r] = c[r];
}
// brute force: for every A element p (wave-uniform loop) and every B element q: is D non-zero anywhere?  out[p * 1024 + q]
// = 1 + index (lane * regs + reg) of the non-zero D element, 0 if none.  Bounded loops, 1 M instructions, one wave.
template <int REGS>
__global__ __launch_bounds__(64) void brute(uint16_t *out) {
  const int l = threadIdx.x;
  for (int p = 0; p < 1024; p++) {
    v4i a = {0, 0, 0, 0};
    if ((p >> 4) == l) a[(p & 15) >> 2] = 1 << (8 * (p & 3));
    for (int q = 0; q < 1024; q++) {
      v4i b = {0, 0, 0, 0};
      if ((q >> 4) == l) b[(q & 15) >> 2] = 1 << (8 * (q & 3));
      int hit = 0;
      if (REGS == 16) {
        v16i c = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0);
        for (int r = 0; r < 16; r++)
          if (c[r]) hit = 1 + l * 16 + r;
      } else {
        v4i c = {0, 0, 0, 0};
        c = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0);
        for (int r = 0; r < 4; r++)
          if (c[r]) hit = 1 + l * 4 + r;
      }
      // at most one lane holds the non-zero element: a wave-wide maximum brings it to lane 0
      for (int off = 32; off > 0; off >>= 1) hit = max(hit, __shfl_xor(hit, off));
      if (l == 0) out[p * 1024 + q] = (uint16_t)hit;
    }
  }
}

static int check(const char *name, int M, int N, int K, bool big) {
  std::vector<int8_t> A((size_t)M * K), B((size_t)K * N);
  uint32_t s = 12345u + (uint32_t)K;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (int8_t)((int)(s >> 24) - 128); };
  for (auto &v : A) v = rnd();
  for (auto &v : B) v = rnd();
  std::vector<int8_t> fa(64 * 16), fb(64 * 16);
  for (int l = 0; l < 64; l++)
    for (int j = 0; j < 16; j++) {
      const int r = big ? (l & 31) : (l & 15), h = big ? (l >> 5) : (l >> 4);
      fa[l * 16 + j] = A[(size_t)r * K + 16 * h + j];
      fb[l * 16 + j] = B[(size_t)(16 * h + j) * N + r];
    }
  int8_t *da, *db;
  int *dd;
  const int regs = big ? 16 : 4;
  CK(hipMalloc(&da, 1024));
  CK(hipMalloc(&db, 1024));
  CK(hipMalloc(&dd, 64 * 16 * 4));
  CK(hipMemcpy(da, fa.data(), 1024, hipMemcpyHostToDevice));
  CK(hipMemcpy(db, fb.data(), 1024, hipMemcpyHostToDevice));
  if (big) hipLaunchKernelGGL(mfma_32x32x32, dim3(1), dim3(64), 0, 0, (const v4i *)da, (const v4i *)db, dd);
  else hipLaunchKernelGGL(mfma_16x16x64, dim3(1), dim3(64), 0, 0, (const v4i *)da, (const v4i *)db, dd);
  CK(hipDeviceSynchronize());
  std::vector<int> D(64 * 16);
  CK(hipMemcpy(D.data(), dd, (size_t)64 * regs * 4, hipMemcpyDeviceToHost));
  int bad = 0;
  for (int l = 0; l < 64; l++)
    for (int r = 0; r < regs; r++) {
      const int col = big ? (l & 31) : (l & 15);
      const int row = big ? (r & 3) + 8 * (r >> 2) + 4 * (l >> 5) : 4 * (l >> 4) + r;
      long want = 0;
      for (int k = 0; k < K; k++) want += (long)A[(size_t)row * K + k] * B[(size_t)k * N + col];
      if (want != D[l * regs + r]) bad++;
    }
  std::printf("%s: %s (%d of %d elements differ from the host product under the hypothesised maps)\n", name, bad ? "FAIL" : "PASS", bad,
              64 * regs);
  CK(hipFree(da));
  CK(hipFree(db));
  CK(hipFree(dd));
  return bad;
}

template <int REGS>
static void find_maps(const char *name) {
  uint16_t *dout;
  CK(hipMalloc(&dout, (size_t)1024 * 1024 * 2));
  CK(hipMemset(dout, 0, (size_t)1024 * 1024 * 2));
  hipLaunchKernelGGL(brute<REGS>, dim3(1), dim3(64), 0, 0, dout);
  CK(hipDeviceSynchronize());
  std::vector<uint16_t> out((size_t)1024 * 1024);
  CK(hipMemcpy(out.data(), dout, out.size() * 2, hipMemcpyDeviceToHost));
  CK(hipFree(dout));
  // k classes: A element p and B element q share k iff out[p][q] != 0.  Print, for A element p, the B elements of its class
  // as (lane, byte) -- enough to read the map off by eye for the first lanes.
  std::printf("%s: B elements (lane.byte) that meet A element (lane.byte) in one product term, first 40 A elements:\n", name);
  for (int p = 0; p < 40; p++) {
    std::printf("  A %2d.%-2d :", p >> 4, p & 15);
    int shown = 0;
    for (int q = 0; q < 1024 && shown < 6; q++)
      if (out[(size_t)p * 1024 + q]) {
        std::printf(" %2d.%-2d->D[%d]", q >> 4, q & 15, out[(size_t)p * 1024 + q] - 1);
        shown++;
      }
    std::printf(" ...\n");
  }
}

// issue cadence: cycles (s_memtime) per v_mfma_i32_32x32x32_i8 of one wave, four independent accumulators, 256 in a row;
// several waves per SIMD (the workgroups are 256 threads = one wave per SIMD, `wgs` of them per CU are resident together)
__global__ __launch_bounds__(256) void mfma_rate(const v4i *a, const v4i *b, int *sink, unsigned long long *cycles) {
  const int l = threadIdx.x & 63;
  v4i fa = a[l], fb = b[l];
  v16i c0 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  __builtin_amdgcn_s_barrier();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < 64; i++) {
    c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa, fb, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa, fb, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa, fb, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa, fb, c3, 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  int x = 0;
  for (int r = 0; r < 16; r++) x += c0[r] + c1[r] + c2[r] + c3[r];
  sink[blockIdx.x * 256 + threadIdx.x] = x;
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

static void rate(const int blocks) {
  int8_t *da, *db;
  int *sink;
  unsigned long long *cyc;
  CK(hipMalloc(&da, 1024));
  CK(hipMalloc(&db, 1024));
  CK(hipMemset(da, 1, 1024));
  CK(hipMemset(db, 1, 1024));
  CK(hipMalloc(&sink, (size_t)blocks * 256 * 4));
  CK(hipMalloc(&cyc, (size_t)blocks * 8));
  for (int rep = 0; rep < 2; rep++) {
    hipLaunchKernelGGL(mfma_rate, dim3(blocks), dim3(256), 0, 0, (const v4i *)da, (const v4i *)db, sink, cyc);
    CK(hipDeviceSynchronize());
  }
  std::vector<unsigned long long> h(blocks);
  CK(hipMemcpy(h.data(), cyc, (size_t)blocks * 8, hipMemcpyDeviceToHost));
  std::sort(h.begin(), h.end());
  std::printf("v_mfma_i32_32x32x32_i8 issue cadence, 256 per wave, %d workgroups of 4 waves on the device: min %.1f  median %.1f  max %.1f cycles per instruction of one wave\n",
              blocks, h.front() / 256.0, h[h.size() / 2] / 256.0, h.back() / 256.0);
  CK(hipFree(da));
  CK(hipFree(db));
  CK(hipFree(sink));
  CK(hipFree(cyc));
}

int main() {
  rate(256);
  rate(1024);
  const int bad32 = check("v_mfma_i32_32x32x32_i8", 32, 32, 32, true);
  const int bad16 = check("v_mfma_i32_16x16x64_i8", 16, 16, 64, false);
  if (bad32) find_maps<16>("v_mfma_i32_32x32x32_i8");
  if (bad16) find_maps<4>("v_mfma_i32_16x16x64_i8");
  return 0;
}
