// Laboratory for round 6's question: do the three radix-16 DFT stages of the f32 first pass belong on
// the matrix pipe?  Not part of the product.  Three parts:
//   probe   v_mfma_f32_32x32x16_f16: operand / result lane maps with exact integer data, whether f16 SUBNORMAL operands are
//           flushed (the lo halves of a split operand live there), what the accumulation rounds
//   verify  the transform core below -- one 4096-point complex transform per frame pair as three products
//           Y[32 x 256] = F[32 x 32] X[32 x 256] with x = hi + lo (two f16), F = Fh + Fl, products Fh lo + Fl hi + Fh hi,
//           f32 accumulators, twiddles and the 2^-4 per stage on the vector ALU, two exchanges through LDS -- against an
//           f64 transform on the host (relative l2 error; tools/mfma_dft_model.py predicts ~1.5e-7)
//   time    the core on BASELINE.json configs[1]'s launch shape (28 streams x 5 813 frames = 81 396 frame pairs),
//           variants taking turns: what tools/stft32_lab's "transform alone" (0.327 of 0.455 ms) is compared with
// Build:  hipcc -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize --offload-arch=gfx950 tools/stft_mfma_lab.hip -o tools/stft_mfma_lab
// Run:    tools/stft_mfma_lab [probe] [verify] [time]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "stft_mfma_core.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

using namespace mfmalab;

// ---- probe ------------------------------------------------------------------------------------------------------------
__global__ void probe_kernel(const half8 *a, const half8 *b, float *d) {
  const int l = threadIdx.x;
  f32x16 acc = {};
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[l], b[l], acc, 0, 0, 0);
  for (int i = 0; i < 16; i++) d[16 * l + i] = acc[i];
}

static uint16_t f16_bits(float f) {  // round to nearest even, subnormals kept
  _Float16 h = (_Float16)f;
  uint16_t u;
  std::memcpy(&u, &h, 2);
  return u;
}

static int probe() {
  half8 *da, *db;
  float *dd;
  CK(hipMalloc(&da, 64 * 16));
  CK(hipMalloc(&db, 64 * 16));
  CK(hipMalloc(&dd, 64 * 16 * 4));
  std::vector<uint16_t> a(512), b(512);
  std::vector<float> d(1024);
  auto run = [&](const std::vector<float> &A, const std::vector<float> &B) {  // A[32][16], B[16][32] -> D[32][32]
    for (int l = 0; l < 64; l++)
      for (int j = 0; j < 8; j++) {
        a[8 * l + j] = f16_bits(A[(l & 31) * 16 + 8 * (l >> 5) + j]);
        b[8 * l + j] = f16_bits(B[(8 * (l >> 5) + j) * 32 + (l & 31)]);
      }
    CK(hipMemcpy(da, a.data(), 1024, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, b.data(), 1024, hipMemcpyHostToDevice));
    probe_kernel<<<1, 64>>>(da, db, dd);
    CK(hipMemcpy(d.data(), dd, 4096, hipMemcpyDeviceToHost));
    std::vector<float> D(1024);
    for (int l = 0; l < 64; l++)
      for (int r = 0; r < 16; r++) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = d[16 * l + r];
    return D;
  };
  int bad = 0;
  {  // 1. lane maps, exact integers, asymmetric operands
    std::vector<float> A(512), B(512);
    uint32_t x = 7;
    for (auto &v : A) { x = x * 1664525u + 1013904223u; v = (float)((int)(x >> 28) - 8); }
    for (auto &v : B) { x = x * 1664525u + 1013904223u; v = (float)((int)(x >> 28) - 8); }
    auto D = run(A, B);
    for (int i = 0; i < 32; i++)
      for (int j = 0; j < 32; j++) {
        float s = 0;
        for (int k = 0; k < 16; k++) s += A[i * 16 + k] * B[k * 32 + j];
        bad += s != D[i * 32 + j];
      }
    std::printf("probe: lane maps (A[l&31][8(l>>5)+j], B[8(l>>5)+j][l&31], D row (r&3)+8(r>>2)+4(l>>5), col l&31): %s\n", bad ? "WRONG" : "ok");
  }
  {  // 2. subnormal f16 operands: A = selector of k = row % 16, B[k][c] = (k + 1) 2^-20 (subnormal: below 2^-14)
    std::vector<float> A(512, 0.f), B(512);
    for (int i = 0; i < 32; i++) A[i * 16 + (i & 15)] = 1.0f;
    for (int k = 0; k < 16; k++)
      for (int c = 0; c < 32; c++) B[k * 32 + c] = (float)(k + 1) * 9.5367431640625e-07f;
    auto D = run(A, B);
    int flushed = 0;
    for (int i = 0; i < 32; i++) flushed += D[i * 32 + 3] != (float)((i & 15) + 1) * 9.5367431640625e-07f;
    std::printf("probe: subnormal B operand (k+1) 2^-20: D[5][3] = %.6e (exact %.6e): %s\n", D[5 * 32 + 3], 6 * 9.5367431640625e-07,
                flushed ? "FLUSHED" : "kept");
    // subnormal A operand times a large B
    std::fill(A.begin(), A.end(), 0.f);
    for (int i = 0; i < 32; i++) A[i * 16 + (i & 15)] = 3.0f * 9.5367431640625e-07f;
    for (int k = 0; k < 16; k++)
      for (int c = 0; c < 32; c++) B[k * 32 + c] = 1024.0f;
    D = run(A, B);
    std::printf("probe: subnormal A operand 3 x 2^-20 times 1024: D[5][3] = %.6e (exact %.6e): %s\n", D[5 * 32 + 3], 3 * 9.5367431640625e-07 * 1024,
                D[5 * 32 + 3] == 3.0f * 9.5367431640625e-07f * 1024.0f ? "kept" : "FLUSHED");
    bad += flushed;
  }
  {  // 3. what the accumulation rounds: 2^12 + fourteen 2^-12 ... - 2^12, in several orders of k
    for (int pos = 0; pos < 3; pos++) {
      std::vector<float> A(512, 1.0f), B(512);
      for (int k = 0; k < 16; k++) {
        float v = 0.000244140625f;
        if (pos == 0) { if (k == 0) v = 4096.f; if (k == 15) v = -4096.f; }
        if (pos == 1) { if (k == 7) v = 4096.f; if (k == 8) v = -4096.f; }
        if (pos == 2) { if (k == 0) v = 4096.f; if (k == 1) v = -4096.f; }
        for (int c = 0; c < 32; c++) B[k * 32 + c] = v;
      }
      auto D = run(A, B);
      std::printf("probe: accumulation, +-2^12 at k = %s among fourteen 2^-12: D = %.9e (exact %.9e; an f32 chain in k order gives %s)\n",
                  pos == 0 ? "0 / 15" : pos == 1 ? "7 / 8" : "0 / 1", D[0], 14 * 0.000244140625,
                  pos == 0 ? "0" : pos == 1 ? "1.708984375e-03 or so" : "exact");
    }
  }
  CK(hipFree(da));
  CK(hipFree(db));
  CK(hipFree(dd));
  return bad;
}

// ---- the lab's data ---------------------------------------------------------------------------------------------------
struct Lab {
  int eps = 28, frames = 5813;
  size_t samples_per_ep = 7938000;
  uint32_t total_pairs = 0;
  int16_t *d_pcm = nullptr;
  Stream *d_streams = nullptr;
  Tables tab{};
  float *d_out = nullptr;
  cf *d_z = nullptr;
  std::vector<int16_t> pcm;
  hipEvent_t a, b;
};

static void setup(Lab &L) {
  const int pairs_per_ep = (L.frames + 1) / 2;
  L.total_pairs = (uint32_t)(L.eps * pairs_per_ep);
  L.pcm.resize(L.samples_per_ep * L.eps + 8192);
  uint32_t x = 12345;
  for (size_t i = 0; i < L.pcm.size(); i++) {  // a few tones + noise: not silence, not white (as tools/stft32_lab.hip)
    x = x * 1664525u + 1013904223u;
    const double ph = (double)i / 11025.0;
    L.pcm[i] = (int16_t)(6000.0 * std::sin(6.2831853 * 220.0 * ph) + 3000.0 * std::sin(6.2831853 * 1333.0 * ph) + (double)((int)(x >> 20) - 2048));
  }
  std::vector<Stream> st(L.eps);
  for (int e = 0; e < L.eps; e++) st[e] = Stream{L.samples_per_ep * e, (uint32_t)L.frames, (uint32_t)(pairs_per_ep * e)};
  CK(hipMalloc(&L.d_pcm, L.pcm.size() * 2));
  CK(hipMemcpy(L.d_pcm, L.pcm.data(), L.pcm.size() * 2, hipMemcpyHostToDevice));
  CK(hipMalloc(&L.d_streams, st.size() * sizeof(Stream)));
  CK(hipMemcpy(L.d_streams, st.data(), st.size() * sizeof(Stream), hipMemcpyHostToDevice));
  HostTables ht;
  build_tables(&ht);
  auto up = [&](const void *src, size_t bytes) {
    void *p = nullptr;
    CK(hipMalloc(&p, bytes));
    CK(hipMemcpy(p, src, bytes, hipMemcpyHostToDevice));
    return p;
  };
  L.tab.afrag = (const u32x4 *)up(ht.afrag.data(), ht.afrag.size() * 4);
  L.tab.tw0 = (const cf *)up(ht.tw0.data(), ht.tw0.size() * sizeof(cf));
  L.tab.tw1 = (const cf *)up(ht.tw1.data(), ht.tw1.size() * sizeof(cf));
  L.tab.win = (const float *)up(ht.win.data(), ht.win.size() * 4);
  CK(hipMalloc(&L.d_out, (size_t)4096 * 256 * 4));
  CK(hipMalloc(&L.d_z, (size_t)16 * kN * sizeof(cf)));
  CK(hipEventCreate(&L.a));
  CK(hipEventCreate(&L.b));
}

template <typename K>
static float launch(Lab &L, K kernel, int wgs_per_cu, uint32_t first_pair, uint32_t pairs, cf *z, size_t lds_bytes) {
  const uint32_t slots = 256u * (uint32_t)wgs_per_cu;
  const uint32_t grid = std::min(pairs, slots * 4u);  // ~4 workgroups per slot: the product's long launches are similar
  const uint32_t per = (pairs + grid - 1) / grid;
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  CK(hipEventRecord(L.a));
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), lds_bytes, 0, L.d_pcm, L.d_streams, L.eps, L.tab, first_pair, pairs, per, L.d_out, z);
  CK(hipEventRecord(L.b));
  CK(hipEventSynchronize(L.b));
  CK(hipGetLastError());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, L.a, L.b));
  return ms;
}

static int verify(Lab &L) {
  const uint32_t first = 2900, n = 12;  // pairs 2900 .. 2911: crosses the end of stream 0 (2907 pairs: its last pair has no frame B)
  float ms = launch(L, stft_mfma_core_kernel<3, kVerify>, 3, first, n, L.d_z, kLdsBytes);
  (void)ms;
  std::vector<cf> z((size_t)n * kN);
  CK(hipMemcpy(z.data(), L.d_z, z.size() * sizeof(cf), hipMemcpyDeviceToHost));
  const int pairs_per_ep = (L.frames + 1) / 2;
  double worst = 0;
  std::vector<std::complex<double>> x(kN), y(kN);
  std::vector<std::complex<double>> tw(kN);
  for (int k = 0; k < kN; k++) tw[k] = std::polar(1.0, -2.0 * M_PI * k / kN);
  for (uint32_t p = 0; p < n; p++) {
    const uint32_t g = first + p;
    const int e = (int)(g / pairs_per_ep), fa = 2 * (int)(g % pairs_per_ep);
    const bool has_b = fa + 1 < L.frames;
    const int16_t *a = &L.pcm[L.samples_per_ep * e + (size_t)fa * kHop], *b = a + kHop;
    for (int i = 0; i < kN; i++) {
      const double w = 0.54 - 0.46 * std::cos(2.0 * M_PI * i / (kN - 1));
      x[i] = std::complex<double>(a[i] * w, has_b ? b[i] * w : 0.0);
    }
    // f64 transform (radix 2, decimation in time, plain recursion unrolled iteratively)
    for (int i = 0; i < kN; i++) {
      int r = 0;
      for (int bit = 0; bit < 12; bit++) r |= ((i >> bit) & 1) << (11 - bit);
      y[r] = x[i];
    }
    for (int len = 2; len <= kN; len <<= 1)
      for (int s = 0; s < kN; s += len)
        for (int k = 0; k < len / 2; k++) {
          const auto u = y[s + k], v = y[s + k + len / 2] * tw[(size_t)k * (kN / len)];
          y[s + k] = u + v;
          y[s + k + len / 2] = u - v;
        }
    double num = 0, den = 0;
    for (int k = 0; k < kN; k++) {
      const std::complex<double> got((double)z[(size_t)p * kN + k].x * 256.0, (double)z[(size_t)p * kN + k].y * 256.0);
      num += std::norm(got - y[k]);
      den += std::norm(y[k]);
    }
    const double rel = std::sqrt(num / den);
    worst = std::max(worst, rel);
    std::printf("verify: pair %u (stream %d, frames %d%s): relative l2 error %.3e\n", g, e, fa, has_b ? " + next" : ", no frame B", rel);
  }
  std::printf("verify: worst %.3e (2^-24 = 5.96e-08; an f32 radix-16 transform is ~1.3e-07): %s\n", worst, worst < 1e-6 ? "ok" : "WRONG");
  return worst < 1e-6 ? 0 : 1;
}

template <typename K>
static void time_variant(Lab &L, const char *name, K kernel, int wgs, int reps, size_t lds = kLdsBytes) {
  std::vector<float> ms;
  for (int r = 0; r < reps + 2; r++) {
    float v = launch(L, kernel, wgs, 0, L.total_pairs, nullptr, lds);
    if (r >= 2) ms.push_back(v);
  }
  std::sort(ms.begin(), ms.end());
  std::printf("time: %-58s median %.4f ms  best %.4f  (%u pairs, %d workgroups per CU)\n", name, ms[ms.size() / 2], ms[0], L.total_pairs, wgs);
  std::fflush(stdout);
}

int main(int argc, char **argv) {
  bool do_probe = argc < 2, do_verify = argc < 2, do_time = argc < 2;
  for (int i = 1; i < argc; i++) {
    do_probe |= !std::strcmp(argv[i], "probe");
    do_verify |= !std::strcmp(argv[i], "verify");
    do_time |= !std::strcmp(argv[i], "time");
  }
  int rc = 0;
  if (do_probe) rc |= probe();
  if (!do_verify && !do_time) return rc;
  Lab L;
  setup(L);
  if (do_verify) rc |= verify(L);
  if (do_time) {
    for (int round = 0; round < 2; round++) {
      time_variant(L, "core, 3 workgroups per CU", stft_mfma_core_kernel<3, 0>, 3, 7);
      time_variant(L, "core, 3 workgroups per CU, first round staggered", stft_mfma_core_kernel<3, kStagger>, 3, 7);
      time_variant(L, "core, 2 workgroups per CU", stft_mfma_core_kernel<2, 0>, 2, 7);
      time_variant(L, "core, 4 workgroups per CU (register cap 128)", stft_mfma_core_kernel<4, 0>, 4, 7);
      time_variant(L, "  3/CU without the matrix products", stft_mfma_core_kernel<3, kNoMfma>, 3, 7);
      time_variant(L, "  3/CU without the LDS exchanges", stft_mfma_core_kernel<3, kNoLds>, 3, 7);
      time_variant(L, "  3/CU without twiddles and splits (stage outputs)", stft_mfma_core_kernel<3, kNoEpilogue>, 3, 7);
      time_variant(L, "  3/CU without the barriers", stft_mfma_core_kernel<3, kNoBarrier>, 3, 7);
      time_variant(L, "  3/CU without the global loads", stft_mfma_core_kernel<3, kNoLoads>, 3, 7);
    }
  }
  return rc;
}
