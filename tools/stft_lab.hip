// Timing laboratory for stft_chroma_kernel (not part of the product): compiles the PRODUCT kernel source
// (needle_amd/csrc/stft_kernel.h) with its LAB switches, runs every variant on BASELINE.json configs[1]'s launch shape
// (28 streams x 5 813 frames = 81 382 frame pairs) and prints the time of each.  Variants whose results should equal the
// product's are compared with it (max relative difference of the chroma rows).
//   hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 tools/stft_lab.hip -o tools/stft_lab
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../needle_amd/csrc/stft_kernel.h"

using needle::core::cd;
namespace core = needle::core;
namespace stft = needle::stft;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

struct Lab {
  int eps = 28, frames = 5813;
  size_t samples_per_ep = 7938000;
  uint32_t total_pairs = 0;
  int16_t *d_pcm = nullptr;
  stft::FpStream *d_streams = nullptr;
  cd *d_tw = nullptr;
  double *d_wcos = nullptr, *d_chroma = nullptr;
  core::WindowConst wconst;
  uint16_t *d_bin_slot = nullptr;
  uint32_t *d_fold_tab = nullptr;
  std::vector<double> ref;
  hipEvent_t a, b;
};

static void setup(Lab &L) {
  const int pairs_per_ep = (L.frames + 1) / 2;
  L.total_pairs = (uint32_t)(L.eps * pairs_per_ep);
  std::vector<int16_t> pcm(L.samples_per_ep * L.eps + 8192);
  uint32_t x = 12345;
  for (size_t i = 0; i < pcm.size(); i++) {  // a few tones + noise: not silence, not white
    x = x * 1664525u + 1013904223u;
    const double ph = (double)i / 11025.0;
    pcm[i] = (int16_t)(6000.0 * std::sin(6.2831853 * 220.0 * ph) + 3000.0 * std::sin(6.2831853 * 1333.0 * ph) + (double)((int)(x >> 20) - 2048));
  }
  std::vector<stft::FpStream> st(L.eps);
  for (int e = 0; e < L.eps; e++) {
    stft::FpStream m{};
    m.pcm_off = L.samples_per_ep * e;
    m.frames = (uint32_t)L.frames;
    m.frame_base = (uint32_t)(L.frames * e);
    m.pair_base = (uint32_t)(pairs_per_ep * e);
    st[e] = m;
  }
  std::vector<cd> tw(4096);
  for (int k = 0; k < 4096; k++) {
    long double ang = -2.0L * 3.14159265358979323846264338327950288L * k / 4096.0L;
    tw[k] = cd{(double)cosl(ang), (double)sinl(ang)};
  }
  const long double theta = 2.0L * 3.14159265358979323846264338327950288L / 4095.0L;
  std::vector<double> wcos(512);
  for (int i = 0; i < 512; i++) wcos[i] = (double)cosl(theta * (long double)(i - 256));
  L.wconst.k2 = (double)(2.0L * cosl(256.0L * theta));
  L.wconst.a = core::kPairInputScale * (0.54 / 32767.0);
  L.wconst.b = core::kPairInputScale * (0.46 / 32767.0);
  std::vector<uint8_t> class_of_bin(core::kNumBins);
  for (int i = core::kMinBin; i < core::kMaxBin; i++) {
    double freq = (double)i * 11025 / 4096;
    double octave = std::log(freq / (440.0 / 16.0)) / std::log(2.0);
    double note = 12 * (octave - std::floor(octave));
    class_of_bin[i - core::kMinBin] = (uint8_t)(int)(signed char)note;
  }
  static core::PowerLayout layout;
  if (!core::build_power_layout(class_of_bin.data(), &layout)) { std::fprintf(stderr, "power layout does not fit\n"); std::exit(1); }
  CK(hipMalloc(&L.d_pcm, pcm.size() * 2));
  CK(hipMalloc(&L.d_streams, st.size() * sizeof(stft::FpStream)));
  CK(hipMalloc(&L.d_tw, 4096 * sizeof(cd)));
  CK(hipMalloc(&L.d_wcos, 512 * 8));
  CK(hipMalloc(&L.d_chroma, (size_t)L.eps * (L.frames + 1) * 12 * 8));
  CK(hipMalloc(&L.d_bin_slot, sizeof(layout.bin_slot)));
  CK(hipMalloc(&L.d_fold_tab, sizeof(layout.fold)));
  CK(hipMemcpy(L.d_pcm, pcm.data(), pcm.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(L.d_streams, st.data(), st.size() * sizeof(stft::FpStream), hipMemcpyHostToDevice));
  CK(hipMemcpy(L.d_tw, tw.data(), 4096 * sizeof(cd), hipMemcpyHostToDevice));
  CK(hipMemcpy(L.d_wcos, wcos.data(), 512 * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(L.d_bin_slot, layout.bin_slot, sizeof(layout.bin_slot), hipMemcpyHostToDevice));
  CK(hipMemcpy(L.d_fold_tab, layout.fold, sizeof(layout.fold), hipMemcpyHostToDevice));
  CK(hipEventCreate(&L.a));
  CK(hipEventCreate(&L.b));
}

template <typename K>
static void run(Lab &L, K kern, const char *name, bool check, uint32_t ppb = 16, size_t lds_bytes = 0) {
  const size_t lds = lds_bytes ? lds_bytes : core::kLds2Slots * sizeof(cd);
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const uint32_t grid = (uint32_t)(((L.total_pairs + ppb - 1) / ppb + 7) / 8 * 8);
  std::vector<float> ms;
  CK(hipMemset(L.d_chroma, 0, (size_t)L.eps * (L.frames + 1) * 12 * 8));
  for (int rep = 0; rep < 8; rep++) {
    CK(hipEventRecord(L.a));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, L.d_pcm, L.d_streams, L.eps, L.d_tw, L.d_wcos, L.wconst,
                       L.d_bin_slot, L.d_fold_tab, L.d_chroma, L.total_pairs, ppb);
    CK(hipEventRecord(L.b));
    CK(hipEventSynchronize(L.b));
    float t;
    CK(hipEventElapsedTime(&t, L.a, L.b));
    if (rep >= 2) ms.push_back(t);
  }
  CK(hipGetLastError());
  std::sort(ms.begin(), ms.end());
  double diff = -1.0;
  std::vector<double> out((size_t)L.eps * L.frames * 12);
  CK(hipMemcpy(out.data(), L.d_chroma, out.size() * 8, hipMemcpyDeviceToHost));
  if (L.ref.empty()) {
    L.ref = out;
  } else if (check) {
    diff = 0.0;
    for (size_t i = 0; i < out.size(); i++) diff = std::max(diff, std::fabs(out[i] - L.ref[i]) / std::max(std::fabs(L.ref[i]), 1e-300));
  }
  std::printf("%-56s ppb=%2u lds=%6zu  min %.4f  med %.4f ms", name, ppb, lds, ms.front(), ms[ms.size() / 2]);
  if (diff >= 0.0) std::printf("   max rel diff vs product %.2e", diff);
  std::printf("\n");
  std::fflush(stdout);
}

int main(int argc, char **argv) {
  Lab L;
  setup(L);
  using namespace needle::stft;
  const int rounds = argc > 1 ? std::atoi(argv[1]) : 2;
  // warm-up: the first launches run at a lower clock
  for (int i = 0; i < 4; i++) run(L, stft_chroma_kernel<1, 0>, "warm-up (product)", true);
  for (int round = 0; round < rounds; round++) {
    std::printf("---- round %d\n", round);
    run(L, stft_chroma_kernel<1, 0>, "product", true);
    run(L, stft_chroma_kernel<1, kLabSerial>, "round-1 order (stage butterflies, then its stores)", true);
    run(L, stft_chroma_kernel<1, kLabExtraB>, "extra barrier between publish and partner reads", true);
    run(L, stft_chroma_kernel<1, kLabSerial | kLabExtraB>, "round-1 order + extra barrier (the round-1 schedule)", true);
    run(L, stft_chroma_kernel<1, 0>, "product, one workgroup per CU (100 KB of LDS)", true, 16, 100 * 1024);
    run(L, stft_chroma_kernel<1, kLabNoB1>, "no barrier 1 (fold reads -> stage-0 stores)", false);
    run(L, stft_chroma_kernel<1, kLabNoB2>, "no barrier 2 (stage-0 stores -> stage-1 reads)", false);
    run(L, stft_chroma_kernel<1, kLabNoB3>, "no barrier 3 (power stores -> fold reads)", false);
    run(L, stft_chroma_kernel<1, kLabNoB1 | kLabNoB2 | kLabNoB3>, "no workgroup barrier at all", false);
  }
  return 0;
}
