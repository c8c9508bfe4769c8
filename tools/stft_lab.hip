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
  const size_t generated = std::min(pcm.size(), L.samples_per_ep * 28 + 8192);  // beyond 28 streams: copies of them
  for (size_t i = 0; i < generated; i++) {  // a few tones + noise: not silence, not white
    x = x * 1664525u + 1013904223u;
    const double ph = (double)i / 11025.0;
    pcm[i] = (int16_t)(6000.0 * std::sin(6.2831853 * 220.0 * ph) + 3000.0 * std::sin(6.2831853 * 1333.0 * ph) + (double)((int)(x >> 20) - 2048));
  }
  for (size_t i = generated; i < pcm.size(); i += L.samples_per_ep * 28)
    std::memcpy(&pcm[i], &pcm[0], std::min(L.samples_per_ep * 28, pcm.size() - i) * sizeof(int16_t));
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

// One variant: a launcher, its name, whether its chroma must equal the product's, its launch shape.
struct Variant {
  const char *name;
  void (*launch)(Lab &, uint32_t grid, size_t lds, uint32_t ppb);
  bool check;
  uint32_t ppb;
  size_t lds_bytes;
  std::vector<float> ms;
  double diff = -1.0;
};

template <int LAB>
static void launch_variant(Lab &L, uint32_t grid, size_t lds, uint32_t ppb) {
  static bool attr = false;
  static size_t attr_lds = 0;
  auto kern = stft::stft_chroma_kernel<1, LAB>;
  if (!attr || attr_lds != lds) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr = true;
    attr_lds = lds;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, L.d_pcm, L.d_streams, L.eps, L.d_tw, L.d_wcos, L.wconst,
                     L.d_bin_slot, L.d_fold_tab, L.d_chroma, L.total_pairs, ppb, stft::ChunkList{nullptr, nullptr});
}

static void time_once(Lab &L, Variant &v, bool record) {
  const size_t lds = v.lds_bytes ? v.lds_bytes : core::kLds2Slots * sizeof(cd);
  const uint32_t grid = (uint32_t)(((L.total_pairs + v.ppb - 1) / v.ppb + 7) / 8 * 8);
  CK(hipEventRecord(L.a));
  v.launch(L, grid, lds, v.ppb);
  CK(hipEventRecord(L.b));
  CK(hipEventSynchronize(L.b));
  CK(hipGetLastError());
  float t;
  CK(hipEventElapsedTime(&t, L.a, L.b));
  if (record) v.ms.push_back(t);
}

static void check_variant(Lab &L, Variant &v) {
  CK(hipMemset(L.d_chroma, 0, (size_t)L.eps * (L.frames + 1) * 12 * 8));
  time_once(L, v, false);
  std::vector<double> out((size_t)L.eps * L.frames * 12);
  CK(hipMemcpy(out.data(), L.d_chroma, out.size() * 8, hipMemcpyDeviceToHost));
  if (L.ref.empty()) L.ref = out;
  if (v.check) {
    v.diff = 0.0;
    for (size_t i = 0; i < out.size(); i++) v.diff = std::max(v.diff, std::fabs(out[i] - L.ref[i]) / std::max(std::fabs(L.ref[i]), 1e-300));
  }
}

int main(int argc, char **argv) {
  Lab L;
  if (argc > 2) L.eps = std::atoi(argv[2]);  // more streams: PCM far beyond the caches (library scale)
  setup(L);
  using namespace needle::stft;
  const int reps = argc > 1 ? std::atoi(argv[1]) : 40;
  const size_t one_per_cu = 100 * 1024;
  std::vector<Variant> vs = {
      {"product", launch_variant<0>, true, 16, 0},
      {"round-1 order (a stage's butterflies, then its stores)", launch_variant<kLabSerial>, true, 16, 0},
      {"extra barrier between publish and partner reads", launch_variant<kLabExtraB>, true, 16, 0},
      {"round-1 order + extra barrier (the round-1 schedule)", launch_variant<kLabSerial | kLabExtraB>, true, 16, 0},
      {"product, one workgroup per CU (100 KB of LDS)", launch_variant<0>, true, 16, one_per_cu},
      {"no barrier 1 (fold reads -> stage-0 stores)", launch_variant<kLabNoB1>, false, 16, 0},
      {"no barrier 2 (stage-0 stores -> stage-1 reads)", launch_variant<kLabNoB2>, false, 16, 0},
      {"no barrier 3 (power stores -> fold reads)", launch_variant<kLabNoB3>, false, 16, 0},
      {"no workgroup barrier at all", launch_variant<kLabNoB1 | kLabNoB2 | kLabNoB3>, false, 16, 0},
      {"stores spaced out under the next tail", launch_variant<kLabSpaced>, true, 16, 0},
      {"round-1 arithmetic in the 16-point transforms", launch_variant<kLabPlainFft>, true, 16, 0},
      {"round-1 arithmetic, stores spaced", launch_variant<kLabPlainFft | kLabSpaced>, true, 16, 0},
      {"product, 8 pairs per workgroup", launch_variant<0>, true, 8, 0},
      {"product, 32 pairs per workgroup", launch_variant<0>, true, 32, 0},
  };
  for (int i = 0; i < 30; i++) time_once(L, vs[0], false);  // warm-up: the first launches run at a lower clock
  for (Variant &v : vs) check_variant(L, v);
  // the variants take turns, so that clock drift and neighbours on the node hit all of them alike
  for (int r = 0; r < reps; r++)
    for (Variant &v : vs) time_once(L, v, true);
  for (Variant &v : vs) {
    std::sort(v.ms.begin(), v.ms.end());
    std::printf("%-58s ppb=%2u  min %.4f  q25 %.4f  med %.4f ms", v.name, v.ppb, v.ms.front(), v.ms[v.ms.size() / 4], v.ms[v.ms.size() / 2]);
    if (v.diff >= 0.0) std::printf("   max rel diff vs product %.1e", v.diff);
    std::printf("\n");
  }
  return 0;
}
