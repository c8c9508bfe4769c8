// Timing laboratory for stft_chroma32_kernel (not part of the product): compiles the PRODUCT kernel source
// (needle_amd/csrc/stft32_kernel.h) with its LAB switches and occupancy targets, runs every variant on BASELINE.json
// configs[1]'s launch shape (28 streams x 5 813 frames = 81 382 frame pairs), the variants taking turns, and prints the
// time of each.  Build twice to see what the SLP vectoriser does to it:
//   hipcc -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize --offload-arch=gfx950 tools/stft32_lab.hip -o tools/stft32_lab
//   hipcc -O3 -std=c++17 -ffp-contract=off                    --offload-arch=gfx950 tools/stft32_lab.hip -o tools/stft32_lab_slp
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../needle_amd/csrc/stft32_kernel.h"

using needle::core::cf;
namespace core = needle::core;
namespace stft = needle::stft;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

struct Lab {
  int eps = 28, frames = 5813;
  size_t samples_per_ep = 7938000;
  uint32_t total_pairs = 0;
  int16_t *d_pcm = nullptr;
  stft::FpStream *d_streams = nullptr;
  cf *d_tw = nullptr;
  float *d_win = nullptr, *d_energy = nullptr;
  double *d_chroma = nullptr;
  uint16_t *d_bin_slot = nullptr;
  uint32_t *d_fold_tab = nullptr;
  std::vector<double> ref;
  hipEvent_t a, b;
  hipStream_t stream = nullptr;  // LAB_RESERVE_CUS=n: a stream whose CU mask leaves n CUs out (hipctx.hip stft_stream)
};

static void setup(Lab &L) {
  const int pairs_per_ep = (L.frames + 1) / 2;
  L.total_pairs = (uint32_t)(L.eps * pairs_per_ep);
  std::vector<int16_t> pcm(L.samples_per_ep * L.eps + 8192);
  uint32_t x = 12345;
  const size_t generated = std::min(pcm.size(), L.samples_per_ep * 28 + 8192);
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
  std::vector<cf> tw(4096);
  std::vector<float> win(4096);
  const long double theta = 2.0L * 3.14159265358979323846264338327950288L / 4095.0L;
  for (int k = 0; k < 4096; k++) {
    long double ang = -2.0L * 3.14159265358979323846264338327950288L * k / 4096.0L;
    tw[k] = cf{(float)cosl(ang), (float)sinl(ang)};
    win[k] = (float)((long double)core::kPairInputScale * (0.54L - 0.46L * cosl(theta * (long double)k)) / 32767.0L);
  }
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
  CK(hipMalloc(&L.d_tw, 4096 * sizeof(cf)));
  CK(hipMalloc(&L.d_win, 4096 * 4));
  CK(hipMalloc(&L.d_chroma, (size_t)L.eps * (L.frames + 1) * 12 * 8));
  CK(hipMalloc(&L.d_energy, (size_t)L.eps * (L.frames + 1) * 4 * 4));
  CK(hipMalloc(&L.d_bin_slot, sizeof(layout.bin_slot)));
  CK(hipMalloc(&L.d_fold_tab, sizeof(layout.fold)));
  CK(hipMemcpy(L.d_pcm, pcm.data(), pcm.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(L.d_streams, st.data(), st.size() * sizeof(stft::FpStream), hipMemcpyHostToDevice));
  CK(hipMemcpy(L.d_tw, tw.data(), 4096 * sizeof(cf), hipMemcpyHostToDevice));
  CK(hipMemcpy(L.d_win, win.data(), 4096 * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(L.d_bin_slot, layout.bin_slot, sizeof(layout.bin_slot), hipMemcpyHostToDevice));
  CK(hipMemcpy(L.d_fold_tab, layout.fold, sizeof(layout.fold), hipMemcpyHostToDevice));
  CK(hipEventCreate(&L.a));
  CK(hipEventCreate(&L.b));
}

struct Variant {
  const char *name;
  void (*launch)(Lab &, uint32_t grid, uint32_t ppb);
  bool check;
  uint32_t ppb;
  std::vector<float> ms;
  double diff = -1.0;
};

static needle::Stft32Schedule g_schedule;  // set by time_once from the variant's pairs per workgroup and LAB_GUIDED
static size_t g_lds_bytes = (core::kLds2Slots + 240) * sizeof(cf);  // LAB_LDS_BYTES: more, to cap the workgroups per CU
template <int WAVES, int LAB>
static void launch_variant(Lab &L, uint32_t grid, uint32_t ppb) {
  hipLaunchKernelGGL((stft::stft_chroma32_kernel<1, WAVES, LAB>), dim3(grid), dim3(256), g_lds_bytes, L.stream, L.d_pcm,
                     L.d_streams, L.eps, L.d_tw, L.d_win, L.d_bin_slot, L.d_fold_tab, L.d_chroma, L.d_energy, L.total_pairs,
                     g_schedule, (uint32_t *)nullptr, 0u);
}

static void time_once(Lab &L, Variant &v, bool record) {
  static const bool guided = getenv("LAB_GUIDED") && atoi(getenv("LAB_GUIDED")) != 0;
  g_schedule = needle::stft32_schedule(L.total_pairs, v.ppb, 96, guided);
  const uint32_t grid = 8u * g_schedule.blocks_per_xcd;
  CK(hipEventRecord(L.a, L.stream));
  v.launch(L, grid, v.ppb);
  CK(hipEventRecord(L.b, L.stream));
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
  if (argc > 2) L.eps = std::atoi(argv[2]);
  setup(L);
  if (const char *e = getenv("LAB_LDS_BYTES")) {
    g_lds_bytes = (size_t)atol(e);
    std::printf("dynamic LDS per workgroup: %zu bytes -> %d workgroups per CU by LDS\n", g_lds_bytes, (int)(163840 / g_lds_bytes));
  }
  if (const char *e = getenv("LAB_RESERVE_CUS")) {
    int cus = 0, reserve = atoi(e);
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    std::vector<uint32_t> mask((size_t)(cus + 31) / 32, 0);
    for (int c = 0; c < cus - reserve; c++) mask[(size_t)c / 32] |= 1u << (c % 32);
    CK(hipExtStreamCreateWithCUMask(&L.stream, (uint32_t)mask.size(), mask.data()));
    std::printf("launching on a stream confined to %d of %d CUs\n", cus - reserve, cus);
  }
  using namespace needle::stft;
  const int reps = argc > 1 ? std::atoi(argv[1]) : 40;
  std::vector<Variant> vs = {
      {"product (3 waves/SIMD)", launch_variant<3, 0>, true, 16},
      {"4 waves/SIMD (128 VGPRs)", launch_variant<4, 0>, true, 16},
      {"2 waves/SIMD", launch_variant<2, 0>, true, 16},
      {"window re-read per pair, 3 waves/SIMD", launch_variant<3, kLab32WinLoad>, true, 16},
      {"window re-read per pair, 4 waves/SIMD", launch_variant<4, kLab32WinLoad>, true, 16},
      {"stage inputs through the compiler's ds_read2_b64 (round 3's product)", launch_variant<3, kLab32CompilerReads>, true, 16},
      {"samples converted in the load (tbuffer format load)", launch_variant<3, kLab32FormatLoad>, true, 16},
      {"consumer-side twiddles (tan form) + window folded in", launch_variant<3, kLab32ConsumerTw>, true, 16},
      {"power stores / fold reads conflict-free (wrong slots)", launch_variant<3, kLab32NoConflict>, false, 16},
      {"stage-1 twiddles from LDS [j][n0], 3 waves/SIMD", launch_variant<3, kLab32Tw1Lds>, true, 16},
      {"stage-1 twiddles from LDS, 4 waves/SIMD", launch_variant<4, kLab32Tw1Lds>, true, 16},
      {"stage-1 twiddles from LDS + window re-read, 4 waves", launch_variant<4, kLab32Tw1Lds | kLab32WinLoad>, true, 16},
      {"stage-1 twiddles from LDS + window re-read, 3 waves", launch_variant<3, kLab32Tw1Lds | kLab32WinLoad>, true, 16},
      {"no energy partials", launch_variant<3, kLab32NoEnergy>, true, 16},
      {"no barrier 1", launch_variant<3, kLab32NoB1>, false, 16},
      {"no barrier 2", launch_variant<3, kLab32NoB2>, false, 16},
      {"no barrier 3", launch_variant<3, kLab32NoB3>, false, 16},
      {"no workgroup barrier at all", launch_variant<3, kLab32NoB1 | kLab32NoB2 | kLab32NoB3>, false, 16},
      {"no fold", launch_variant<3, kLab32NoFold>, false, 16},
      {"no partner reads / powers / power stores", launch_variant<3, kLab32NoPower>, false, 16},
      {"no fold, no powers, no energy (transform only)", launch_variant<3, kLab32NoFold | kLab32NoPower | kLab32NoEnergy>, false, 16},
      {"product, 8 pairs per workgroup", launch_variant<3, 0>, true, 8},
      {"product, 24 pairs per workgroup", launch_variant<3, 0>, true, 24},
      {"product, 32 pairs per workgroup", launch_variant<3, 0>, true, 32},
  };
  if (const char *only = getenv("LAB_ONLY")) {  // "0,2,18": run these variants only (counter passes)
    std::vector<Variant> keep;
    for (const char *p = only; *p;) {  // indices only; anything else ends the list (a name here once looped forever)
      char *end = nullptr;
      const size_t i = (size_t)std::strtoul(p, &end, 10);
      if (end == p) break;
      if (i < vs.size() && keep.size() < vs.size()) keep.push_back(vs[i]);
      p = end;
      while (*p == ',') p++;
    }
    if (keep.empty()) {
      std::fprintf(stderr, "LAB_ONLY takes variant indices (\"0,2,18\")\n");
      return 2;
    }
    vs = keep;
  }
  for (int i = 0; i < 30; i++) time_once(L, vs[0], false);  // warm-up: the first launches run at a lower clock
  for (Variant &v : vs) check_variant(L, v);
  for (int r = 0; r < reps; r++)
    for (Variant &v : vs) time_once(L, v, true);
  {  // the clock the product schedule really runs at: stamps of every workgroup, taken after the timing loop (chip warm)
    Variant clk{"clock stamps", launch_variant<3, kLab32Clock>, false, 16};
    CK(hipMemset(L.d_energy, 0, (size_t)L.eps * (L.frames + 1) * 4 * 4));
    for (int i = 0; i < 20; i++) time_once(L, vs[0], false);
    time_once(L, clk, true);
    const size_t wgs = (L.total_pairs + 15) / 16;
    std::vector<uint64_t> st(2 * wgs);
    CK(hipMemcpy(st.data(), L.d_energy, st.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> mhz;
    for (size_t w = 0; w < wgs; w++)
      if (st[2 * w + 1]) mhz.push_back(100.0 * (double)st[2 * w] / (double)st[2 * w + 1]);
    std::sort(mhz.begin(), mhz.end());
    if (!mhz.empty())
      std::printf("in-kernel clock (s_memtime / s_memrealtime x 100 MHz over the pair loop, %zu workgroups): min %.0f  median %.0f  max %.0f MHz; "
                  "that launch took %.4f ms\n", mhz.size(), mhz.front(), mhz[mhz.size() / 2], mhz.back(), clk.ms[0]);
  }
  for (Variant &v : vs) {
    std::sort(v.ms.begin(), v.ms.end());
    std::printf("%-52s ppb=%2u  min %.4f  q25 %.4f  med %.4f ms", v.name, v.ppb, v.ms.front(), v.ms[v.ms.size() / 4], v.ms[v.ms.size() / 2]);
    if (v.diff >= 0.0) std::printf("   max rel diff vs product %.1e", v.diff);
    std::printf("\n");
  }
  return 0;
}
