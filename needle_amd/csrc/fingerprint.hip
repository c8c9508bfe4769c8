// GPU fingerprinter: the chromaprint Context replacement behind
// needle/src/audio/analyzer.rs:176-300 (start/feed/finish/get_fingerprint_raw), batched over streams.
//
//   stft_chroma       : s16 PCM -> Hamming window -> two real frames per 4096-pt complex FFT (f64, in place in
//                       LDS) -> |X|^2 over bins 10..1307 -> 12 pitch-class energies per frame   [frames][12] f64
//   features_classify : 5-tap temporal FIR {.25,.75,1,.75,.25} + L2 normalise (zero if norm < 0.01) into LDS,
//                       then 16 Haar-like filters over a 16x12 window, log-ratio quantised to 2 bits, Gray
//                       coded, packed MSB first -> u32 per kept item (items 0, step, 2*step, ...)
//   (fir_norm + classify: the same two steps as separate kernels, for callers that want the features)
//
// HBM traffic that matters is the PCM read (2 B/sample, each sample touched by 3 overlapping frames:
// re-reads are served by L2) and 96 B/frame of chroma; everything else stays on chip.
#include "fingerprint32.h"
#include "fp_core.h"
#include "hipctx.h"
#include "stft32_kernel.h"
#include "stft_kernel.h"

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <memory>
#include <thread>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <type_traits>

namespace needle {

using core::cd;

namespace {


// ---- constant tables, generated on the host in double and uploaded once per device --------------------
struct FpTables {
  cd *tw = nullptr;                 // [4096] e^{-2 pi i k/4096}
  uint16_t *bin_slot = nullptr;     // [kNumBins] slot in the LDS image of the power pair of bin (kMinBin + i)
  double *wcos = nullptr;           // [512] cos(theta (i - 256)), theta = 2 pi / 4095: window recurrence seeds
  core::WindowConst wconst;         // fp_core.h window_step
  uint32_t *fold_tab = nullptr;     // [12 * 16] fold thread -> first slot | positions << 16 (fp_core.h PowerLayout)
  core::ClassifierThresholds *thr = nullptr;
  // f32 first pass (stft32_kernel.h): correctly rounded twiddles and window
  core::cf *tw32 = nullptr;         // [4096]
  float *win32 = nullptr;           // [4096] (0.54 - 0.46 cos(theta n)) / 32767 / 2
};

std::mutex g_tab_mu;
std::map<int, FpTables> g_tables;

const double kThresholds[16][3] = {
    {1.98215, 2.35817, 2.63523},          {-1.03809, -0.651211, -0.282167},  {-0.298702, 0.119262, 0.558497},
    {-0.105439, 0.0153946, 0.135898},     {-0.142891, 0.0258736, 0.200632},  {-0.826319, -0.590612, -0.368214},
    {-0.557409, -0.233035, 0.0534525},    {-0.0646826, 0.00620476, 0.0784847}, {-0.192387, -0.029699, 0.215855},
    {-0.0397818, -0.00568076, 0.0292026}, {-0.53823, -0.369934, -0.190235},  {-0.124877, 0.0296483, 0.139239},
    {-0.101475, 0.0225617, 0.231971},     {-0.0799915, -0.00729616, 0.063262}, {-0.272556, 0.019424, 0.302559},
    {-0.164292, -0.0321188, 0.0846339},
};

Status get_tables(FpTables *out) {
  int dev = 0;
  NEEDLE_HIP_TRY(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(g_tab_mu);
  auto it = g_tables.find(dev);
  if (it != g_tables.end()) {
    *out = it->second;
    return Status::Ok();
  }
  std::vector<cd> tw(4096);
  for (int k = 0; k < 4096; k++) {
    long double a = -2.0L * 3.14159265358979323846264338327950288L * k / 4096.0L;
    tw[k] = cd{(double)cosl(a), (double)sinl(a)};
  }
  // chromaprint PrepareHammingWindow(scale = 1/INT16_MAX), times fp_core.h's 2^-1, as recurrence seeds and constants
  const long double theta = 2.0L * 3.14159265358979323846264338327950288L / 4095.0L;
  std::vector<double> wcos(512);
  for (int i = 0; i < 512; i++) wcos[i] = (double)cosl(theta * (long double)(i - 256));
  core::WindowConst wconst;
  wconst.k2 = (double)(2.0L * cosl(256.0L * theta));
  wconst.a = core::kPairInputScale * (0.54 / 32767.0);
  wconst.b = core::kPairInputScale * (0.46 / 32767.0);
  // chromaprint Chroma::PrepareNotes: bin -> pitch class; then where each bin's power pair lives in the LDS image
  // and what each fold lane reads (fp_core.h build_power_layout)
  std::vector<uint8_t> class_of_bin(core::kNumBins);
  for (int i = core::kMinBin; i < core::kMaxBin; i++) {
    double freq = (double)i * kSampleRate / kFrameSize;
    double octave = std::log(freq / (440.0 / 16.0)) / std::log(2.0);
    double note = kBands * (octave - std::floor(octave));
    class_of_bin[i - core::kMinBin] = (uint8_t)(int)(signed char)note;
  }
  auto layout = std::make_unique<core::PowerLayout>();
  if (!core::build_power_layout(class_of_bin.data(), layout.get()))
    return Status::Make(NeedleError_Unknown, "pitch classes do not fit the fold's power layout");
  core::ClassifierThresholds thr;
  for (int i = 0; i < 16; i++)
    for (int j = 0; j < 3; j++) thr.e[i][j] = std::exp(kThresholds[i][j]);

  FpTables t;
  NEEDLE_HIP_TRY(hipMalloc((void **)&t.bin_slot, sizeof(layout->bin_slot)));
  NEEDLE_HIP_TRY(hipMemcpy(t.bin_slot, layout->bin_slot, sizeof(layout->bin_slot), hipMemcpyHostToDevice));
  NEEDLE_HIP_TRY(hipMalloc((void **)&t.fold_tab, sizeof(layout->fold)));
  NEEDLE_HIP_TRY(hipMemcpy(t.fold_tab, layout->fold, sizeof(layout->fold), hipMemcpyHostToDevice));
  NEEDLE_HIP_TRY(hipMalloc((void **)&t.tw, tw.size() * sizeof(cd)));
  NEEDLE_HIP_TRY(hipMalloc((void **)&t.wcos, wcos.size() * sizeof(double)));
  t.wconst = wconst;
  NEEDLE_HIP_TRY(hipMalloc((void **)&t.thr, sizeof(thr)));
  NEEDLE_HIP_TRY(hipMemcpy(t.tw, tw.data(), tw.size() * sizeof(cd), hipMemcpyHostToDevice));
  NEEDLE_HIP_TRY(hipMemcpy(t.wcos, wcos.data(), wcos.size() * sizeof(double), hipMemcpyHostToDevice));
  NEEDLE_HIP_TRY(hipMemcpy(t.thr, &thr, sizeof(thr), hipMemcpyHostToDevice));
  std::vector<core::cf> tw32(4096);
  std::vector<float> win32(4096);
  for (int k = 0; k < 4096; k++) {
    const long double a = -2.0L * 3.14159265358979323846264338327950288L * k / 4096.0L;
    tw32[k] = core::cf{(float)cosl(a), (float)sinl(a)};
    win32[k] = (float)((long double)core::kPairInputScale * (0.54L - 0.46L * cosl(theta * (long double)k)) / 32767.0L);
  }
  NEEDLE_HIP_TRY(hipMalloc((void **)&t.tw32, tw32.size() * sizeof(core::cf)));
  NEEDLE_HIP_TRY(hipMalloc((void **)&t.win32, win32.size() * sizeof(float)));
  NEEDLE_HIP_TRY(hipMemcpy(t.tw32, tw32.data(), tw32.size() * sizeof(core::cf), hipMemcpyHostToDevice));
  NEEDLE_HIP_TRY(hipMemcpy(t.win32, win32.data(), win32.size() * sizeof(float), hipMemcpyHostToDevice));
  g_tables[dev] = t;
  *out = t;
  return Status::Ok();
}

// NEEDLE_HIP_STFT_SHARE (hipctx.hip stft_stream): 2 = default, the next job's first pass beside this job's tail; 1 = behind
// this job's recomputation (round-4 measurement); 0 = one stream
int share_mode() {
  static const int mode = [] {
    const char *e = getenv("NEEDLE_HIP_STFT_SHARE");
    return e ? atoi(e) : 2;
  }();
  return mode;
}

using stft::FpStream;
using stft::find_stream;
using stft::stft_chroma_kernel;
using stft::wave_lds_fence;
using stft::kPairsPerBlock;

// One feature row: 5-tap temporal FIR over chroma rows in[0..4] + L2 normalise (zero if the norm is < 0.01).
__device__ __forceinline__ void feature_row(const double *__restrict__ in, double *out) {
  const double coef[5] = {0.25, 0.75, 1.0, 0.75, 0.25};
  double v[kBands];
  double squares = 0.0;
#pragma unroll
  for (int c = 0; c < kBands; c++) {
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < 5; j++) acc += in[j * kBands + c] * coef[j];
    v[c] = acc;
    squares += acc * acc;
  }
  const double norm = squares > 0.0 ? sqrt(squares) : 0.0;
  if (norm < 0.01) {
#pragma unroll
    for (int c = 0; c < kBands; c++) out[c] = 0.0;
  } else {
#pragma unroll
    for (int c = 0; c < kBands; c++) out[c] = v[c] / norm;
  }
}

// ---- kernel 2: temporal FIR + L2 normalise, one thread per output row -------------------------------------
// (kernels 2 and 3 run separately only when a caller asks for the intermediate features; otherwise kernel 2+3)
__global__ __launch_bounds__(256) void fir_norm_kernel(const double *__restrict__ chroma,
                                                       const FpStream *__restrict__ streams, int num_streams,
                                                       double *__restrict__ feat, uint32_t total_rows) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= total_rows) return;
  const int si = find_stream<&FpStream::fir_base>(streams, num_streams, g);
  const FpStream st = streams[si];
  const uint32_t r = g - st.fir_base;
  feature_row(chroma + ((uint64_t)st.frame_base + r) * kBands, feat + (uint64_t)g * kBands);
}

// ---- kernel 2+3: features of a tile in LDS, then its items -------------------------------------------------------
// A wave owns a tile of up to 64 consecutive kept items of one stream: it computes the (items - 1) step + 16
// feature rows the tile's windows cover into its own LDS region (each row once; neighbouring tiles repeat only the
// 15-row halo), then every lane classifies its window out of LDS.  The features never go to HBM and one dependent
// launch disappears.  Row pitch 13: with step 2 a lane's window starts 26 doubles after its neighbour's, which
// spreads the lanes over all banks (pitch 12 would put every fourth lane on the same ones).
constexpr int kTileRowsMax = 63 * 2 + 16;  // 64 items at the default step 2
constexpr int kFeatPitch = 13;
__global__ __launch_bounds__(256) void features_classify_kernel(const double *__restrict__ chroma,
                                                                const FpStream *__restrict__ streams, int num_streams,
                                                                const core::ClassifierThresholds *__restrict__ thr,
                                                                uint32_t step, uint32_t items_per_tile,
                                                                uint32_t *__restrict__ items, uint32_t total_tiles) {
  __shared__ double tiles[4][kTileRowsMax * kFeatPitch];
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const uint32_t g = blockIdx.x * 4 + wave;
  if (g >= total_tiles) return;  // wave-uniform; the waves of a workgroup never wait for each other
  const int si = find_stream<&FpStream::tile_base>(streams, num_streams, g);
  const FpStream st = streams[si];
  const uint32_t k0 = (g - st.tile_base) * items_per_tile;
  const uint32_t count = min(items_per_tile, st.kept - k0);
  const uint32_t x0 = k0 * step;  // raw item index = first feature row of the tile
  const uint32_t rows = (count - 1) * step + 16;
  double *mine = tiles[wave];
  const double *in = chroma + ((uint64_t)st.frame_base + x0) * kBands;
  for (uint32_t r = lane; r < rows; r += 64) feature_row(in + (uint64_t)r * kBands, mine + r * kFeatPitch);
  wave_lds_fence();
  if (lane < count)
    items[st.item_off + k0 + lane] = core::classify_window<kFeatPitch>(mine + lane * step * kFeatPitch, thr);
}

// ---- certified first pass -----------------------------------------------------------------------------------------
// The chroma of stft_chroma32_kernel carries the f32 transform's error.  Measured with the kernel's own arithmetic
// stepped on the CPU (tools/f32_gate.py, profiles/r03_f32_gate.log): with v = (1 + a) / (1 + b) the input of a
// classifier,
//     |log v32 - log v64|  <=  1.8 * S,     S = max over the item's 16 feature rows of  u sqrt(E_row / n_row),
// u = 2^-24, n_row the row's L2 norm (what the features are divided by) and E_row its frames' total energy
// sum |X_k|^2 over ALL bins through the same 5-tap FIR -- on 28 x 24 min of synthetic episodes and on a zoo of signals
// that spans S from 1e-7 (tonal, in band) to 3e-4 (a strong tone outside chromaprint's band over a weak one inside).
// An item is ACCEPTED only if all 16 x 3 comparisons "v < exp(t)" clear their threshold by more than r = K S in
// log v (K = 64: 35 x the worst ratio observed) and none of its rows is within the same relative distance of the
// 0.01 norm cut; every other item (0.1 % of them on audio) is listed, the chunks of frame pairs its 20 frames span
// are listed once, stft_chroma_kernel<LISTED> overwrites those chroma rows in f64 and fixup_items_kernel recomputes
// the item from them with the arithmetic of features_classify_kernel.  So every emitted u32 is either certified to
// equal the f64 pipeline's or IS the f64 pipeline's.
struct CertItem {
  uint32_t row;   // chroma row of the item's first frame (global in the batch)
  uint32_t pad;
  uint64_t out;   // where its u32 goes in d_items
};
struct CertWork {      // zeroed before every batch (header + bitmap)
  uint32_t item_count, chunk_count, pad[2];
};
struct CertStats {     // cumulative, read by needle_hip_fingerprint_cert_stats
  unsigned long long items_recomputed, chunks_recomputed;
};
constexpr float kCertU = 5.9604644775390625e-08f;  // 2^-24
constexpr float kEnergyScale = 16384.0f;            // N * 4: the kernel's samples carry a factor 1/2 (fp_core.h)

// feature_row + the row's error scale sigma = u sqrt(E_row / n_row); +inf if the row sits within k sigma (relative) of
// the 0.01 cut, 0 if it is safely under it (features exactly zero in both pipelines) or silent.
// (T = float: a tile staged in LDS by features_classify_cert_kernel -- the first pass's chroma IS f32, kept as doubles in
// the buffer the f64 recomputation overwrites; the conversion back is exact and the row is the same bit for bit.)
template <typename T>
__device__ __forceinline__ float feature_row_cert(const T *__restrict__ in, const float *__restrict__ en, float k,
                                                  double *out) {
  const double coef[5] = {0.25, 0.75, 1.0, 0.75, 0.25};
  double v[kBands];
  double squares = 0.0;
#pragma unroll
  for (int c = 0; c < kBands; c++) {
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < 5; j++) acc += (double)in[j * kBands + c] * coef[j];
    v[c] = acc;
    squares += acc * acc;
  }
  const double norm = squares > 0.0 ? sqrt(squares) : 0.0;
  float e_row = 0.0f;
#pragma unroll
  for (int j = 0; j < 5; j++)
    e_row += (float)coef[j] * ((en[j * stft::kEnergyParts] + en[j * stft::kEnergyParts + 1]) +
                               (en[j * stft::kEnergyParts + 2] + en[j * stft::kEnergyParts + 3]));
  float sigma = 0.0f;
  if (e_row > 0.0f) {
    const float n32 = fmaxf((float)norm, 1e-30f);
    sigma = kCertU * sqrtf(kEnergyScale * e_row / n32);
    if (fabsf(n32 - 0.01f) <= k * sigma * n32 + 1e-9f) sigma = __builtin_inff();
  }
  if (norm < 0.01) {
#pragma unroll
    for (int c = 0; c < kBands; c++) out[c] = 0.0;
    return sigma == __builtin_inff() ? sigma : 0.0f;
  }
  // one reciprocal instead of twelve divisions: a feature may differ from the f64 pipeline's by an ulp, 10^9 times less
  // than the radius an ACCEPTED item clears; every other item is recomputed by fixup_items_kernel with the divisions
  const double inv = 1.0 / norm;
#pragma unroll
  for (int c = 0; c < kBands; c++) out[c] = v[c] * inv;
  return sigma;
}

// classify_window + "is any of the 48 comparisons within rr (relative) of its threshold"
template <int PITCH, int ODD = 0>
__device__ __forceinline__ uint32_t classify_window_cert(const double *w, const core::ClassifierThresholds *thr, double rr,
                                                         bool *uncertain) {
  double a[16], b[16];
#pragma unroll
  for (int i = 0; i < 16; i++) a[i] = b[i] = 0.0;
  core::WindowStep<0, 0, PITCH, ODD>::run(w, a, b);
  uint32_t bits = 0;
  bool unc = false;
#pragma unroll
  for (int i = 0; i < 16; i++) {
    // ratio < e^t  <=>  1 + a < e^t (1 + b)  (b >= 0): no division -- see feature_row_cert for why an ulp is harmless here
    const double num = 1.0 + a[i], den = 1.0 + b[i];
    const double d0 = thr->e[i][0] * den, d1 = thr->e[i][1] * den, d2 = thr->e[i][2] * den;
    const unsigned q = num < d1 ? (num < d0 ? 0u : 1u) : (num < d2 ? 2u : 3u);
    // |log ratio - t| <= r  <=  |ratio - e^t| <= e^t (r + r^2)  <=>  |num - e^t den| <= e^t den (r + r^2)   (rr = r + r^2, r < 1)
    unc = unc || fabs(num - d0) <= d0 * rr || fabs(num - d1) <= d1 * rr || fabs(num - d2) <= d2 * rr;
    bits = (bits << 2) | (q ^ (q >> 1));
  }
  *uncertain = unc;
  return bits;
}

// kCertWaves waves per workgroup, each with a tile of its own and no barrier between them: two, so that five workgroups
// (30 KB of LDS each) fit a CU -- the kernel is latency-bound and now needs 150 VGPRs, not 376 (fp_core.h WindowStep).
constexpr int kCertWaves = 2;
constexpr int kHalfRows = kTileRowsMax / 2;  // SPLIT: even rows of a tile first, its odd rows from here on
// SPLIT (chosen by the launcher when step == 2): the tile's rows lie de-interleaved in LDS (fp_core.h WindowStep ODD) --
// same values, same additions in the same order; only where a row is kept differs.
template <bool SPLIT>
__global__ __launch_bounds__(64 * kCertWaves) void features_classify_cert_kernel(
    const double *__restrict__ chroma, const float *__restrict__ energy, const FpStream *__restrict__ streams, int num_streams,
    const core::ClassifierThresholds *__restrict__ thr, uint32_t step, uint32_t items_per_tile, uint32_t *__restrict__ items,
    uint32_t total_tiles, float cert_k, uint32_t chunk_pairs, CertWork *__restrict__ work, uint32_t *__restrict__ chunk_bitmap,
    uint32_t *__restrict__ chunk_list, CertItem *__restrict__ item_list) {
  __shared__ double tiles[kCertWaves][kTileRowsMax * kFeatPitch];
  __shared__ float sigmas[kCertWaves][kTileRowsMax];
  // The tile's input -- rows + 4 chroma rows and their energy partials, one contiguous span each -- is staged first, with
  // coalesced 16-byte loads: every lane building its feature rows straight from global memory is 60 loads of 8 bytes at a
  // lane stride of 96 bytes, 48 cache lines per instruction, and the wave spent half its life waiting for them (SQ_WAIT_ANY
  // 49 % of SQ_WAVE_CYCLES, profiles/r04_final_summary.md).  The chroma is staged as the f32 it is (feature_row_cert).
  __shared__ __attribute__((aligned(16))) float stage[kCertWaves][(kTileRowsMax + 4) * kBands];
  __shared__ __attribute__((aligned(16))) float stage_en[kCertWaves][(kTileRowsMax + 4) * stft::kEnergyParts];
  static_assert(((kTileRowsMax + 4) * kBands) % 2 == 0 && stft::kEnergyParts == 4, "the staging loops move double2 / float4");
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const uint32_t g = blockIdx.x * kCertWaves + wave;
  if (g >= total_tiles) return;  // wave-uniform; the waves of a workgroup never wait for each other
  __builtin_amdgcn_s_setprio(3);  // a tail kernel beside the next job's first pass: its waves issue first (fingerprint.hip, shared-CU overlap)
  const int si = find_stream<&FpStream::tile_base>(streams, num_streams, g);
  const FpStream st = streams[si];
  const uint32_t k0 = (g - st.tile_base) * items_per_tile;
  const uint32_t count = min(items_per_tile, st.kept - k0);
  const uint32_t x0 = k0 * step;
  const uint32_t rows = (count - 1) * step + 16;
  double *mine = tiles[wave];
  float *sig = sigmas[wave];
  const double *in = chroma + ((uint64_t)st.frame_base + x0) * kBands;
  const float *en = energy + ((uint64_t)st.frame_base + x0) * stft::kEnergyParts;
#ifdef NEEDLE_CERT_STAMPS
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  uint64_t t1 = 0, t2 = 0, t3 = 0;
#endif
  {
    float *sc = stage[wave], *se = stage_en[wave];
    const double2 *in2 = reinterpret_cast<const double2 *>(in);  // 96-byte rows: 16-byte aligned
    const float4 *en4 = reinterpret_cast<const float4 *>(en);
    const uint32_t n2 = (rows + 4) * kBands / 2, n4 = rows + 4;
    // every load of the tile in flight before the first is used (a rolled loop is one memory round trip per 1 KB:
    // 13 000 of the wave's 30 000 cycles when this was measured)
    constexpr int kLoads2 = ((kTileRowsMax + 4) * kBands / 2 + 63) / 64, kLoads4 = (kTileRowsMax + 4 + 63) / 64;
    double2 v2[kLoads2];
    float4 v4[kLoads4];
#pragma unroll
    for (int u = 0; u < kLoads2; u++) {
      const uint32_t i = lane + 64u * u;
      v2[u] = i < n2 ? in2[i] : double2{0.0, 0.0};
    }
#pragma unroll
    for (int u = 0; u < kLoads4; u++) {
      const uint32_t i = lane + 64u * u;
      v4[u] = i < n4 ? en4[i] : float4{0.0f, 0.0f, 0.0f, 0.0f};
    }
#pragma unroll
    for (int u = 0; u < kLoads2; u++) {
      const uint32_t i = lane + 64u * u;
      if (i < n2) *reinterpret_cast<float2 *>(sc + 2 * i) = float2{(float)v2[u].x, (float)v2[u].y};
    }
#pragma unroll
    for (int u = 0; u < kLoads4; u++) {
      const uint32_t i = lane + 64u * u;
      if (i < n4) *reinterpret_cast<float4 *>(se + 4 * i) = v4[u];
    }
    wave_lds_fence();
#ifdef NEEDLE_CERT_STAMPS
    t1 = __builtin_amdgcn_s_memtime();
#endif
    for (uint32_t r = lane; r < rows; r += 64)
      sig[r] = feature_row_cert(sc + r * kBands, se + r * stft::kEnergyParts, cert_k,
                                mine + (SPLIT ? (r >> 1) + (r & 1u) * kHalfRows : r) * kFeatPitch);
  }
  wave_lds_fence();
#ifdef NEEDLE_CERT_STAMPS
  t2 = __builtin_amdgcn_s_memtime();
#endif
  if (lane < count) {
    float s_max = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; r++) s_max = fmaxf(s_max, sig[lane * step + r]);
    bool unc = false;
    const double r = (double)cert_k * (double)s_max;   // +inf when a row is at the norm cut
    const uint32_t bits = SPLIT ? classify_window_cert<kFeatPitch, kHalfRows * kFeatPitch>(mine + lane * kFeatPitch, thr, r + r * r, &unc)
                                : classify_window_cert<kFeatPitch>(mine + lane * step * kFeatPitch, thr, r + r * r, &unc);
    unc = unc || !(r < 0.25);                           // out of the calibrated regime: recompute
    const uint64_t out = st.item_off + k0 + lane;
    items[out] = bits;
#ifdef NEEDLE_CERT_STAMPS
    t3 = __builtin_amdgcn_s_memtime();
    if (lane == 0 && (g % 251) == 0)
      printf("cert tile %u rows %u: stage %llu, rows %llu, classify %llu cycles\n", g, rows, (unsigned long long)(t1 - t0),
             (unsigned long long)(t2 - t1), (unsigned long long)(t3 - t2));
#endif
    if (unc) {
      const uint32_t x = x0 + lane * step;              // raw item = first frame of the 20 it covers
      item_list[atomicAdd(&work->item_count, 1u)] = CertItem{st.frame_base + x, 0u, out};
      const uint32_t p0 = st.pair_base + x / 2, p1 = st.pair_base + min(x + 19u, st.frames - 1u) / 2;
      for (uint32_t c = p0 / chunk_pairs; c <= p1 / chunk_pairs; c++) {
        const uint32_t bit = 1u << (c & 31u);
        if (!(atomicOr(&chunk_bitmap[c >> 5], bit) & bit)) chunk_list[atomicAdd(&work->chunk_count, 1u)] = c;
      }
    }
  }
}

// one wave per listed item: its 16 feature rows from the (now f64) chroma rows, then the 16 classifiers -- the
// arithmetic of features_classify_kernel, function for function
__global__ __launch_bounds__(256) void fixup_items_kernel(const double *__restrict__ chroma,
                                                          const core::ClassifierThresholds *__restrict__ thr,
                                                          const CertWork *__restrict__ work, const CertItem *__restrict__ item_list,
                                                          uint32_t *__restrict__ items, CertStats *__restrict__ stats,
                                                          uint32_t *__restrict__ zero_word) {
  __shared__ double tiles[4][16 * kFeatPitch];
  __builtin_amdgcn_s_setprio(3);  // a tail kernel beside the next job's first pass: its waves issue first (fingerprint.hip, shared-CU overlap)
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const uint32_t n = work->item_count;
  double *mine = tiles[wave];
  for (uint32_t i = blockIdx.x * 4 + wave; i < n; i += gridDim.x * 4) {
    const CertItem it = item_list[i];
    if (lane < 16) feature_row(chroma + ((uint64_t)it.row + lane) * kBands, mine + lane * kFeatPitch);
    wave_lds_fence();
    if (lane == 0) items[it.out] = core::classify_window<kFeatPitch>(mine, thr);
    wave_lds_fence();  // the next item of this wave overwrites the tile
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (zero_word) *zero_word = 0u;  // the run counter of the scan that follows (a memset dispatch less in front of it)
    atomicAdd(&stats->items_recomputed, (unsigned long long)n);
    atomicAdd(&stats->chunks_recomputed, (unsigned long long)work->chunk_count);
  }
}

// ---- audit of the certified first pass -----------------------------------------------------------------------------
// Not part of a job: needle_hip_fingerprint_audit_device / needle_hip_library_audit run BOTH transforms over the same
// resident PCM -- stft_chroma32_kernel into one chroma buffer, stft_chroma_kernel (f64) into another -- and this kernel
// then looks at every kept item with the certification kernel's own functions: the acceptance decision from the f32
// chroma (feature_row_cert, classify_window_cert, the same K), the f64 pipeline's item from the f64 chroma
// (feature_row, classify_window), and for every ACCEPTED item the quantity the radius is a bound on,
//     max over the 16 classifiers of |log v32 - log v64| / S,      v = (1 + a) / (1 + b),
// reduced to a maximum over the batch.  Counts: accepted items whose f32 bits differ from the f64 item (each one a
// hole in the guarantee: must be 0), and items of the product's own output `items` that differ from the f64 item.
struct AuditCounts {
  unsigned long long items, accepted, accepted_wrong, final_wrong;
  unsigned long long max_ratio_bits, max_sigma_bits;  // doubles >= 0 compare like their bit patterns
};
__global__ __launch_bounds__(64) void audit_items_kernel(const double *__restrict__ chroma32, const float *__restrict__ energy,
                                                         const double *__restrict__ chroma64, const FpStream *__restrict__ streams,
                                                         int num_streams, const core::ClassifierThresholds *__restrict__ thr,
                                                         uint32_t step, uint32_t items_per_tile, const uint32_t *__restrict__ items,
                                                         uint32_t total_tiles, float cert_k, AuditCounts *__restrict__ out) {
  __shared__ double tile32[kTileRowsMax * kFeatPitch], tile64[kTileRowsMax * kFeatPitch];
  __shared__ float sig[kTileRowsMax];
  const uint32_t lane = threadIdx.x, g = blockIdx.x;
  if (g >= total_tiles) return;
  const int si = find_stream<&FpStream::tile_base>(streams, num_streams, g);
  const FpStream st = streams[si];
  const uint32_t k0 = (g - st.tile_base) * items_per_tile;
  const uint32_t count = min(items_per_tile, st.kept - k0);
  const uint32_t x0 = k0 * step;
  const uint32_t rows = (count - 1) * step + 16;
  const uint64_t row0 = (uint64_t)st.frame_base + x0;
  for (uint32_t r = lane; r < rows; r += 64) {
    sig[r] = feature_row_cert(chroma32 + (row0 + r) * kBands, energy + (row0 + r) * stft::kEnergyParts, cert_k, tile32 + r * kFeatPitch);
    feature_row(chroma64 + (row0 + r) * kBands, tile64 + r * kFeatPitch);
  }
  wave_lds_fence();
  if (lane >= count) return;
  float s_max = 0.0f;
#pragma unroll
  for (int r = 0; r < 16; r++) s_max = fmaxf(s_max, sig[lane * step + r]);
  bool unc = false;
  const double r = (double)cert_k * (double)s_max;
  const uint32_t bits32 = classify_window_cert<kFeatPitch>(tile32 + lane * step * kFeatPitch, thr, r + r * r, &unc);
  unc = unc || !(r < 0.25);
  const uint32_t bits64 = core::classify_window<kFeatPitch>(tile64 + lane * step * kFeatPitch, thr);
  atomicAdd(&out->items, 1ull);
  if (items[st.item_off + k0 + lane] != bits64) atomicAdd(&out->final_wrong, 1ull);
  if (unc) return;
  atomicAdd(&out->accepted, 1ull);
  if (bits32 != bits64) atomicAdd(&out->accepted_wrong, 1ull);
  double a32[16], b32[16], a64[16], b64[16];
#pragma unroll
  for (int i = 0; i < 16; i++) a32[i] = b32[i] = a64[i] = b64[i] = 0.0;
  core::WindowStep<0, 0, kFeatPitch>::run(tile32 + lane * step * kFeatPitch, a32, b32);
  core::WindowStep<0, 0, kFeatPitch>::run(tile64 + lane * step * kFeatPitch, a64, b64);
  double err = 0.0;
#pragma unroll
  for (int i = 0; i < 16; i++)
    err = fmax(err, fabs(log((1.0 + a32[i]) / (1.0 + b32[i])) - log((1.0 + a64[i]) / (1.0 + b64[i]))));
  // S = 0: silence or rows under the norm cut in both pipelines -- every feature is exactly zero, err must be too
  const double ratio = s_max > 0.0f ? err / (double)s_max : (err > 0.0 ? __builtin_inf() : 0.0);
  atomicMax(&out->max_ratio_bits, (unsigned long long)__double_as_longlong(ratio));
  atomicMax(&out->max_sigma_bits, (unsigned long long)__double_as_longlong((double)s_max));
}

// ---- kernel 3: 16 classifiers over a 16x12 window, one thread per kept item ----------------------------------
__global__ __launch_bounds__(256) void classify_kernel(const double *__restrict__ feat,
                                                       const FpStream *__restrict__ streams, int num_streams,
                                                       const core::ClassifierThresholds *__restrict__ thr,
                                                       uint32_t step, uint32_t *__restrict__ items,
                                                       uint32_t total_kept) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= total_kept) return;
  const int si = find_stream<&FpStream::kept_base>(streams, num_streams, g);
  const FpStream st = streams[si];
  const uint32_t k = g - st.kept_base;
  const uint32_t x = k * step;  // raw item index = first row of the window
  const double *w = feat + ((uint64_t)st.fir_base + x) * kBands;
  const uint32_t bits = core::classify_window(w, thr);
  items[st.item_off + k] = bits;
}

// workspace reused across calls (per device)
struct FpWorkspace {
  DeviceBuffer<double> chroma, feat;
  // Stream tables, one slot per chunk of a call: a job that is run again finds every chunk's table resident and
  // uploads nothing (with a single slot the chunks of a large batch evict each other, and every upload then waits
  // for the staging buffer behind the kernels already queued).
  struct Descriptors {
    DeviceBuffer<FpStream> streams;
    PinnedStage stage;
    DescriptorUpload<FpStream> upload;
  };
  static constexpr size_t kDescriptorSlots = 64;
  std::vector<std::unique_ptr<Descriptors>> descriptors;
  Descriptors &slot(size_t chunk) {
    const size_t k = chunk % kDescriptorSlots;
    if (descriptors.size() <= k) descriptors.resize(k + 1);
    if (!descriptors[k]) descriptors[k] = std::make_unique<Descriptors>();
    return *descriptors[k];
  }
  bool lds_attr_set = false;  // the STFT kernel's 68 KiB of dynamic LDS needs an explicit opt-in
  // certified first pass: frame energies, control block (CertWork + chunk bitmap), the two lists, cumulative counts
  DeviceBuffer<float> energy;
  DeviceBuffer<uint32_t> cert_ctl, chunk_list;
  DeviceBuffer<CertItem> item_list;
  // the same set twice more for calls that are pipelined two deep (gpu_fingerprint_device's `pipe`): the STFT of call
  // k + 1 writes its chroma while the certification / recomputation / fix-up of call k still read theirs
  struct Pipe {
    DeviceBuffer<double> chroma;
    DeviceBuffer<float> energy;
    DeviceBuffer<uint32_t> cert_ctl, chunk_list;
    DeviceBuffer<CertItem> item_list;
    hipEvent_t stft_begin = nullptr, stft_done = nullptr;  // bound to the first pass's own dispatch (no marker packets)
    hipEvent_t recomputed = nullptr;  // recorded on the library stream behind the f64 recomputation of the listed chunks
    hipEvent_t consumed = nullptr;    // recorded on the library stream behind the last reader of this set
    hipEvent_t descriptors = nullptr; // recorded on the library stream behind a descriptor upload the STFT must see
    bool consumed_valid = false, stft_recorded = false;
  } pipes[2];
  CertStats *stats = nullptr;                     // device
  uint64_t items_total = 0, chunks_total = 0;     // host: what the device counts are fractions of
};
std::mutex g_ws_mu;
std::map<int, FpWorkspace *> g_ws;

FpWorkspace *workspace() {
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lock(g_ws_mu);
  auto it = g_ws.find(dev);
  if (it != g_ws.end()) return it->second;
  FpWorkspace *w = new FpWorkspace();
  g_ws[dev] = w;
  return w;
}

}  // namespace

Status gpu_fingerprint_device(const int16_t *d_pcm, const std::vector<StreamSpan> &spans, int channels,
                              uint32_t step, uint32_t *d_items, bool sync, double *d_chroma_dbg,
                              double *d_feat_dbg, size_t descriptor_slot, int pipe, uint32_t *zero_word, bool *zeroed_out) {
  std::lock_guard<std::recursive_mutex> gpu_lock(gpu_mutex());
  bool zeroed = false;  // *zero_word was cleared by the LAST kernel enqueued here (only then is it still zero for the caller)
  if (zeroed_out) *zeroed_out = false;
  if (channels != 1 && channels != 2)
    return Status::Make(NeedleError_InvalidArgument, "fingerprint: channels must be 1 or 2");
  if (step == 0) return Status::Make(NeedleError_InvalidArgument, "fingerprint: step must be >= 1");
  Status s = ensure_device();
  if (!s.ok()) return s;
  FpTables tab;
  s = get_tables(&tab);
  if (!s.ok()) return s;
  hipStream_t stream = library_stream();
  FpWorkspace *ws = workspace();

  // Streams are processed in chunks so the f64 chroma/feature workspaces stay bounded (96 B/frame each).
  uint64_t kMaxFramesPerChunk = 8u << 20;  // workspace bound: 8 M frames = 0.8 GB of chroma + as much of features
  if (const char *e = getenv("NEEDLE_HIP_MAX_FRAMES_PER_CHUNK")) kMaxFramesPerChunk = (uint64_t)std::max(1, atoi(e));  // tests
  size_t begin = 0, chunk = descriptor_slot;
  while (begin < spans.size()) {
    std::vector<FpStream> meta;
    uint64_t frames = 0, rows = 0, kept = 0, pairs = 0, tiles = 0;
    // items per tile of the fused feature + classify kernel: as many (up to 64) as its LDS rows cover
    uint32_t items_per_tile = (uint32_t)std::min<uint64_t>(64, (uint64_t)(kTileRowsMax - 16) / step + 1);
    if (const char *e = getenv("NEEDLE_HIP_ITEMS_PER_TILE")) items_per_tile = std::min(items_per_tile, (uint32_t)std::max(1, atoi(e)));  // tuning
    size_t end = begin;
    while (end < spans.size()) {
      const size_t samples = spans[end].num_values / (size_t)channels;
      const uint64_t f = num_frames(samples);
      if (!meta.empty() && frames + f > kMaxFramesPerChunk) break;
      if (f > 0xFFFFFFF0ull) return Status::Make(NeedleError_InvalidArgument, "fingerprint: stream too long");
      FpStream m;
      m.pcm_off = spans[end].pcm_off;
      m.item_off = spans[end].item_off;
      m.frames = (uint32_t)f;
      m.frame_base = (uint32_t)frames;
      m.fir_rows = f >= (uint64_t)kFirTaps ? (uint32_t)(f - (kFirTaps - 1)) : 0;
      m.fir_base = (uint32_t)rows;
      m.kept = (uint32_t)num_kept(samples, step);
      m.kept_base = (uint32_t)kept;
      m.pair_base = (uint32_t)pairs;
      m.tile_base = (uint32_t)tiles;
      tiles += (m.kept + items_per_tile - 1) / items_per_tile;
      pairs += (m.frames + 1) / 2;
      frames += m.frames;
      rows += m.fir_rows;
      kept += m.kept;
      meta.push_back(m);
      end++;
    }
    if (frames > 0) {
      // Pipelined call (pipe 0 / 1) small enough to be ONE chunk, with a CU-masked stream available: its f32 STFT goes
      // to that stream and its own workspace; otherwise everything stays on the library stream.
      const char *mode_env0 = getenv("NEEDLE_HIP_STFT");
      hipStream_t stft = (pipe == 0 || pipe == 1) && begin == 0 && end == spans.size() && d_chroma_dbg == nullptr &&
                                 d_feat_dbg == nullptr && tiles > 0 && !(mode_env0 && std::strcmp(mode_env0, "f64") == 0) &&
                                 getenv("NEEDLE_HIP_SEPARATE_CLASSIFY") == nullptr && frames <= (1u << 21)
                             ? stft_stream()
                             : nullptr;
      FpWorkspace::Pipe *pp = stft ? &ws->pipes[pipe] : nullptr;
      if (pp) {
        // ... and only while the OTHER pipe still has work queued: a call that is alone on the device keeps all CUs
        // (same workspace and events, the first pass simply stays on the library stream)
        const FpWorkspace::Pipe &other = ws->pipes[pipe ^ 1];
        const bool busy = other.consumed_valid && hipEventQuery(other.consumed) == hipErrorNotReady;
        (void)hipGetLastError();
        if (!busy) stft = stream;
      }
      DeviceBuffer<double> &chroma_buf = pp ? pp->chroma : ws->chroma;
      if (!(s = chroma_buf.reserve(frames * kBands)).ok()) return s;
      if (!pp && !(s = ws->feat.reserve(std::max<uint64_t>(rows, 1) * kBands)).ok()) return s;
      // (a pipelined call gets descriptor slots of its own: the other pipe's table must stay resident)
      FpWorkspace::Descriptors &desc = ws->slot(pp ? FpWorkspace::kDescriptorSlots - 2 + (size_t)pipe : chunk++);
      bool uploaded = false;
      if (!(s = desc.upload.put(&desc.streams, &desc.stage, meta, stream, &uploaded)).ok()) return s;
      const int n = (int)meta.size();
      if (!ws->lds_attr_set) {
        const void *variants[6] = {reinterpret_cast<const void *>(stft_chroma_kernel<1, 0, false>),
                                   reinterpret_cast<const void *>(stft_chroma_kernel<2, 0, false>),
                                   reinterpret_cast<const void *>(stft_chroma_kernel<1, 0, true>),
                                   reinterpret_cast<const void *>(stft_chroma_kernel<2, 0, true>),
                                   reinterpret_cast<const void *>(stft_chroma_kernel<1, 0, true, 3>),
                                   reinterpret_cast<const void *>(stft_chroma_kernel<2, 0, true, 3>)};
        for (const void *fn : variants)
          NEEDLE_HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize,
                                             (int)(core::kLds2Slots * sizeof(cd))));
        ws->lds_attr_set = true;
      }
      int cus = 256;
      {
        int dev = 0;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        cus = std::max(cus, 1);
      }
      // Pairs per workgroup.  The device holds `slots` workgroups at a time.  A launch of more than
      // about two rounds of workgroups balances itself (workgroups retire at different times and the dispatcher
      // backfills: measured, a "whole rounds" choice of the size changed 7- and 14-episode launches by < 2 %), so
      // long launches keep kPairsPerBlock.  A SHORT launch -- one rank's share of a sharded job, a single file --
      // is cut so that every slot gets one workgroup: 4 episodes x 24 min = 11 626 pairs run as 506 workgroups of
      // 23 pairs (0.153 ms) instead of 727 of 16 (0.162 ms); one episode as 485 workgroups of 6 instead of 182 of 16.
      auto pairs_per_block = [&](uint64_t slots, uint32_t long_launch = kPairsPerBlock) {
        uint32_t ppb = long_launch;
        if (pairs < (long_launch > (uint32_t)kPairsPerBlock ? 40 : 2 * kPairsPerBlock) * slots)
          ppb = (uint32_t)std::min<uint64_t>(40, std::max<uint64_t>(4, (pairs + slots - 1) / slots));
        if (const char *e = getenv("NEEDLE_STFT_PAIRS")) ppb = (uint32_t)std::max(1, atoi(e));
        return ppb;
      };
      // The certified f32 first pass is the default; NEEDLE_HIP_STFT=f64 runs the f64 kernel over everything (the
      // arithmetic the contract is defined on; also taken when a caller asks for the intermediate stages).
      const char *mode_env = getenv("NEEDLE_HIP_STFT");
      const bool certified = !(mode_env && std::strcmp(mode_env, "f64") == 0) && d_chroma_dbg == nullptr && d_feat_dbg == nullptr &&
                             getenv("NEEDLE_HIP_SEPARATE_CLASSIFY") == nullptr && tiles > 0;
      if (certified) {
        float cert_k = 64.0f;  // 35 x the worst |log v32 - log v64| / S observed (profiles/r03_f32_gate.log)
        if (const char *e = getenv("NEEDLE_HIP_CERT_K")) cert_k = std::max(0.0f, (float)atof(e));  // tests: 0 = accept everything
#ifndef NEEDLE_CHUNK_PAIRS
#define NEEDLE_CHUNK_PAIRS 2
#endif
        constexpr uint32_t kChunkPairs = NEEDLE_CHUNK_PAIRS;  // 634 four-pair chunks per 28 x 24 min job are two rounds of the 512 workgroup slots; two-pair chunks fit one
        const uint64_t nchunks = (pairs + kChunkPairs - 1) / kChunkPairs;
        const size_t ctl_words = sizeof(CertWork) / 4 + (size_t)((nchunks + 31) / 32);
        DeviceBuffer<float> &energy_buf = pp ? pp->energy : ws->energy;
        DeviceBuffer<uint32_t> &ctl_buf = pp ? pp->cert_ctl : ws->cert_ctl, &chunk_buf = pp ? pp->chunk_list : ws->chunk_list;
        DeviceBuffer<CertItem> &item_buf = pp ? pp->item_list : ws->item_list;
        if (!(s = energy_buf.reserve(frames * stft::kEnergyParts)).ok() || !(s = ctl_buf.reserve(ctl_words)).ok() ||
            !(s = chunk_buf.reserve(nchunks)).ok() || !(s = item_buf.reserve(std::max<uint64_t>(kept, 1))).ok())
          return s;
        if (pp) {
          for (hipEvent_t *e : {&pp->consumed, &pp->descriptors, &pp->recomputed})
            if (!*e) NEEDLE_HIP_TRY(hipEventCreateWithFlags(e, hipEventDisableTiming));
          for (hipEvent_t *e : {&pp->stft_begin, &pp->stft_done})  // these two also time the kernel (KernelTimer's role)
            if (!*e) NEEDLE_HIP_TRY(hipEventCreate(e));
          // the STFT may start once the previous user of this workspace has read it to the end, and -- only if the
          // descriptor table was uploaded just now -- once that copy has executed (an unconditional wait on the library
          // stream would put the STFT behind the whole previous job, which is the one thing this is here to avoid)
          // (an event that has completed by now needs no packet in the STFT's queue: nothing but the previous first pass
          // should sit in front of this one)
          if (pp->consumed_valid) {
            if (hipEventQuery(pp->consumed) != hipSuccess) {
              (void)hipGetLastError();
              NEEDLE_HIP_TRY(hipStreamWaitEvent(stft, pp->consumed, 0));
            }
          }
          // ... and once the OTHER pipe's first pass is through: two STFTs side by side only slow each other down and
          // leave both tails to run alone afterwards (seen in a kernel trace: pairs of 0.67 / 0.76 ms STFTs, then 0.2 ms
          // of tail kernels on an idle chip); what is wanted beside an STFT is the previous call's TAIL
          // Shared-CU overlap (NEEDLE_HIP_STFT_SHARE, hipctx.hip): behind the other pipe's f64 RECOMPUTATION instead.  That
          // kernel's workgroup (68 KB of LDS, ~250 VGPRs) does not fit the hole a retiring first-pass workgroup leaves
          // (35 KB, 163 VGPRs): started beside a first pass it waits for all of it (kernel trace, profiles/NOTES.md round 4);
          // the kernels behind it (fix-up, scan, simhash) and the certification kernel do fit and run beside it.
          const FpWorkspace::Pipe &other = ws->pipes[pipe ^ 1];
          const bool share = share_mode() == 1;  // (default, 2: behind the first pass only)
          if (share && other.recomputed && other.stft_recorded)
            NEEDLE_HIP_TRY(hipStreamWaitEvent(stft, other.recomputed, 0));
          else if (share_mode() != 2 && other.stft_done && other.stft_recorded)  // (2: both on one stream, in order anyway)
            NEEDLE_HIP_TRY(hipStreamWaitEvent(stft, other.stft_done, 0));
          if (uploaded) {
            NEEDLE_HIP_TRY(hipEventRecord(pp->descriptors, stream));
            NEEDLE_HIP_TRY(hipStreamWaitEvent(stft, pp->descriptors, 0));
          }
        }
        if (!ws->stats) {
          NEEDLE_HIP_TRY(hipMalloc((void **)&ws->stats, sizeof(CertStats)));
          NEEDLE_HIP_TRY(hipMemsetAsync(ws->stats, 0, sizeof(CertStats), stream));
        }
        CertWork *work = reinterpret_cast<CertWork *>(ctl_buf.ptr);
        uint32_t *bitmap = ctl_buf.ptr + sizeof(CertWork) / 4;
        {
          // A pipelined call's first pass carries its events in its own dispatch packet (hipExtLaunchKernelGGL): the
          // completion signal of the kernel is `stft_done`, what the library stream waits for, and nothing but the kernel
          // sits between two first passes on their stream (event records around it were 13 us per job).
          hipStream_t on = pp ? stft : stream;
          const bool bound_timing = pp != nullptr && kernel_timing_on("stft_chroma32");
          std::unique_ptr<KernelTimer> timer;  // event records around the launch: only where there is no pipe
          if (!pp) timer.reset(new KernelTimer("stft_chroma32", on));
          const uint64_t slots = (uint64_t)kStft32WavesPerSimd * (uint64_t)cus;
          // Long launches: 24 pairs per workgroup and, over the last half round of every XCD's part, 12, 6 and 3
          // (stft32_schedule.h): fewer workgroup prologues in the bulk, a short ramp at the end; 0.463 -> 0.453 ms alone at
          // 28 x 24 min against 16 throughout (profiles/NOTES.md).  NEEDLE_STFT_GUIDED: tenths of a round, 0 = off.
          static const int guided = getenv("NEEDLE_STFT_GUIDED") ? atoi(getenv("NEEDLE_STFT_GUIDED")) : 5;
          const uint32_t ppb = pairs_per_block(slots, guided > 0 ? 24u : (uint32_t)kPairsPerBlock);
          const Stft32Schedule schedule = stft32_schedule(pairs, ppb, (slots + 7) / 8, guided > 0, (uint32_t)std::max(guided, 1));
          if (!(s = launch_stft_chroma32(channels, schedule, on, d_pcm, desc.streams.ptr, n, tab.tw32, tab.win32, tab.bin_slot,
                                         tab.fold_tab, chroma_buf.ptr, energy_buf.ptr, (uint32_t)pairs, ctl_buf.ptr,
                                         (uint32_t)ctl_words, bound_timing ? pp->stft_begin : nullptr,
                                         pp ? pp->stft_done : nullptr)).ok())
            return s;
          if (bound_timing) bind_kernel_events("stft_chroma32", pp->stft_begin, pp->stft_done);
        }
        if (pp) {  // everything behind the first pass stays on the library stream, behind the STFT's event
          pp->stft_recorded = true;
          NEEDLE_HIP_TRY(hipStreamWaitEvent(stream, pp->stft_done, 0));
        }
        {
          KernelTimer timer("features_cert");
          hipLaunchKernelGGL(step == 2 ? features_classify_cert_kernel<true> : features_classify_cert_kernel<false>,
                             dim3((uint32_t)((tiles + kCertWaves - 1) / kCertWaves)), dim3(64 * kCertWaves), 0, stream,
                             chroma_buf.ptr, energy_buf.ptr, desc.streams.ptr, n, tab.thr, step, items_per_tile, d_items,
                             (uint32_t)tiles, cert_k, kChunkPairs, work, bitmap, chunk_buf.ptr, item_buf.ptr);
        }
        {
          KernelTimer timer("stft_fallback");
          // A pipelined call runs BESIDE the next call's first pass (shared-CU overlap): then the 168-VGPR form, whose
          // workgroup fits the hole one retiring first-pass workgroup leaves (the 229-VGPR form needs two and waited a whole
          // first pass for them), on a quarter of the workgroups (each holds its slot for its whole, latency-bound life).
          const bool beside = pp != nullptr && share_mode() == 2 && stft_stream() != nullptr;
          uint32_t grid = (uint32_t)std::min<uint64_t>((beside ? 1ull : 4ull) * (uint64_t)cus / 2, nchunks);
          if (const char *e = getenv("NEEDLE_HIP_FALLBACK_GRID")) grid = (uint32_t)std::min<uint64_t>((uint64_t)std::max(1, atoi(e)), nchunks);  // tuning
          const stft::ChunkList list{chunk_buf.ptr, &work->chunk_count};
          auto launch = [&](auto kernel) {
            hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), core::kLds2Slots * sizeof(cd), stream, d_pcm, desc.streams.ptr, n,
                               tab.tw, tab.wcos, tab.wconst, tab.bin_slot, tab.fold_tab, chroma_buf.ptr, (uint32_t)pairs,
                               kChunkPairs, list);
          };
          if (beside) {
            if (channels == 1) launch(stft_chroma_kernel<1, 0, true, 3>); else launch(stft_chroma_kernel<2, 0, true, 3>);
          } else {
            if (channels == 1) launch(stft_chroma_kernel<1, 0, true>); else launch(stft_chroma_kernel<2, 0, true>);
          }
        }
        if (pp) NEEDLE_HIP_TRY(hipEventRecord(pp->recomputed, stream));
#ifndef NEEDLE_LAB_NO_FIXUP   // (timing laboratory, WRONG results: the job without the fix-up's dispatch -- the most that folding it into the recomputation kernel could save)
        {
          KernelTimer timer("fixup_items");
          hipLaunchKernelGGL(fixup_items_kernel, dim3(64), dim3(256), 0, stream, chroma_buf.ptr, tab.thr, work,
                             item_buf.ptr, d_items, ws->stats, zero_word);
          zeroed = zero_word != nullptr;
        }
#endif
        if (pp) {
          NEEDLE_HIP_TRY(hipEventRecord(pp->consumed, stream));
          pp->consumed_valid = true;
        }
        ws->items_total += kept;
        ws->chunks_total += nchunks;
        NEEDLE_HIP_TRY(hipGetLastError());
        begin = end;
        continue;
      }
      zeroed = false;
      {
        KernelTimer timer("stft_chroma");
        const uint32_t ppb = pairs_per_block(2ull * (uint64_t)cus);
        const uint32_t grid = (uint32_t)(((pairs + ppb - 1) / ppb + 7) / 8 * 8);  // multiple of 8: see the XCD mapping
        auto launch = [&](auto kernel) {
          hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), core::kLds2Slots * sizeof(cd), stream, d_pcm,
                             desc.streams.ptr, n, tab.tw, tab.wcos, tab.wconst, tab.bin_slot, tab.fold_tab, ws->chroma.ptr,
                             (uint32_t)pairs, ppb, stft::ChunkList{nullptr, nullptr});
        };
        if (channels == 1) launch(stft_chroma_kernel<1, 0, false>); else launch(stft_chroma_kernel<2, 0, false>);
      }
      const bool separate = d_feat_dbg != nullptr || getenv("NEEDLE_HIP_SEPARATE_CLASSIFY") != nullptr;
      if (separate) {  // a caller wants the features themselves (tests): kernels 2 and 3 one after the other
        if (rows > 0) {
          KernelTimer timer("fir_norm");
          hipLaunchKernelGGL(fir_norm_kernel, dim3((uint32_t)((rows + 255) / 256)), dim3(256), 0, stream,
                             ws->chroma.ptr, desc.streams.ptr, n, ws->feat.ptr, (uint32_t)rows);
        }
        if (kept > 0) {
          KernelTimer timer("classify");
          hipLaunchKernelGGL(classify_kernel, dim3((uint32_t)((kept + 255) / 256)), dim3(256), 0, stream,
                             ws->feat.ptr, desc.streams.ptr, n, tab.thr, step, d_items, (uint32_t)kept);
        }
      } else if (tiles > 0) {
        KernelTimer timer("features_classify");
        hipLaunchKernelGGL(features_classify_kernel, dim3((uint32_t)((tiles + 3) / 4)), dim3(256), 0, stream,
                           ws->chroma.ptr, desc.streams.ptr, n, tab.thr, step, items_per_tile, d_items, (uint32_t)tiles);
      }
      NEEDLE_HIP_TRY(hipGetLastError());
      if (d_chroma_dbg)
        NEEDLE_HIP_TRY(hipMemcpyAsync(d_chroma_dbg, ws->chroma.ptr, frames * kBands * sizeof(double),
                                      hipMemcpyDeviceToDevice, stream));
      if (d_feat_dbg && rows)
        NEEDLE_HIP_TRY(hipMemcpyAsync(d_feat_dbg, ws->feat.ptr, rows * kBands * sizeof(double),
                                      hipMemcpyDeviceToDevice, stream));
      // the next chunk reuses the chroma / feature workspaces: safe without a host wait, the stream runs in order
      // (and every chunk has its own descriptor slot)
    }
    begin = end;
  }
  if (zeroed_out) *zeroed_out = zeroed;
  if (sync) NEEDLE_HIP_TRY(hipStreamSynchronize(stream));
  return Status::Ok();
}

// {items fingerprinted, items recomputed in f64, chunks of frame pairs, chunks recomputed} since the last reset, on the
// current device; waits for the library stream.
Status gpu_fingerprint_cert_stats(uint64_t out[4], bool reset) {
  std::lock_guard<std::recursive_mutex> gpu_lock(gpu_mutex());
  Status s = ensure_device();
  if (!s.ok()) return s;
  FpWorkspace *ws = workspace();
  CertStats host{0, 0};
  hipStream_t stream = library_stream();
  if (ws->stats) {
    NEEDLE_HIP_TRY(hipMemcpyAsync(&host, ws->stats, sizeof(host), hipMemcpyDeviceToHost, stream));
    if (reset) NEEDLE_HIP_TRY(hipMemsetAsync(ws->stats, 0, sizeof(CertStats), stream));
    NEEDLE_HIP_TRY(hipStreamSynchronize(stream));
  }
  out[0] = ws->items_total;
  out[1] = host.items_recomputed;
  out[2] = ws->chunks_total;
  out[3] = host.chunks_recomputed;
  if (reset) ws->items_total = ws->chunks_total = 0;
  return Status::Ok();
}

// Audit (include/needle_hip.h needle_hip_fingerprint_audit_device): both transforms over the same resident PCM, every
// kept item compared on the device.  out = {items, accepted by the first pass, accepted items whose f32 bits are not the
// f64 item, items of d_items that are not the f64 item}; *max_ratio = max |log v32 - log v64| / S over accepted items.
Status gpu_fingerprint_audit_device(const int16_t *d_pcm, const std::vector<StreamSpan> &spans, int channels, uint32_t step,
                                    const uint32_t *d_items, uint64_t out[4], double *max_ratio, double *max_sigma) {
  std::lock_guard<std::recursive_mutex> gpu_lock(gpu_mutex());
  if (channels != 1 && channels != 2) return Status::Make(NeedleError_InvalidArgument, "fingerprint: channels must be 1 or 2");
  if (step == 0) return Status::Make(NeedleError_InvalidArgument, "fingerprint: step must be >= 1");
  Status s = ensure_device();
  if (!s.ok()) return s;
  FpTables tab;
  if (!(s = get_tables(&tab)).ok()) return s;
  hipStream_t stream = library_stream();
  // buffers of its own, freed on return: an audit must not disturb the workspaces of jobs in flight
  DeviceBuffer<double> chroma32, chroma64;
  DeviceBuffer<float> energy;
  DeviceBuffer<uint32_t> ctl;
  DeviceBuffer<FpStream> d_streams;
  DeviceBuffer<AuditCounts> d_counts;
  if (!(s = d_counts.reserve(1)).ok() || !(s = ctl.reserve(64)).ok()) return s;
  NEEDLE_HIP_TRY(hipMemsetAsync(d_counts.ptr, 0, sizeof(AuditCounts), stream));
  float cert_k = 64.0f;
  if (const char *e = getenv("NEEDLE_HIP_CERT_K")) cert_k = std::max(0.0f, (float)atof(e));
  int cus = 256;
  {
    int dev = 0;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    cus = std::max(cus, 1);
  }
  {
    const void *variants[2] = {reinterpret_cast<const void *>(stft_chroma_kernel<1, 0, false>),
                               reinterpret_cast<const void *>(stft_chroma_kernel<2, 0, false>)};
    for (const void *fn : variants)
      NEEDLE_HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(core::kLds2Slots * sizeof(cd))));
  }
  const uint64_t kMaxFramesPerChunk = 4u << 20;
  const uint32_t items_per_tile = (uint32_t)std::min<uint64_t>(64, (uint64_t)(kTileRowsMax - 16) / step + 1);
  size_t begin = 0;
  while (begin < spans.size()) {
    std::vector<FpStream> meta;
    uint64_t frames = 0, rows = 0, kept = 0, pairs = 0, tiles = 0;
    size_t end = begin;
    while (end < spans.size()) {
      const size_t samples = spans[end].num_values / (size_t)channels;
      const uint64_t f = num_frames(samples);
      if (!meta.empty() && frames + f > kMaxFramesPerChunk) break;
      if (f > 0xFFFFFFF0ull) return Status::Make(NeedleError_InvalidArgument, "fingerprint: stream too long");
      FpStream m;
      m.pcm_off = spans[end].pcm_off;
      m.item_off = spans[end].item_off;
      m.frames = (uint32_t)f;
      m.frame_base = (uint32_t)frames;
      m.fir_rows = f >= (uint64_t)kFirTaps ? (uint32_t)(f - (kFirTaps - 1)) : 0;
      m.fir_base = (uint32_t)rows;
      m.kept = (uint32_t)num_kept(samples, step);
      m.kept_base = (uint32_t)kept;
      m.pair_base = (uint32_t)pairs;
      m.tile_base = (uint32_t)tiles;
      tiles += (m.kept + items_per_tile - 1) / items_per_tile;
      pairs += (m.frames + 1) / 2;
      frames += m.frames;
      rows += m.fir_rows;
      kept += m.kept;
      meta.push_back(m);
      end++;
    }
    if (frames > 0 && tiles > 0) {
      if (!(s = chroma32.reserve(frames * kBands)).ok() || !(s = chroma64.reserve(frames * kBands)).ok() ||
          !(s = energy.reserve(frames * stft::kEnergyParts)).ok() || !(s = d_streams.reserve(meta.size())).ok())
        return s;
      NEEDLE_HIP_TRY(hipStreamSynchronize(stream));  // the previous chunk still reads d_streams; `meta` is pageable
      NEEDLE_HIP_TRY(hipMemcpyAsync(d_streams.ptr, meta.data(), meta.size() * sizeof(FpStream), hipMemcpyHostToDevice, stream));
      NEEDLE_HIP_TRY(hipStreamSynchronize(stream));
      const int n = (int)meta.size();
      const uint32_t grid = (uint32_t)(((pairs + kPairsPerBlock - 1) / kPairsPerBlock + 7) / 8 * 8);
      if (!(s = launch_stft_chroma32(channels, stft32_schedule(pairs, kPairsPerBlock, 0, false), stream, d_pcm, d_streams.ptr, n,
                                     tab.tw32, tab.win32, tab.bin_slot, tab.fold_tab, chroma32.ptr, energy.ptr, (uint32_t)pairs,
                                     ctl.ptr, 4)).ok())
        return s;
      auto launch = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), core::kLds2Slots * sizeof(cd), stream, d_pcm, d_streams.ptr, n, tab.tw,
                           tab.wcos, tab.wconst, tab.bin_slot, tab.fold_tab, chroma64.ptr, (uint32_t)pairs, (uint32_t)kPairsPerBlock,
                           stft::ChunkList{nullptr, nullptr});
      };
      if (channels == 1) launch(stft_chroma_kernel<1, 0, false>); else launch(stft_chroma_kernel<2, 0, false>);
      hipLaunchKernelGGL(audit_items_kernel, dim3((uint32_t)tiles), dim3(64), 0, stream, chroma32.ptr, energy.ptr, chroma64.ptr,
                         d_streams.ptr, n, tab.thr, step, items_per_tile, d_items, (uint32_t)tiles, cert_k, d_counts.ptr);
      NEEDLE_HIP_TRY(hipGetLastError());
    }
    begin = end;
  }
  AuditCounts host;
  NEEDLE_HIP_TRY(hipMemcpyAsync(&host, d_counts.ptr, sizeof(host), hipMemcpyDeviceToHost, stream));
  NEEDLE_HIP_TRY(hipStreamSynchronize(stream));
  out[0] = host.items;
  out[1] = host.accepted;
  out[2] = host.accepted_wrong;
  out[3] = host.final_wrong;
  double ratio = 0.0, sigma = 0.0;
  std::memcpy(&ratio, &host.max_ratio_bits, sizeof(double));
  std::memcpy(&sigma, &host.max_sigma_bits, sizeof(double));
  if (max_ratio) *max_ratio = ratio;
  if (max_sigma) *max_sigma = sigma;
  return Status::Ok();
}

namespace {

// Per device, kept for the life of the process like the other workspaces (never destroyed: HIP may already be
// gone when static destructors run).  The arenas grow to the largest batch seen (at most 2 GiB of PCM).
struct HostEntryWorkspace {
  DeviceBuffer<int16_t> d_pcm, d_mono;
  DeviceBuffer<uint32_t> d_items;
};

HostEntryWorkspace *host_entry_workspace() {
  static std::mutex mu;
  static std::map<int, HostEntryWorkspace *> all;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lock(mu);
  HostEntryWorkspace *&w = all[dev];
  if (!w) w = new HostEntryWorkspace();
  return w;
}

// Plans device batches over the streams and, inside a batch, overlaps the host -> device copies with the kernels:
// `upload` copies the batch's streams in order on the upload stream and reports each stream as its last copy is
// enqueued; every time about `group_bytes` of PCM have been enqueued, an event is recorded behind them and the
// (resampler +) fingerprinter of those streams is launched on the library stream behind that event.  The copy
// engine therefore never waits for kernels and the kernels of all but the last group are hidden under the copies
// that follow.  Items go to the host (`items`) or stay on the device (`d_items_out` + `item_off_out`).
using BatchUpload = std::function<Status(size_t begin, size_t end, const std::vector<uint64_t> &in_off, int16_t *d_pcm,
                                         hipStream_t stream, const StreamIssued &issued)>;

struct OverlapEvents {  // per device, reused by every call (guarded by gpu_mutex())
  hipEvent_t landed = nullptr, batch_done = nullptr, entry = nullptr;
};
OverlapEvents *overlap_events() {
  static std::mutex mu;
  static std::map<int, OverlapEvents *> all;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lock(mu);
  OverlapEvents *&e = all[dev];
  if (!e) {
    e = new OverlapEvents();
    (void)hipEventCreateWithFlags(&e->landed, hipEventDisableTiming);
    (void)hipEventCreateWithFlags(&e->batch_done, hipEventDisableTiming);
    (void)hipEventCreateWithFlags(&e->entry, hipEventDisableTiming);
  }
  return e;
}

Status fingerprint_in_batches(const std::vector<size_t> &num_values, int channels, uint32_t step,
                              std::vector<std::vector<uint32_t>> *items, int rate, const BatchUpload &upload,
                              uint32_t *d_items_out = nullptr, const std::vector<uint64_t> *item_off_out = nullptr) {
  std::lock_guard<std::recursive_mutex> gpu_lock(gpu_mutex());
  Status s = ensure_device();
  if (!s.ok()) return s;
  if (channels != 1 && channels != 2)
    return Status::Make(NeedleError_InvalidArgument, "fingerprint: channels must be 1 or 2");
  if (step == 0) return Status::Make(NeedleError_InvalidArgument, "fingerprint: step must be >= 1");
  const bool resample = rate != kSampleRate;
  const size_t n = num_values.size();
  if (items) items->assign(n, {});
  // Batches bounded by bytes so the device arena stays modest for huge libraries.
  uint64_t kMaxBatchValues = 1ull << 30;  // 2 GiB of s16
  if (const char *e = getenv("NEEDLE_HIP_MAX_BATCH_VALUES")) kMaxBatchValues = (uint64_t)std::max(1ll, atoll(e));  // tests
  uint64_t group_values = (32ull << 20) / sizeof(int16_t);
  if (const char *e = getenv("NEEDLE_HIP_LAUNCH_GROUP_BYTES")) group_values = (uint64_t)std::max(2ll, atoll(e)) / sizeof(int16_t);
  hipStream_t stream = library_stream(), up = upload_stream();
  // Every exit of this function -- the error returns included -- waits for the copies already enqueued: they read the
  // caller's (pinned) buffers and the slab ring asynchronously, and the contract is that those are free on return.
  struct DrainUploads {
    hipStream_t s;
    ~DrainUploads() { (void)hipStreamSynchronize(s); }
  } drain_uploads{up};
  OverlapEvents *ev = overlap_events();
  HostEntryWorkspace *ws = host_entry_workspace();  // grow-only arenas, guarded by gpu_mutex()
  DeviceBuffer<int16_t> &d_pcm = ws->d_pcm, &d_mono = ws->d_mono;
  DeviceBuffer<uint32_t> &d_items = ws->d_items;
  size_t begin = 0, descriptor_slot = 0;
  bool first_batch = true;
  while (begin < n) {
    std::vector<StreamSpan> spans;        // what the fingerprinter reads (11025 Hz; mono if resampled)
    std::vector<ResampleSpan> rspans;     // what the resampler reads, when the input rate differs
    std::vector<uint64_t> in_off;
    uint64_t values = 0, mono = 0, kept = 0;
    size_t end = begin;
    while (end < n) {
      if (!spans.empty() && values + num_values[end] > kMaxBatchValues) break;
      const size_t in_samples = num_values[end] / (size_t)channels;
      const size_t out_samples = resample ? resample_out_len(in_samples, rate) : in_samples;
      const uint64_t item_off = d_items_out ? (*item_off_out)[end] : kept;
      in_off.push_back(values);
      if (resample) {
        rspans.push_back(ResampleSpan{values, in_samples, mono});
        spans.push_back(StreamSpan{mono, out_samples, item_off});
        mono += (out_samples + 1) & ~(uint64_t)1;
      } else {
        spans.push_back(StreamSpan{values, num_values[end], item_off});
      }
      values += (num_values[end] + 7) & ~(uint64_t)7;  // keep every stream 16-byte aligned in the arena
      kept += num_kept(out_samples, step);
      end++;
    }
    const bool trace = getenv("NEEDLE_HIP_TRACE") != nullptr;
    auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
      if (!trace) return;
      const auto now = std::chrono::steady_clock::now();
      std::fprintf(stderr, "[needle_hip] fingerprint_host %s: %.2f ms\n", what,
                   std::chrono::duration<double, std::milli>(now - t0).count());
      t0 = now;
    };
    if (!(s = d_pcm.reserve(std::max<uint64_t>(values, 1))).ok()) return s;
    if (!d_items_out && !(s = d_items.reserve(std::max<uint64_t>(kept, 1))).ok()) return s;
    if (resample && !(s = d_mono.reserve(std::max<uint64_t>(mono, 1))).ok()) return s;
    uint32_t *const d_out = d_items_out ? d_items_out : d_items.ptr;
    lap("device allocations");
    // the copies must not overtake kernels that still read the PCM arena: those of the previous batch, or of an
    // earlier call on the library stream
    NEEDLE_HIP_TRY(hipEventRecord(first_batch ? ev->entry : ev->batch_done, stream));
    NEEDLE_HIP_TRY(hipStreamWaitEvent(up, first_batch ? ev->entry : ev->batch_done, 0));
    first_batch = false;
    size_t launched = 0;       // streams of this batch whose kernels have been enqueued
    uint64_t pending_values = 0;
    auto launch_group = [&](size_t upto) -> Status {  // streams [launched, upto) of the batch have been enqueued on `up`
      if (upto <= launched) return Status::Ok();
      NEEDLE_HIP_TRY(hipEventRecord(ev->landed, up));
      NEEDLE_HIP_TRY(hipStreamWaitEvent(stream, ev->landed, 0));
      const std::vector<StreamSpan> group(spans.begin() + launched, spans.begin() + upto);
      Status gs;
      if (resample) {  // decode-rate PCM -> mono 11025 Hz, on the device, then straight into the fingerprinter
        const std::vector<ResampleSpan> rgroup(rspans.begin() + launched, rspans.begin() + upto);
        gs = gpu_resample_device(d_pcm.ptr, rgroup, channels, rate, d_mono.ptr, false);
        if (gs.ok()) gs = gpu_fingerprint_device(d_mono.ptr, group, 1, step, d_out, false, nullptr, nullptr, descriptor_slot++);
      } else {
        gs = gpu_fingerprint_device(d_pcm.ptr, group, channels, step, d_out, false, nullptr, nullptr, descriptor_slot++);
      }
      launched = upto;
      pending_values = 0;
      return gs;
    };
    const StreamIssued issued = [&](size_t i) -> Status {  // i: index inside the batch
      pending_values += num_values[begin + i];
      if (pending_values >= group_values) return launch_group(i + 1);
      return Status::Ok();
    };
    if (!(s = upload(begin, end, in_off, d_pcm.ptr, up, issued)).ok()) return s;
    if (!(s = launch_group(end - begin)).ok()) return s;
    lap("upload + kernel launches");
    if (items) {
      std::vector<uint32_t> host(std::max<uint64_t>(kept, 1));
      NEEDLE_HIP_TRY(hipMemcpyAsync(host.data(), d_items.ptr, kept * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
      NEEDLE_HIP_TRY(hipStreamSynchronize(stream));
      for (size_t i = begin; i < end; i++) {
        const size_t in_samples = num_values[i] / (size_t)channels;
        const size_t k = num_kept(resample ? resample_out_len(in_samples, rate) : in_samples, step);
        (*items)[i].assign(host.begin() + spans[i - begin].item_off, host.begin() + spans[i - begin].item_off + k);
      }
      lap("download + scatter");
    }
    begin = end;
  }
  // every copy out of host memory has executed when this returns (the callers' buffers and the slab ring are free)
  NEEDLE_HIP_TRY(hipStreamSynchronize(up));
  return Status::Ok();  // (drain_uploads then finds the stream idle)
}

}  // namespace

Status gpu_fingerprint_host(const std::vector<const int16_t *> &pcm, const std::vector<size_t> &num_values,
                            int channels, uint32_t step, std::vector<std::vector<uint32_t>> *items, int rate) {
  if (pcm.size() != num_values.size())
    return Status::Make(NeedleError_InvalidArgument, "fingerprint: one length per stream is required");
  return fingerprint_in_batches(
      num_values, channels, step, items, rate,
      [&](size_t begin, size_t end, const std::vector<uint64_t> &in_off, int16_t *d_pcm, hipStream_t up,
          const StreamIssued &issued) -> Status {
        return gpu_upload_pcm(std::vector<const int16_t *>(pcm.begin() + begin, pcm.begin() + end),
                              std::vector<size_t>(num_values.begin() + begin, num_values.begin() + end), in_off, d_pcm, up,
                              issued);
      });
}

Status gpu_fingerprint_streamed(const std::vector<size_t> &num_values, const PcmReader &read, unsigned readers,
                                int channels, uint32_t step, std::vector<std::vector<uint32_t>> *items, int rate) {
  return fingerprint_in_batches(
      num_values, channels, step, items, rate,
      [&](size_t begin, size_t end, const std::vector<uint64_t> &in_off, int16_t *d_pcm, hipStream_t up,
          const StreamIssued &issued) -> Status {
        const PcmReader shifted = [&](size_t stream, uint64_t first, uint64_t count, int16_t *dst) {
          return read(begin + stream, first, count, dst);
        };
        return gpu_upload_pcm_streamed(std::vector<size_t>(num_values.begin() + begin, num_values.begin() + end), in_off,
                                       shifted, readers, d_pcm, up, issued);
      });
}

Status gpu_fingerprint_streamed_device(const std::vector<const int16_t *> &pcm, const std::vector<size_t> &num_values,
                                       int channels, uint32_t step, uint32_t *d_items,
                                       const std::vector<uint64_t> &item_off) {
  if (pcm.size() != num_values.size() || item_off.size() != num_values.size())
    return Status::Make(NeedleError_InvalidArgument, "fingerprint: one length and one item offset per stream are required");
  return fingerprint_in_batches(
      num_values, channels, step, nullptr, kSampleRate,
      [&](size_t begin, size_t end, const std::vector<uint64_t> &in_off, int16_t *d_pcm, hipStream_t up,
          const StreamIssued &issued) -> Status {
        return gpu_upload_pcm(std::vector<const int16_t *>(pcm.begin() + begin, pcm.begin() + end),
                              std::vector<size_t>(num_values.begin() + begin, num_values.begin() + end), in_off, d_pcm, up,
                              issued);
      },
      d_items, &item_off);
}

}  // namespace needle
