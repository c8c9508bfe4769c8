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
#include "fp_core.h"
#include "hipctx.h"

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

struct FpStream {
  uint64_t pcm_off;     // s16 values
  uint64_t item_off;    // where this stream's kept items go in d_items
  uint32_t frames;
  uint32_t frame_base;  // prefix of frames
  uint32_t fir_rows;    // frames - 4 (or 0)
  uint32_t fir_base;
  uint32_t kept;
  uint32_t kept_base;
  uint32_t pair_base;   // prefix of ceil(frames / 2): the STFT kernel transforms two frames per FFT
  uint32_t tile_base;   // prefix of ceil(kept / items per tile): tiles of features_classify_kernel
};

// ---- constant tables, generated on the host in double and uploaded once per device --------------------
struct FpTables {
  cd *tw = nullptr;                 // [4096] e^{-2 pi i k/4096}
  uint16_t *bin_slot = nullptr;     // [kNumBins] position of bin (kMinBin + i) in the class-sorted order
  double *wcos = nullptr;           // [512] cos(theta (i - 256)), theta = 2 pi / 4095: window recurrence seeds
  core::WindowConst wconst;         // fp_core.h window_step
  uint16_t *class_bins = nullptr;   // [kNumBins] spectrum bins grouped by pitch class
  uint32_t *class_start = nullptr;  // [13]
  core::ClassifierThresholds *thr = nullptr;
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
  // chromaprint Chroma::PrepareNotes: bin -> pitch class
  std::vector<std::vector<uint16_t>> by_class(kBands);
  for (int i = core::kMinBin; i < core::kMaxBin; i++) {
    double freq = (double)i * kSampleRate / kFrameSize;
    double octave = std::log(freq / (440.0 / 16.0)) / std::log(2.0);
    double note = kBands * (octave - std::floor(octave));
    by_class[(int)(signed char)note].push_back((uint16_t)i);
  }
  std::vector<uint16_t> bins;
  std::vector<uint32_t> start(kBands + 1, 0);
  for (int c = 0; c < kBands; c++) {
    start[c] = (uint32_t)bins.size();
    bins.insert(bins.end(), by_class[c].begin(), by_class[c].end());
  }
  start[kBands] = (uint32_t)bins.size();
  core::ClassifierThresholds thr;
  for (int i = 0; i < 16; i++)
    for (int j = 0; j < 3; j++) thr.e[i][j] = std::exp(kThresholds[i][j]);

  std::vector<uint16_t> bin_slot(core::kNumBins);
  for (size_t pos = 0; pos < bins.size(); pos++)  // where the bin's power goes in the LDS image (frame A)
    bin_slot[bins[pos] - core::kMinBin] = (uint16_t)core::dif_power_slot((int)pos);
  for (int c = 0; c < kBands; c++)
    if (start[c + 1] - start[c] > (uint32_t)(core::kClassLanes * core::kClassLaneMax))
      return Status::Make(NeedleError_Unknown, "pitch class larger than the fold's lane budget");
  FpTables t;
  NEEDLE_HIP_TRY(hipMalloc((void **)&t.bin_slot, bin_slot.size() * sizeof(uint16_t)));
  NEEDLE_HIP_TRY(hipMemcpy(t.bin_slot, bin_slot.data(), bin_slot.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
  NEEDLE_HIP_TRY(hipMalloc((void **)&t.tw, tw.size() * sizeof(cd)));
  NEEDLE_HIP_TRY(hipMalloc((void **)&t.wcos, wcos.size() * sizeof(double)));
  t.wconst = wconst;
  NEEDLE_HIP_TRY(hipMalloc((void **)&t.class_bins, bins.size() * sizeof(uint16_t)));
  NEEDLE_HIP_TRY(hipMalloc((void **)&t.class_start, start.size() * sizeof(uint32_t)));
  NEEDLE_HIP_TRY(hipMalloc((void **)&t.thr, sizeof(thr)));
  NEEDLE_HIP_TRY(hipMemcpy(t.tw, tw.data(), tw.size() * sizeof(cd), hipMemcpyHostToDevice));
  NEEDLE_HIP_TRY(hipMemcpy(t.wcos, wcos.data(), wcos.size() * sizeof(double), hipMemcpyHostToDevice));
  NEEDLE_HIP_TRY(hipMemcpy(t.class_bins, bins.data(), bins.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
  NEEDLE_HIP_TRY(hipMemcpy(t.class_start, start.data(), start.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  NEEDLE_HIP_TRY(hipMemcpy(t.thr, &thr, sizeof(thr), hipMemcpyHostToDevice));
  g_tables[dev] = t;
  *out = t;
  return Status::Ok();
}

// index of the stream whose [base, next base) range holds g; `base` is a field of FpStream
template <uint32_t FpStream::*BASE>
__device__ __forceinline__ int find_stream(const FpStream *streams, int n, uint32_t g) {
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    int mid = (lo + hi + 1) >> 1;
    if (streams[mid].*BASE <= g) lo = mid; else hi = mid - 1;
  }
  return lo;
}

// ---- kernel 1: one 256-thread workgroup per PAIR of consecutive frames ---------------------------------------
// z = frameA + i*frameB through one 4096-point complex FFT (fp_core.h, radix 16 x 3, padded LDS), split into
// the two real spectra, |X|^2 over bins 10..1307 folded into 12 pitch classes per frame.  A workgroup walks
// kPairsPerBlock CONSECUTIVE pairs of one region of the batch, so the 3x overlap between neighbouring frames
// (hop 1365 of 4096) is re-read from this XCD's L2 rather than from HBM.
constexpr int kPairsPerBlock = 16;  // default; NEEDLE_STFT_PAIRS overrides for tuning

// LDS-only workgroup barrier: waits for this wave's LDS traffic, not for its outstanding global loads
// (__syncthreads() would also drain vmcnt and with it the prefetch of the next pair's PCM).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Orders this wave's LDS writes before its later LDS reads for the compiler; the hardware executes one wave's LDS
// operations in order, so lanes of the same wave see each other's data without a workgroup barrier.
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// x from another lane of the row, by a DPP control word (no LDS round trip, unlike __shfl_xor)
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double x) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}

struct PairSrc {
  const int16_t *a, *b;  // first value of frame A / frame B (B = A when the stream has an odd frame count)
  uint64_t row;          // chroma row of frame A
  bool has_b;
};

template <int CH>
__global__ __launch_bounds__(256, 2) void stft_chroma_kernel(const int16_t *__restrict__ pcm,
                                                             const FpStream *__restrict__ streams, int num_streams,
                                                             const cd *__restrict__ tw,
                                                             const double *__restrict__ wcos, core::WindowConst wconst,
                                                             const uint16_t *__restrict__ bin_slot,
                                                             const uint32_t *__restrict__ class_start,
                                                             double *__restrict__ chroma, uint32_t total_pairs,
                                                             uint32_t pairs_per_block) {
  extern __shared__ cd lds[];  // core::kLds2Slots complex slots
  using raw_t = typename std::conditional<CH == 1, int16_t, int>::type;  // one sample, or one packed L|R pair
  const int t = threadIdx.x;
  // Workgroups are dealt to the 8 XCDs round-robin (blockIdx.x & 7) and each XCD has its own L2.  Neighbouring
  // stretches of the timeline share 2731 of their samples (the frame overlap), so each XCD gets one contiguous
  // eighth of the timeline: the workgroups that run side by side on an XCD are then neighbours in time and the
  // overlap is re-read from that XCD's L2 (the grid is a multiple of 8).  Measured: fabric fetches per launch
  // 482 MB either way for 445 MB of PCM -- the boundary overlap of a plain mapping is only 28 MB and was mostly
  // caught by the memory-side cache already -- and no change in kernel time; kept because it is never worse.
  const uint32_t per_xcd = gridDim.x >> 3;
  const uint32_t logical = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
  const uint32_t first = logical * pairs_per_block;
  const uint32_t last = min(total_pairs, first + pairs_per_block);
  if (first >= last) return;
  const cd base0 = tw[t], base1 = tw[16 * (t & 15)];  // W_4096^t, W_4096^{16 n0}: loop-invariant twiddle bases

  // ---- loop invariants of this thread, packed so they cost few registers --------------------------------------
  // where the powers of its six bins (register j of stage 2) go in the class-sorted LDS image
  uint32_t slot_pk[core::kBinsPerThread / 2];
#pragma unroll
  for (int j = 0; j < core::kBinsPerThread; j++) {
    const int kf = core::dif_bin_of(t, j);
    const uint32_t idx = (kf >= core::kMinBin && kf < core::kMaxBin) ? bin_slot[kf - core::kMinBin] : core::kPowerTrashSlot;
    slot_pk[j >> 1] = (j & 1) ? (slot_pk[j >> 1] | (idx << 16)) : idx;
  }
  if (t == 0) lds[core::kPowerZeroSlot] = cd{0.0, 0.0};  // first read after the loop's barriers
  // its share of the pitch-class fold: 12 classes x 16 lanes (one DPP row per class), each lane both frames
  const bool folds = t < kBands * core::kClassLanes;
  const int fold_c = t >> 4, fold_l = t & 15;
  const uint32_t fold_bounds = folds ? (class_start[fold_c] | (class_start[fold_c + 1] << 16)) : 0;

  // ---- the stream (region of the batch) the current pair belongs to; consecutive pairs rarely change it ------
  int si = find_stream<&FpStream::pair_base>(streams, num_streams, first);
  FpStream st = streams[si];
  uint32_t st_end = st.pair_base + (st.frames + 1) / 2;
  auto locate = [&](uint32_t g) {  // g must not decrease between calls
    while (g >= st_end) {
      st = streams[++si];
      st_end = st.pair_base + (st.frames + 1) / 2;
    }
    const uint32_t fa = 2 * (g - st.pair_base);
    PairSrc p;
    p.has_b = fa + 1 < st.frames;
    p.a = pcm + st.pcm_off + (uint64_t)fa * kHop * CH;
    p.b = p.has_b ? p.a + kHop * CH : p.a;  // no frame B: read A again, zeroed after conversion
    p.row = (uint64_t)st.frame_base + fa;
    return p;
  };
  using reg_t = int;  // one 16-bit sample sign-extended by the load, or one packed L|R pair
  reg_t ra[16], rb[16];
  // PCM of both frames: issued one pair ahead, while the previous pair's powers are still being produced (the
  // spectrum registers are dead by then).  The window comes from a recurrence (fp_core.h window_step), seeded per
  // thread with cos(theta (t - 256)) and cos(theta t).
  const double wseed_prev = wcos[t], wseed = wcos[t + 256];
  auto issue_loads = [&](const PairSrc &p) {
    const raw_t *qa = reinterpret_cast<const raw_t *>(p.a), *qb = reinterpret_cast<const raw_t *>(p.b);
    // an opaque copy of the thread index keeps these loads (and their addresses) in the loop; laundering the
    // POINTER would do that too but loses its address space: flat loads, which count in lgkmcnt and so stall
    // every LDS-only barrier
    int tt = t;
    asm volatile("" : "+v"(tt));
#pragma unroll
    for (int k = 0; k < 16; k++) {
      ra[k] = (reg_t)qa[tt + 256 * k];
      rb[k] = (reg_t)qb[tt + 256 * k];
    }
  };
  // The pitch-class fold of a pair runs one pair late, between the next pair's sample conversion and its first
  // butterflies: the LDS reads are issued, the conversion hides their latency, then 8 lanes per class add up.
  cd fv[core::kClassLaneMax];
  auto fold_issue = [&]() {
    if (folds) {
      uint32_t fb = fold_bounds;
      asm volatile("" : "+v"(fb));  // recompute the addresses per pair rather than keep them in registers
      core::class_lane_load(lds, (int)(fb & 0xffffu), (int)(fb >> 16), fold_l, fv);
    }
  };
  auto fold_finish = [&](const PairSrc &p) {
    if (folds) {
      cd acc = core::class_lane_add(fv);
      // fixed-order tree over the class's 16 lanes (fp_core.h class_tree_partner)
      acc = cd{acc.x + dpp_f64<0xB1>(acc.x), acc.y + dpp_f64<0xB1>(acc.y)};    // quad_perm [1,0,3,2]
      acc = cd{acc.x + dpp_f64<0x4E>(acc.x), acc.y + dpp_f64<0x4E>(acc.y)};    // quad_perm [2,3,0,1]
      acc = cd{acc.x + dpp_f64<0x141>(acc.x), acc.y + dpp_f64<0x141>(acc.y)};  // row_half_mirror
      acc = cd{acc.x + dpp_f64<0x140>(acc.x), acc.y + dpp_f64<0x140>(acc.y)};  // row_mirror
      if (fold_l == 0) {
        chroma[p.row * kBands + fold_c] = acc.x;
        if (p.has_b) chroma[(p.row + 1) * kBands + fold_c] = acc.y;
      }
    }
  };
  PairSrc cur = locate(first), prev = cur;
  issue_loads(cur);

  for (uint32_t g = first; g < last; g++) {
    // all per-thread address arithmetic is redone per pair from this opaque copy of the thread index: kept
    // loop-invariant by the compiler it costs more registers than the kernel has (spills to scratch)
    int tt = t;
    asm volatile("" : "+v"(tt));
    if (g != first) fold_issue();
    cd r[16];
    double wc = wseed, wc_prev = wseed_prev;
    asm volatile("" : "+v"(wc), "+v"(wc_prev));  // per pair: the 16 window values are not kept across the loop
#pragma unroll
    for (int k = 0; k < 16; k++) {
      int sa, sb;
      if (CH == 1) {
        sa = ra[k];
        sb = rb[k];
      } else {  // AudioProcessor::LoadStereo: (L + R) / 2, C truncation
        sa = ((int)(int16_t)ra[k] + (ra[k] >> 16)) / 2;
        sb = ((int)(int16_t)rb[k] + (rb[k] >> 16)) / 2;
      }
      const double w = core::window_step(wconst, &wc, &wc_prev);
      r[k] = cd{(double)sa * w, (double)sb * w};
    }
    if (!cur.has_b) {  // odd frame count: the stream's last pair has no frame B (uniform branch)
#pragma unroll
      for (int k = 0; k < 16; k++) r[k].y = 0.0;
    }
    if (g != first) fold_finish(prev);
    // in-place decimation-in-frequency stages; which exchanges need a workgroup barrier: fp_core.h
    core::fft16(r);
    lds_barrier();                // every thread has read its share of the previous pair's powers
    core::dif0_store(tt, base0, lds, r);
    lds_barrier();
    core::dif1(tt, base1, lds, r);
    wave_lds_fence();             // stage 1 -> 2 stays inside 16 consecutive lanes
    core::dif2(tt, lds, r);       // r[out16(j)] = Z[bin (t>>4) + 16 (t&15) + 256 j]
    core::dif2_publish(tt, lds, r);  // own slots; only the partner values Z[N - k] other threads need
    lds_barrier();

    uint32_t spk[core::kBinsPerThread / 2];
#pragma unroll
    for (int j = 0; j < core::kBinsPerThread / 2; j++) {
      spk[j] = slot_pk[j];
      asm volatile("" : "+v"(spk[j]));  // unpack per pair: unpacked copies kept across the loop would spill
    }
#pragma unroll
    for (int j = 0; j < core::kBinsPerThread; j++) {
      const uint32_t idx = (spk[j >> 1] >> (16 * (j & 1))) & 0xffffu;
      double pa, pb;
      core::dif_bin_power_any(tt, j, lds, r, &pa, &pb);
      // class-sorted power pairs into dead slots (fp_core.h dif_power_slot): no barrier after the partner reads;
      // bins outside 10..1307 land in a slot nobody reads
      lds[idx] = cd{pa, pb};
    }
    const PairSrc nxt = locate(min(g + 1, last - 1));  // last pair: harmless re-read
    issue_loads(nxt);
    lds_barrier();                // the power image is complete
    prev = cur;
    cur = nxt;
  }
  fold_issue();
  fold_finish(prev);
}

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
                              double *d_feat_dbg, size_t descriptor_slot) {
  std::lock_guard<std::recursive_mutex> gpu_lock(gpu_mutex());
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
    const uint32_t items_per_tile = (uint32_t)std::min<uint64_t>(64, (uint64_t)(kTileRowsMax - 16) / step + 1);
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
      if (!(s = ws->chroma.reserve(frames * kBands)).ok()) return s;
      if (!(s = ws->feat.reserve(std::max<uint64_t>(rows, 1) * kBands)).ok()) return s;
      FpWorkspace::Descriptors &desc = ws->slot(chunk++);
      if (!(s = desc.upload.put(&desc.streams, &desc.stage, meta, stream)).ok()) return s;
      const int n = (int)meta.size();
      if (!ws->lds_attr_set) {
        const void *variants[2] = {reinterpret_cast<const void *>(stft_chroma_kernel<1>),
                                   reinterpret_cast<const void *>(stft_chroma_kernel<2>)};
        for (const void *fn : variants)
          NEEDLE_HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize,
                                             (int)(core::kLds2Slots * sizeof(cd))));
        ws->lds_attr_set = true;
      }
      {
        KernelTimer timer("stft_chroma");
        // Pairs per workgroup.  The device holds `slots` workgroups at a time (2 per CU).  A launch of more than
        // about two rounds of workgroups balances itself (workgroups retire at different times and the dispatcher
        // backfills: measured, a "whole rounds" choice of the size changed 7- and 14-episode launches by < 2 %), so
        // long launches keep kPairsPerBlock.  A SHORT launch -- one rank's share of a sharded job, a single file --
        // is cut so that every slot gets one workgroup: 4 episodes x 24 min = 11 626 pairs run as 506 workgroups of
        // 23 pairs (0.153 ms) instead of 727 of 16 (0.162 ms); one episode as 485 workgroups of 6 instead of 182 of 16.
        uint32_t ppb = kPairsPerBlock;
        {
          int cus = 256, dev = 0;
          (void)hipGetDevice(&dev);
          (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
          const uint64_t slots = 2ull * (uint64_t)std::max(cus, 1);
          if ((pairs + kPairsPerBlock - 1) / kPairsPerBlock < 2 * slots)
            ppb = (uint32_t)std::min<uint64_t>(40, std::max<uint64_t>(4, (pairs + slots - 1) / slots));
        }
        if (const char *e = getenv("NEEDLE_STFT_PAIRS")) ppb = (uint32_t)std::max(1, atoi(e));
        const uint32_t grid = (uint32_t)(((pairs + ppb - 1) / ppb + 7) / 8 * 8);  // multiple of 8: see the XCD mapping
        auto launch = [&](auto kernel) {
          hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), core::kLds2Slots * sizeof(cd), stream, d_pcm,
                             desc.streams.ptr, n, tab.tw, tab.wcos, tab.wconst, tab.bin_slot, tab.class_start, ws->chroma.ptr,
                             (uint32_t)pairs, ppb);
        };
        if (channels == 1) launch(stft_chroma_kernel<1>); else launch(stft_chroma_kernel<2>);
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
  if (sync) NEEDLE_HIP_TRY(hipStreamSynchronize(stream));
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
  return Status::Ok();
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
