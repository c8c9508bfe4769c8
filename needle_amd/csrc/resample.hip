// GPU resampler + down-mix: the step in front of the fingerprinter.  The reference does this with FFmpeg's
// swresample (needle/src/audio/analyzer.rs:180-187,231-282), a third-party library that cannot be reproduced
// bit for bit; this front-end is this project's own specification (oracle/ora_resample.h states it):
// integer down-mix, rational polyphase Kaiser-windowed-sinc FIR to 11025 Hz with f32 coefficients and f32 fused
// multiply-adds in tap order, round-to-nearest-even, clamp.  Integer/f32 work with a fixed operation order, so
// the GPU output equals the oracle's bit for bit.
//
// Each output is a serial chain of T (100-300) FMAs, so the kernel's job is to feed two operands per FMA
// cheaply (measured: the first version, one LDS read and one scattered coefficient load per tap, ran at 4-8 % of
// the HBM rate; the integer VALU work of addressing alone was the next limit).  A workgroup owns a tile of n*L
// consecutive outputs and stages the n*M + T input samples they read in LDS once, as f32.  Lanes are grouped by
// filter PHASE: the n lanes of a group compute the outputs p, p + L, p + 2L, ... of the tile, which share one
// coefficient row and read inputs exactly M samples apart.
//  * samples: four at a time from 16-byte aligned LDS reads at compile-time offsets from one address per lane.
//    The misalignment of a lane's first tap is absorbed by using the coefficient row pre-shifted by 0..3 taps
//    (rows are stored four times, zero-padded: a zero coefficient leaves the accumulator unchanged).  When M is
//    large the region is stored as n rows of M + T samples (the T samples after a row repeat the start of the
//    next row), row pitch odd in 16-byte slots: a lane's window never leaves its row, and the 16 lanes of an LDS
//    lane group, one row apart, hit 16 different banks.
//  * coefficients: each wave copies the rows of its (at most four) phases into a private LDS scratch with one
//    coalesced load per row and reads them back as broadcast LDS reads: a broadcast vector-memory load returns
//    16 bytes to every lane whatever the addresses (16 cycles of the CU's L1 path per 4 taps), a broadcast LDS
//    read costs 4.
//
// Four kernels, chosen by rate (gpu_resample_device): integer decimation with scalar coefficients (44.1 / 22.05 kHz),
// the matrix-core kernel of resample_mfma.h (decimation steps M >= 64 with M % 4 == 0 and rows that fit: 48, 32, 24,
// 16, 8 kHz), the DPP-operand kernel with four outputs per lane (the other row-layout rates: 96 kHz, and the above
// with NEEDLE_HIP_RESAMPLE_QUAD=1), and the general kernel described above.  All four are the same arithmetic.
#include "hipctx.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <mutex>
#include <type_traits>

namespace needle {

namespace {

constexpr int kTarget = 11025;
constexpr int kZeroCrossings = 16;
constexpr double kRolloff = 0.94;
constexpr double kKaiserBeta = 9.0;

struct Design {
  int L = 1, M = 1, T = 0;
  int G = 0;                // 16-byte groups per shifted row
  int n = 0;                // lanes (outputs L apart) that share a phase within a tile
  std::vector<float> coef;  // [L][T]
  float *d_coef = nullptr;  // [4 shifts][L][4 G]: row a holds coef[phase][k - a] at k, zero elsewhere
  // the same rows for resample_quad_kernel: [4 shifts][L][row_len] 16-byte groups, quad_pad zero groups in front and
  // zeros behind, so that a row may be read from kQuadPad groups before its start to `steps` groups past it
  float *d_coefq = nullptr;
  void *d_quad_info = nullptr;  // [ceil(L / 4)] QuadInfo (resample_quad_kernel)
  float *d_coef_plain = nullptr;  // [L][T] as designed (resample_dec_kernel reads row 0)
  int row_len = 0, steps = 0, quad_pad = 0, delta = 0;
  // resample_mfma_kernel (resample_mfma.h): [nblocks][mfma_steps][64] B operands, the blocks' window starts
  float *d_coef_b = nullptr;
  int *d_block_k0 = nullptr;
  std::vector<int> block_k0;
  int mfma_steps = 0, nblocks = 0;  // mfma_steps 0: the kernel does not apply
};

double bessel_i0(double x) {
  double sum = 1.0, term = 1.0;
  for (int k = 1; k <= 40; k++) {
    term *= (x / (2.0 * k)) * (x / (2.0 * k));
    sum += term;
  }
  return sum;
}

int gcd_i(int a, int b) {
  while (b) {
    int t = a % b;
    a = b;
    b = t;
  }
  return a;
}

void design_filter(int rate, Design *d) {
  if (rate == kTarget) {  // nothing to resample: identity (a single unit tap), only the down-mix applies
    d->L = d->M = 1;
    d->T = 2;
    d->coef = {1.0f, 0.0f};
    return;
  }
  const int g = gcd_i(kTarget, rate);
  d->L = kTarget / g;
  d->M = rate / g;
  const double ratio = (double)d->L / (double)d->M;
  const double scale = kRolloff * (ratio < 1.0 ? ratio : 1.0);
  const int half = (int)std::ceil(kZeroCrossings / (ratio < 1.0 ? ratio : 1.0));
  d->T = 2 * half;
  d->coef.assign((size_t)d->L * d->T, 0.f);
  const double pi = 3.14159265358979323846;
  const double i0b = bessel_i0(kKaiserBeta);
  std::vector<double> tmp(d->T);
  for (int p = 0; p < d->L; p++) {
    double sum = 0.0;
    for (int k = 0; k < d->T; k++) {
      const double tau = (double)(k - half + 1) - (double)p / (double)d->L;
      const double x = tau * scale;
      const double sinc = x == 0.0 ? 1.0 : std::sin(pi * x) / (pi * x);
      const double u = tau / (double)half;
      const double win = std::fabs(u) >= 1.0 ? 0.0 : bessel_i0(kKaiserBeta * std::sqrt(1.0 - u * u)) / i0b;
      tmp[k] = scale * sinc * win;
      sum += tmp[k];
    }
    for (int k = 0; k < d->T; k++) d->coef[(size_t)p * d->T + k] = (float)(tmp[k] / sum);
  }
}

std::mutex g_mu;
std::map<std::pair<int, int>, Design> g_designs;  // (device, rate)

struct RsStream {
  uint64_t in_off;   // s16 values into the input arena
  uint64_t n_in;     // samples per channel
  uint64_t out_off;  // samples into the output arena
  uint64_t n_out;
  uint32_t block_base;
  uint32_t pad;
};

constexpr int kMaxRegionSamples = 30000;  // f32 region of a tile in LDS (120 KB) at most
constexpr int kThreads = 512;
constexpr int kRowModeMinM = 64;   // row layout from this decimation step on (below it lanes are few slots apart)
constexpr int kRowModeMaxN = 32;

// Orders this wave's LDS writes before its later LDS reads for the compiler; the hardware executes one wave's LDS
// operations in order.
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct RsGeom {
  int L, M, T, G;   // L/M = 11025/rate in lowest terms, T taps, G 16-byte groups per shifted coefficient row
  int n_log2;       // lanes per phase
  int pitch;        // row layout: 16-byte slots per row (odd); 0 = the region is stored contiguously
  int region_slots; // 16-byte slots of samples
  int delta;        // region[0] is input sample first-tap-of-the-tile minus delta, so that it is a multiple of 4
};

// ROWS_IN_LDS needs the shift of a row to be the same for all lanes of its phase: always so in the row layout,
// and in the contiguous layout when M % 4 == 0.
// VEC4: M % 4 == 0, so every row (and the region) starts a multiple of 4 samples after the stream's first sample
// and may be staged by aligned groups of four.
template <int CH, bool ROWS_IN_LDS, bool VEC4>
__global__ __launch_bounds__(kThreads) void resample_kernel(const int16_t *__restrict__ in,
                                                            const RsStream *__restrict__ streams, int num_streams,
                                                            const float4 *__restrict__ coef, RsGeom geo,
                                                            int16_t *__restrict__ out) {
  extern __shared__ float4 lds4[];  // samples | the tile's outputs (s16) | per-wave coefficient rows
  float *stage = reinterpret_cast<float *>(lds4);
  const int L = geo.L, M = geo.M, G = geo.G, n_log2 = geo.n_log2;
  int lo = 0, hi = num_streams - 1;
  while (lo < hi) {
    int mid = (lo + hi + 1) >> 1;
    if (streams[mid].block_base <= blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const RsStream st = streams[lo];
  const int n = 1 << n_log2;
  const int tile_outputs = n * L;
  const uint64_t tile = blockIdx.x - st.block_base;
  const uint64_t tile_base = tile * (uint64_t)tile_outputs;          // first output of the tile
  const int half = geo.T / 2;
  // input index of region[0]: the first tap of the tile's first output, moved down by delta to a multiple of 4
  const long long first0 = (long long)(tile * (uint64_t)n * (uint64_t)M) - half + 1 - geo.delta;
  const int16_t *src = in + st.in_off;
  // one down-mixed input sample, 0 outside the stream.  Branch-free (the load is always issued, at an index
  // clamped into the stream) so that the unrolled staging loops keep their loads in flight; a stereo sample is
  // one aligned 32-bit load.
  auto sample = [&](long long idx) -> int {
    const bool ok = idx >= 0 && (uint64_t)idx < st.n_in;
    const long long at = idx < 0 ? 0 : ((uint64_t)idx < st.n_in ? idx : (long long)st.n_in - 1);
    int sv;
    if (CH == 1) {
      sv = src[at];
    } else {
      const int v = reinterpret_cast<const int *>(src)[at];
      sv = ((int)(int16_t)v + (v >> 16)) / 2;  // integer down-mix, C truncation
    }
    return ok ? sv : 0;
  };
  // staging: `count` samples from input index `from`, by `nt` cooperating threads of which this is number `me`;
  // whole passes keep eight loads in flight per thread (one at a time would pay a memory latency per pass), the
  // ragged tail is a plain loop (no dummy loads: thousands of workgroups reading one dummy address made that
  // cache line the bottleneck)
  auto stage_run = [&](float *dst, long long from, int count, int me, int nt) {
    constexpr int kStageUnroll = 8;
    int o = me;
    for (; o + (kStageUnroll - 1) * nt < count; o += nt * kStageUnroll) {
      int v[kStageUnroll];
#pragma unroll
      for (int u = 0; u < kStageUnroll; u++) v[u] = sample(from + o + u * nt);
#pragma unroll
      for (int u = 0; u < kStageUnroll; u++) dst[o + u * nt] = (float)v[u];
    }
    for (; o < count; o += nt) dst[o] = (float)sample(from + o);
  };
  // the same by groups of four samples: one 16-byte (stereo) or 8-byte (mono) load and one 128-bit LDS store per
  // group.  `from` and the stream's first sample are multiples of 4 samples from a 16-byte boundary; groups that
  // stick out of the stream take the scalar path.
  auto stage_run4 = [&](float4 *dst, long long from, int groups, int me, int nt) {
    constexpr int kStageUnroll = 4;
    const long long last_group = ((long long)st.n_in - 4) & ~3ll;  // last aligned group wholly inside the stream
    auto group = [&](long long idx) -> float4 {  // branch-free: out-of-range groups load a clamped one (fixed below)
      const long long at = idx < 0 ? 0 : (idx > last_group ? last_group : idx);
      int s0, s1, s2, s3;
      if (CH == 1) {
        const int2 v = *reinterpret_cast<const int2 *>(src + at);
        s0 = (int16_t)v.x; s1 = v.x >> 16; s2 = (int16_t)v.y; s3 = v.y >> 16;
      } else {
        const int4 v = *reinterpret_cast<const int4 *>(src + 2 * at);
        s0 = ((int)(int16_t)v.x + (v.x >> 16)) / 2; s1 = ((int)(int16_t)v.y + (v.y >> 16)) / 2;
        s2 = ((int)(int16_t)v.z + (v.z >> 16)) / 2; s3 = ((int)(int16_t)v.w + (v.w >> 16)) / 2;
      }
      return float4{(float)s0, (float)s1, (float)s2, (float)s3};
    };
    int o = me;
    for (; o + (kStageUnroll - 1) * nt < groups; o += nt * kStageUnroll) {
      float4 v[kStageUnroll];
#pragma unroll
      for (int u = 0; u < kStageUnroll; u++) v[u] = group(from + 4 * (o + u * nt));
#pragma unroll
      for (int u = 0; u < kStageUnroll; u++) dst[o + u * nt] = v[u];
    }
    for (; o < groups; o += nt) dst[o] = group(from + 4 * o);
    // groups that stick out of the stream (only in its first and last tiles): sample by sample, zeros outside
    for (o = me; o < groups; o += nt) {
      const long long idx = from + 4 * o;
      if (idx < 0 || idx > last_group)
        dst[o] = float4{(float)sample(idx), (float)sample(idx + 1), (float)sample(idx + 2), (float)sample(idx + 3)};
    }
  };
  const bool by_groups = VEC4 && (reinterpret_cast<uintptr_t>(src) & 15) == 0 && st.n_in >= 4;
  if (geo.pitch) {  // row r: its own M samples and the overlap into the next row
    const int tpr = kThreads >> n_log2;  // threads per row
    const int r = threadIdx.x / tpr;
    if (by_groups)
      stage_run4(lds4 + geo.pitch * r, first0 + (long long)r * M, (M + 4 * G + 4 + 3) >> 2, threadIdx.x % tpr, tpr);
    else
      stage_run(stage + 4 * geo.pitch * r, first0 + (long long)r * M, M + 4 * G + 4, threadIdx.x % tpr, tpr);
  } else {
    if (by_groups) stage_run4(lds4, first0, geo.region_slots, threadIdx.x, kThreads);
    else stage_run(stage, first0, 4 * geo.region_slots, threadIdx.x, kThreads);
  }
  int16_t *out_tile = reinterpret_cast<int16_t *>(lds4 + geo.region_slots);
  const int rows_per_wave = n >= 64 ? 1 : (64 >> n_log2);
  float4 *scratch = lds4 + geo.region_slots + ((tile_outputs * 2 + 15) >> 4) + (threadIdx.x >> 6) * rows_per_wave * G;
  const int lane = threadIdx.x & 63;
  // Work index of a lane within its wave: a 128-bit LDS read is served in four groups of 16 lanes that are NOT
  // consecutive ({0-3,12-15,20-27}, {4-11,16-19,28-31}, and the same + 32); numbering the lanes group by group
  // gives each hardware group one phase (one coefficient row, 16 sample windows one row pitch apart: 16 banks).
  const int quad = (lane & 31) >> 2;
  const int vlane = (lane & 32) | (((0x96 >> quad) & 1) << 4) | ((quad >> 1) << 2) | (lane & 3);
  __syncthreads();
  const size_t shift_stride = (size_t)L * G;  // float4 per shifted copy of the table
  const int rounds = (tile_outputs + kThreads - 1) / kThreads;
  for (int it = 0; it < rounds; it++) {
    const int idx = it * kThreads + (threadIdx.x - lane) + vlane;
    const int p_first = (idx - vlane) >> n_log2;  // phase class of the wave's first work item
    if (ROWS_IN_LDS) {  // (fetching the rows a round ahead into registers was measured: slower)
      wave_lds_fence();  // the previous round's reads of the scratch are done
      for (int r = 0; r < rows_per_wave; r++) {
        const int pr = p_first + r;
        if (pr < L) {
          const int pm = pr * M, cp = pm / L, phase = pm - cp * L;
          const float4 *row = coef + (size_t)((cp + geo.delta) & 3) * shift_stride + (size_t)phase * G;
          for (int g = lane; g < G; g += 64) scratch[r * G + g] = row[g];
        }
      }
      wave_lds_fence();
    }
    if (idx < tile_outputs) {
      const int p = idx >> n_log2, j = idx & (n - 1);
      const uint64_t m = tile_base + (uint64_t)p + (uint64_t)L * j;
      // pos = m * M = (tile * n * M + j * M) * L + p * M: centre and phase follow from p alone
      const int pm = p * M;
      const int cp = pm / L, phase = pm - cp * L;
      // first tap of this output: sample cp of row j, or sample cp + j M of the contiguous region
      const int rel = (geo.pitch ? cp : cp + j * M) + geo.delta;
      const float4 *xs = lds4 + (geo.pitch ? j * geo.pitch : 0) + (rel >> 2);
      const float4 *row = ROWS_IN_LDS ? scratch + (p - p_first) * G
                                      : coef + (size_t)(rel & 3) * shift_stride + (size_t)phase * G;
      float acc = 0.0f;
      for (int g = 0; g < G; g += 4) {  // G is a multiple of 4: eight loads in flight, then sixteen FMAs in tap order
        float4 x[4], c[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          x[u] = xs[g + u];
          c[u] = row[g + u];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
          acc = fmaf(c[u].x, x[u].x, acc);
          acc = fmaf(c[u].y, x[u].y, acc);
          acc = fmaf(c[u].z, x[u].z, acc);
          acc = fmaf(c[u].w, x[u].w, acc);
        }
      }
      float r = rintf(acc);
      r = fminf(fmaxf(r, -32768.0f), 32767.0f);
      if (m < st.n_out) out_tile[p + L * j] = (int16_t)r;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < tile_outputs; i += kThreads)
    if (tile_base + i < st.n_out) out[st.out_off + tile_base + i] = out_tile[i];
}

// ---- row layout, four consecutive outputs per lane ------------------------------------------------------------------
// The kernel above reads two 16-byte LDS operands per four FMAs and runs one serial FMA chain per lane: it is bound by
// the LDS pipeline and by the latency of that chain.  Here a lane computes FOUR CONSECUTIVE outputs 4 u + q (q = 0..3)
// of its row.  Their windows start ~M/L samples apart (4.35 at 48 kHz), so one aligned 16-byte read of the row serves
// all four: output q multiplies it by its own coefficient row, shifted by the whole groups d_q and the 0..3 samples a_q
// its first tap lies past the first group of the common window (zero coefficients in front and behind leave an
// accumulator unchanged, so the result is bit-identical).  The 16 lanes of a DPP row work on the same quad u of 16
// different rows (outputs L apart: same four phases), so the coefficients need no LDS at all: lane i of the row loads
// group 16 b + i of each of the four coefficient rows from the table, and step 16 b + i of every lane takes them from
// lane i as the DPP row broadcast operand of v_fmac_f32.  Per 16 FMAs: one LDS read instead of eight, four independent
// chains, and no per-wave copy of coefficient rows.
constexpr int kQuadRows = 16;   // rows (outputs L apart) per tile = lanes of a DPP row

#define NEEDLE_FMAC_BCAST(N)                                                                                      \
  template <>                                                                                                    \
  __device__ __forceinline__ void fmac_bcast<N>(float &acc, float c, float x) {                                  \
    asm("v_fmac_f32_dpp %0, %1, %2 row_newbcast:" #N " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(c), "v"(x)); \
  }
// acc += (c of lane N of this 16-lane row) * x
template <int N>
__device__ __forceinline__ void fmac_bcast(float &acc, float c, float x);
NEEDLE_FMAC_BCAST(0) NEEDLE_FMAC_BCAST(1) NEEDLE_FMAC_BCAST(2) NEEDLE_FMAC_BCAST(3) NEEDLE_FMAC_BCAST(4)
NEEDLE_FMAC_BCAST(5) NEEDLE_FMAC_BCAST(6) NEEDLE_FMAC_BCAST(7) NEEDLE_FMAC_BCAST(8) NEEDLE_FMAC_BCAST(9)
NEEDLE_FMAC_BCAST(10) NEEDLE_FMAC_BCAST(11) NEEDLE_FMAC_BCAST(12) NEEDLE_FMAC_BCAST(13) NEEDLE_FMAC_BCAST(14)
NEEDLE_FMAC_BCAST(15)
#undef NEEDLE_FMAC_BCAST
// The compiler does not see the DPP reads inside an asm statement, so it inserts no wait states between whatever last
// wrote the coefficient registers (or EXEC) and the first of them: one s_nop 4 in front of every block of steps,
// tied to the registers so that it cannot be moved away from them.
__device__ __forceinline__ void dpp_wait_states(float4 (&c)[4]) {
  asm volatile("s_nop 4"
               : "+v"(c[0].x), "+v"(c[0].y), "+v"(c[0].z), "+v"(c[0].w), "+v"(c[1].x), "+v"(c[1].y), "+v"(c[1].z), "+v"(c[1].w),
                 "+v"(c[2].x), "+v"(c[2].y), "+v"(c[2].z), "+v"(c[2].w), "+v"(c[3].x), "+v"(c[3].y), "+v"(c[3].z), "+v"(c[3].w));
}
template <int N, bool PLAIN = false>
__device__ __forceinline__ void quad_step(const float4 (&c)[4], float4 x, float (&acc)[4]) {
#pragma unroll
  for (int q = 0; q < 4; q++) {
    if (PLAIN) {
      acc[q] = fmaf(c[q].x, x.x, acc[q]); acc[q] = fmaf(c[q].y, x.y, acc[q]);
      acc[q] = fmaf(c[q].z, x.z, acc[q]); acc[q] = fmaf(c[q].w, x.w, acc[q]);
    } else {
      fmac_bcast<N>(acc[q], c[q].x, x.x);
      fmac_bcast<N>(acc[q], c[q].y, x.y);
      fmac_bcast<N>(acc[q], c[q].z, x.z);
      fmac_bcast<N>(acc[q], c[q].w, x.w);
    }
  }
}

// Measured on the first version of this kernel (profiles/r02_resample_lab.log): staging alone 0.19 ms, arithmetic
// alone 0.45 ms, together 0.75 ms -- they ADD, although two workgroups share a CU, because the staging was bound by
// its own integer arithmetic (64-bit index clamps and divisions per group) on the same SIMDs, not by HBM.  Neither
// persistent workgroups with a start skew (0.92 ms) nor prefetching the next tile into registers (128 VGPRs, one
// workgroup per CU: 1.26 ms) helped.  Hence: tiles that lie wholly inside their stream take a staging path with no
// clamps and 32-bit offsets from a uniform base, and what a quad needs (first group, the four coefficient rows and
// their shifts) comes from a table built on the host instead of four integer divisions per lane.
// LAB (timing experiments, results wrong; instantiated only in a library built with -DNEEDLE_HIP_LAB_BUILD, which
// tools/rslab.sh builds beside the product, and selected there through NEEDLE_HIP_RESAMPLE_LAB): 1 no staging, 2 no FMA
// loop, 4 plain v_fmac_f32 instead of the DPP form
constexpr int kQuadGroupsInFlight = 6;  // 16-byte loads a thread keeps in flight while staging by groups
constexpr int kQuadMaxRounds = 2;       // rounds of quads per tile (L <= 4 * 64 * 2 = 512 at 1024 threads)

struct QuadInfo {     // per quad u of a row (outputs 4 u .. 4 u + 3), the same for every row and tile
  uint32_t b0;        // first 16-byte group of the common window, from the start of the row
  uint32_t coef[4];   // group offset into the padded coefficient table of output q's row, shift and delay applied
};

// four down-mixed samples of one aligned group as it lies in memory
template <int CH>
__device__ __forceinline__ float4 group_to_f32(typename std::conditional<CH == 1, int2, int4>::type v) {
  int s0, s1, s2, s3;
  if constexpr (CH == 1) {
    s0 = (int16_t)v.x; s1 = v.x >> 16; s2 = (int16_t)v.y; s3 = v.y >> 16;
  } else {  // integer down-mix (L + R) / 2, C truncation
    s0 = ((int)(int16_t)v.x + (v.x >> 16)) / 2; s1 = ((int)(int16_t)v.y + (v.y >> 16)) / 2;
    s2 = ((int)(int16_t)v.z + (v.z >> 16)) / 2; s3 = ((int)(int16_t)v.w + (v.w >> 16)) / 2;
  }
  return float4{(float)s0, (float)s1, (float)s2, (float)s3};
}

// Staging of a tile's sixteen rows for the kernels with four outputs per lane: row r = `groups` aligned groups of four
// down-mixed samples from input index first0 + r M on, as f32, at lds4 + pitch r; zeros outside the stream.  blockDim.x /
// 16 threads per row.
template <int CH, bool VEC4>
__device__ __forceinline__ void stage_quad_rows(float4 *lds4, const int16_t *src, uint64_t n_in, long long first0, int M,
                                                int pitch, int groups) {
  using raw_t = typename std::conditional<CH == 1, int2, int4>::type;  // four samples as they lie in memory
  float *stage = reinterpret_cast<float *>(lds4);
  const int tpr = blockDim.x / kQuadRows, row = threadIdx.x / tpr, me = threadIdx.x % tpr;
  const int count = 4 * groups;
  auto sample = [&](long long idx) -> int {  // one down-mixed input sample, 0 outside the stream (branch-free load)
    const bool ok = idx >= 0 && (uint64_t)idx < n_in;
    const long long at = idx < 0 ? 0 : ((uint64_t)idx < n_in ? idx : (long long)n_in - 1);
    int sv;
    if (CH == 1) {
      sv = src[at];
    } else {
      const int v = reinterpret_cast<const int *>(src)[at];
      sv = ((int)(int16_t)v + (v >> 16)) / 2;  // integer down-mix, C truncation
    }
    return ok ? sv : 0;
  };
  const bool by_groups = VEC4 && (reinterpret_cast<uintptr_t>(src) & 15) == 0 && n_in >= 4;
  const long long tile_last = first0 + (long long)(kQuadRows - 1) * M + 4 * (long long)groups;  // one past the last sample staged
  if (by_groups && first0 >= 0 && tile_last <= (long long)n_in) {
    // the tile lies inside the stream (all but the first and last tiles of a stream): a uniform base, 32-bit
    // offsets, no clamps
    const raw_t *base = reinterpret_cast<const raw_t *>(src + (size_t)CH * first0);  // first0 is a multiple of 4
    const uint32_t row_groups = (uint32_t)(row * M) >> 2;                             // M is a multiple of 4 (VEC4)
    float4 *dst = lds4 + pitch * row;
    for (int o0 = me; o0 < groups; o0 += tpr * kQuadGroupsInFlight) {
      raw_t v[kQuadGroupsInFlight];
#pragma unroll
      for (int u = 0; u < kQuadGroupsInFlight; u++) v[u] = base[row_groups + (uint32_t)min(o0 + u * tpr, groups - 1)];
#pragma unroll
      for (int u = 0; u < kQuadGroupsInFlight; u++)
        if (o0 + u * tpr < groups) dst[o0 + u * tpr] = group_to_f32<CH>(v[u]);
    }
  } else if (by_groups) {  // a tile that sticks out of its stream: clamped loads, then the groups outside sample by sample
    float4 *dst = lds4 + pitch * row;
    const long long from = first0 + (long long)row * M;
    const long long last_group = ((long long)n_in - 4) & ~3ll;  // last aligned group wholly inside the stream
    for (int o = me; o < groups; o += tpr) {
      const long long idx = from + 4 * (long long)o;
      if (idx < 0 || idx > last_group)
        dst[o] = float4{(float)sample(idx), (float)sample(idx + 1), (float)sample(idx + 2), (float)sample(idx + 3)};
      else
        dst[o] = group_to_f32<CH>(*reinterpret_cast<const raw_t *>(src + (size_t)CH * idx));
    }
  } else {  // unaligned stream or M not a multiple of 4: sample by sample, eight loads in flight
    float *dst = stage + 4 * pitch * row;
    const long long from = first0 + (long long)row * M;
    constexpr int kU = 8;
    int o = me;
    for (; o + (kU - 1) * tpr < count; o += tpr * kU) {
      int v[kU];
#pragma unroll
      for (int u = 0; u < kU; u++) v[u] = sample(from + o + u * tpr);
#pragma unroll
      for (int u = 0; u < kU; u++) dst[o + u * tpr] = (float)v[u];
    }
    for (; o < count; o += tpr) dst[o] = (float)sample(from + o);
  }
}

// SMALL: at most 640 threads (L <= 160), compiled for at least five waves per SIMD = two workgroups per CU (forcing 64 VGPRs for three costs spills: 0.76 ms against 0.68).
template <int CH, bool VEC4, bool SMALL, int LAB = 0>
__global__ __launch_bounds__(SMALL ? 640 : 1024, SMALL ? 5 : 4) void resample_quad_kernel(
    const int16_t *__restrict__ in, const RsStream *__restrict__ streams, int num_streams,
    const float4 *__restrict__ coefq, const QuadInfo *__restrict__ quad_info, RsGeom geo, int steps, int splits,
    uint32_t skew_blocks, int skew_unit, int16_t *__restrict__ out) {
  extern __shared__ float4 lds4[];  // kQuadRows rows of samples | the tile's outputs (s16)
  // Every tile takes the same time, so the workgroups that start together (the first ones of a launch, two per CU)
  // would stage together -- HBM saturated, the SIMDs idle -- and then compute together, HBM idle, and every later
  // workgroup inherits that rhythm from the slot it takes over.  The first `skew_blocks` workgroups therefore start
  // 0..3 units late (a quarter of a tile's time each, chosen by a hash of the block index): at any moment some
  // workgroups of the device are staging and some computing.
  if (blockIdx.x < skew_blocks && skew_unit > 0) {
    const uint32_t late = (blockIdx.x * 2654435761u) >> 30;
    for (uint32_t i = 0; i < late * (uint32_t)skew_unit; i++) __builtin_amdgcn_s_sleep(127);
  }
  const int L = geo.L, M = geo.M;
  int lo = 0, hi = num_streams - 1;
  while (lo < hi) {
    int mid = (lo + hi + 1) >> 1;
    if (streams[mid].block_base <= blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const RsStream st = streams[lo];
  const int tile_outputs = kQuadRows * L;
  // A tile (16 rows x L outputs) is cut into `splits` workgroups along its rows: quads [q0, q1) of every row.  A
  // workgroup then stages a fraction of each row (plus the window overlap) -- smaller workgroups, more of them per CU,
  // so that some stage while others compute.
  const int quads = (L + 3) >> 2, quads_per_split = (quads + splits - 1) / splits;
  const uint32_t in_stream = blockIdx.x - st.block_base;
  const uint64_t tile = in_stream / (uint32_t)splits;
  const int q0 = (int)(in_stream % (uint32_t)splits) * quads_per_split, q1 = min(quads, q0 + quads_per_split);
  if (q0 >= q1) return;
  const uint64_t tile_base = tile * (uint64_t)tile_outputs;
  const int half = geo.T / 2;
  const int g0 = (int)quad_info[q0].b0;  // first group of a row this workgroup reads
  // input index of region[0]: the first tap of the tile's first output, moved down by delta to a multiple of 4, plus
  // the groups of a row in front of this workgroup's share
  const long long first0 = (long long)(tile * (uint64_t)kQuadRows * (uint64_t)M) - half + 1 - geo.delta + 4ll * g0;
  const int16_t *src = in + st.in_off;
  const int groups = (int)quad_info[q1 - 1].b0 - g0 + steps + 2;  // groups of a row that some lane may read
  // the tile's outputs are collected in LDS for a coalesced copy to HBM, over the START of the sample region once every
  // lane has finished reading it (a barrier more, 4.7 KB of LDS less: three workgroups per CU at 48 kHz instead of two)
  int16_t *out_tile = reinterpret_cast<int16_t *>(lds4);

  if (!(LAB & 1)) stage_quad_rows<CH, VEC4>(lds4, src, st.n_in, first0, M, geo.pitch, groups);
  __syncthreads();

  // ---- quads: DPP row dr of the workgroup takes quad u = round * rows + dr; lane j of the row takes row j ----------
  const int dpp_rows = blockDim.x >> 4;
  const int j = threadIdx.x & 15, dr = threadIdx.x >> 4;
  const int nblk = (LAB & 2) ? 0 : (steps + 15) >> 4;
  const bool last_half = (steps & 15) != 0;  // steps is a multiple of 8: the last block may have 8 steps only
  int16_t res[kQuadMaxRounds][4] = {};
  for (int u0 = q0; u0 < q1; u0 += dpp_rows) {
    const int u = u0 + dr;
    const bool live = u < q1;              // uniform per DPP row, not per wave: dead rows compute and store nothing
    const QuadInfo qi = quad_info[live ? u : q1 - 1];
    const float4 *cptr[4];
#pragma unroll
    for (int q = 0; q < 4; q++) cptr[q] = coefq + qi.coef[q] + j;
    const float4 *xs = lds4 + j * geo.pitch + ((int)qi.b0 - g0);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    float4 c[4];
#pragma unroll
    for (int q = 0; q < 4; q++) c[q] = cptr[q][0];
    for (int b = 0; b < nblk; b++) {
      float4 cn[4];
      const int nb = b + 1 < nblk ? 16 * (b + 1) : 16 * b;  // (the last block re-reads its own groups: no branch)
#pragma unroll
      for (int q = 0; q < 4; q++) cn[q] = cptr[q][nb];
      const float4 *x = xs + 16 * b;
      dpp_wait_states(c);
      quad_step<0, (LAB & 4) != 0>(c, x[0], acc);
      quad_step<1, (LAB & 4) != 0>(c, x[1], acc);
      quad_step<2, (LAB & 4) != 0>(c, x[2], acc);
      quad_step<3, (LAB & 4) != 0>(c, x[3], acc);
      quad_step<4, (LAB & 4) != 0>(c, x[4], acc);
      quad_step<5, (LAB & 4) != 0>(c, x[5], acc);
      quad_step<6, (LAB & 4) != 0>(c, x[6], acc);
      quad_step<7, (LAB & 4) != 0>(c, x[7], acc);
      if (!(last_half && b + 1 == nblk)) {
        quad_step<8, (LAB & 4) != 0>(c, x[8], acc);
        quad_step<9, (LAB & 4) != 0>(c, x[9], acc);
        quad_step<10, (LAB & 4) != 0>(c, x[10], acc);
        quad_step<11, (LAB & 4) != 0>(c, x[11], acc);
        quad_step<12, (LAB & 4) != 0>(c, x[12], acc);
        quad_step<13, (LAB & 4) != 0>(c, x[13], acc);
        quad_step<14, (LAB & 4) != 0>(c, x[14], acc);
        quad_step<15, (LAB & 4) != 0>(c, x[15], acc);
      }
#pragma unroll
      for (int q = 0; q < 4; q++) c[q] = cn[q];
    }
    // results of this round, kept in registers until every lane is done with the sample region (rounds <= kMaxRounds)
    const int rd = (u0 - q0) / dpp_rows;
#pragma unroll
    for (int k = 0; k < kQuadMaxRounds; k++)
      if (k == rd) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
          float r = rintf(acc[q]);
          res[k][q] = (int16_t)fminf(fmaxf(r, -32768.0f), 32767.0f);
        }
      }
  }
  __syncthreads();  // the sample region is dead: its start becomes the output tile, [row][output - 4 q0]
  const int o0 = 4 * q0, seg = min(L, 4 * q1) - o0;  // this workgroup's outputs of a row
#pragma unroll
  for (int k = 0; k < kQuadMaxRounds; k++) {
    const int u = q0 + k * dpp_rows + dr;
    if (u < q1) {
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int o = 4 * u + q;
        if (o < L) out_tile[j * seg + (o - o0)] = res[k][q];
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < kQuadRows * seg; i += blockDim.x) {
    const int r = i / seg, o = o0 + i - r * seg;
    const uint64_t m = tile_base + (uint64_t)o + (uint64_t)L * r;
    if (m < st.n_out) out[st.out_off + m] = out_tile[i];
  }
}

// ---- integer decimation (L = 1: 44.1 kHz -> 11.025 kHz and its relatives) -----------------------------------------------
// Every output uses the SAME coefficient row, so the coefficients are wave-uniform: they sit in SGPRs (scalar loads
// through the constant cache) and the FMAs are plain v_fmac_f32 with a scalar operand -- no LDS, no DPP, 2.8 cycles
// each instead of 4.3.  A lane computes Q consecutive outputs, whose windows start M samples apart: one aligned
// 16-byte LDS read of the contiguous region serves all Q, each output taking from it the taps that fall into it (the
// loop over the window is fully unrolled, so which coefficient meets which sample is known at compile time and no FMA is
// spent on padding).  Q M / 4 (the distance between two lanes' windows in 16-byte slots) is odd, which makes the reads of
// a lane group conflict-free: Q = 5 for M = 4, Q = 6 for M = 2.  Per output: T FMAs and T / (4 Q) + 1 LDS reads.
template <int CH, int M, int Q, int THREADS>
__global__ __launch_bounds__(THREADS) void resample_dec_kernel(const int16_t *__restrict__ in,
                                                               const RsStream *__restrict__ streams, int num_streams,
                                                               const float *__restrict__ coef,  // [T] taps of the one phase
                                                               int16_t *__restrict__ out) {
  constexpr int T = 32 * M, half = T / 2;            // design_filter: 2 ceil(16 M) taps
  constexpr int kDelta = (((1 - half) % 4) + 4) % 4;  // region[0] = first tap of the tile minus kDelta: a multiple of 4
  constexpr int kTile = THREADS * Q;                  // outputs per tile
  constexpr int kStride = Q * M / 4;                  // 16-byte slots between the windows of two lanes
  static_assert((Q * M) % 4 == 0 && (kStride & 1) == 1, "lanes must be an odd number of 16-byte slots apart");
  constexpr int kSteps = ((Q - 1) * M + kDelta + T + 3) / 4;  // slots a lane reads
  constexpr int kRegion = kStride * (THREADS - 1) + kSteps;   // slots of the tile's region
  using raw_t = typename std::conditional<CH == 1, int2, int4>::type;
  __shared__ float4 region[kRegion];
  __shared__ int16_t out_tile[kTile];
  int lo = 0, hi = num_streams - 1;
  while (lo < hi) {
    int mid = (lo + hi + 1) >> 1;
    if (streams[mid].block_base <= blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const RsStream st = streams[lo];
  const uint64_t tile = blockIdx.x - st.block_base;
  const uint64_t tile_base = tile * (uint64_t)kTile;
  const long long first0 = (long long)(tile_base * (uint64_t)M) - half + 1 - kDelta;  // input index of region[0]
  const int16_t *src = in + st.in_off;
  auto sample = [&](long long idx) -> int {  // one down-mixed input sample, 0 outside the stream
    const bool ok = idx >= 0 && (uint64_t)idx < st.n_in;
    const long long at = idx < 0 ? 0 : ((uint64_t)idx < st.n_in ? idx : (long long)st.n_in - 1);
    int sv;
    if (CH == 1) {
      sv = src[at];
    } else {
      const int v = reinterpret_cast<const int *>(src)[at];
      sv = ((int)(int16_t)v + (v >> 16)) / 2;
    }
    return ok ? sv : 0;
  };
  // ---- staging by aligned groups of four samples (first0 is a multiple of 4) --------------------------------------
  const bool aligned = (reinterpret_cast<uintptr_t>(src) & 15) == 0 && st.n_in >= 4;
  if (aligned && first0 >= 0 && first0 + 4ll * kRegion <= (long long)st.n_in) {  // inside the stream: no clamps
    const raw_t *base = reinterpret_cast<const raw_t *>(src + (size_t)CH * first0);
    constexpr int kInFlight = 6;
    for (int o0 = threadIdx.x; o0 < kRegion; o0 += THREADS * kInFlight) {
      raw_t v[kInFlight];
#pragma unroll
      for (int u = 0; u < kInFlight; u++) v[u] = base[min(o0 + u * THREADS, kRegion - 1)];
#pragma unroll
      for (int u = 0; u < kInFlight; u++)
        if (o0 + u * THREADS < kRegion) region[o0 + u * THREADS] = group_to_f32<CH>(v[u]);
    }
  } else {  // first / last tile of a stream, or an unaligned stream: sample by sample, zeros outside
    for (int o = threadIdx.x; o < kRegion; o += THREADS) {
      const long long idx = first0 + 4ll * o;
      region[o] = float4{(float)sample(idx), (float)sample(idx + 1), (float)sample(idx + 2), (float)sample(idx + 3)};
    }
  }
  __syncthreads();
  // ---- Q outputs per lane; tap k of output q meets sample q M + kDelta + k of the lane's window -------------------
  const float4 *xs = region + kStride * threadIdx.x;
  float acc[Q];
#pragma unroll
  for (int q = 0; q < Q; q++) acc[q] = 0.f;
  typedef float v4f __attribute__((ext_vector_type(4), aligned(16)));  // one 128-bit LDS read per slot
#pragma unroll
  for (int s = 0; s < kSteps; s++) {
    const v4f x = *reinterpret_cast<const v4f *>(xs + s);
    const float xv[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
    for (int q = 0; q < Q; q++) {
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int k = 4 * s + i - q * M - kDelta;  // compile-time
        if (k >= 0 && k < T) acc[q] = fmaf(coef[k], xv[i], acc[q]);
      }
    }
  }
#pragma unroll
  for (int q = 0; q < Q; q++) {
    float r = rintf(acc[q]);
    r = fminf(fmaxf(r, -32768.0f), 32767.0f);
    out_tile[Q * threadIdx.x + q] = (int16_t)r;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < kTile; i += THREADS)
    if (tile_base + i < st.n_out) out[st.out_off + tile_base + i] = out_tile[i];
}

#include "resample_mfma.h"

}  // namespace

size_t resample_out_len(size_t n_in, int rate) {
  if (rate == kTarget) return n_in;
  const int g = gcd_i(kTarget, rate);
  const unsigned long long L = kTarget / g, M = rate / g;
  return (size_t)(((unsigned long long)n_in * L + M - 1) / M);
}

// streams: (in_off in s16 values, n_in samples per channel, out_off samples); d_out receives mono s16 @ 11025 Hz
Status gpu_resample_device(const int16_t *d_in, const std::vector<ResampleSpan> &spans, int channels, int rate,
                           int16_t *d_out, bool sync) {
  std::lock_guard<std::recursive_mutex> gpu_lock(gpu_mutex());
  if (channels != 1 && channels != 2) return Status::Make(NeedleError_InvalidArgument, "resample: channels must be 1 or 2");
  if (rate < 2000 || rate > 768000) return Status::Make(NeedleError_InvalidArgument, "resample: unsupported sample rate");
  if (channels == 2) {  // the kernel reads a stereo sample as one aligned 32-bit word
    if (reinterpret_cast<uintptr_t>(d_in) & 3) return Status::Make(NeedleError_InvalidArgument, "resample: stereo PCM must be 4-byte aligned");
    for (const ResampleSpan &sp : spans)
      if (sp.in_off & 1) return Status::Make(NeedleError_InvalidArgument, "resample: stereo streams must start on an even value offset");
  }
  Status s = ensure_device();
  if (!s.ok()) return s;
  int dev = 0;
  NEEDLE_HIP_TRY(hipGetDevice(&dev));
  Design *d;
  {
    std::lock_guard<std::mutex> lock(g_mu);
    d = &g_designs[{dev, rate}];
    if (d->T == 0) {
      design_filter(rate, d);
      d->G = (((d->T + 3) / 4 + 1) + 3) & ~3;  // 16-byte groups per shifted row, a multiple of 4 for the kernel's unroll
      // lanes per phase: 16 (one LDS lane group) when a tile then has plenty of outputs (L >= 256), more for small L;
      // fewer when M is so large that 16 inputs M apart do not fit the LDS region
      int n = 16;
      while (n * d->L < 4096) n *= 2;  // small L: tiles of >= 4096 outputs amortise the per-tile latencies
      auto footprint = [&](int lanes) {  // samples of LDS the tile's inputs take (row layout repeats the overlaps)
        const bool rows = d->M >= kRowModeMinM && lanes <= kRowModeMaxN;
        return rows ? (long long)lanes * (d->M + 4 * d->G + 8) : (long long)lanes * d->M + 4 * d->G + 8;
      };
      while (n > 1 && footprint(n) > kMaxRegionSamples) n /= 2;
      if (footprint(n) > kMaxRegionSamples)
        return Status::Make(NeedleError_InvalidArgument, "resample: rate ratio too large for this kernel");
      d->n = n;
      std::vector<float> shifted((size_t)4 * d->L * 4 * d->G, 0.f);
      for (int a = 0; a < 4; a++)
        for (int p = 0; p < d->L; p++)
          for (int k = 0; k < d->T; k++)
            shifted[((size_t)a * d->L + p) * 4 * d->G + k + a] = d->coef[(size_t)p * d->T + k];
      NEEDLE_HIP_TRY(hipMalloc((void **)&d->d_coef, shifted.size() * sizeof(float)));
      NEEDLE_HIP_TRY(hipMemcpy(d->d_coef, shifted.data(), shifted.size() * sizeof(float), hipMemcpyHostToDevice));
      NEEDLE_HIP_TRY(hipMalloc((void **)&d->d_coef_plain, d->coef.size() * sizeof(float)));
      NEEDLE_HIP_TRY(hipMemcpy(d->d_coef_plain, d->coef.data(), d->coef.size() * sizeof(float), hipMemcpyHostToDevice));
      // resample_quad_kernel: the first taps of a lane's four outputs lie up to span = ceil(3 M / L) samples apart, 3
      // more for the alignment of the first and 3 for a row's shift
      const int span = (3 * d->M + d->L - 1) / d->L;
      d->steps = (((d->T + span + 6 + 3) / 4) + 7) & ~7;
      d->quad_pad = (span + 3) / 4 + 1;
      d->row_len = d->quad_pad + std::max(d->G, d->steps) + 16;
      std::vector<float> padded((size_t)4 * d->L * 4 * d->row_len, 0.f);
      for (int a = 0; a < 4; a++)
        for (int p = 0; p < d->L; p++)
          for (int k = 0; k < d->T; k++)
            padded[((size_t)a * d->L + p) * 4 * d->row_len + 4 * d->quad_pad + k + a] = d->coef[(size_t)p * d->T + k];
      NEEDLE_HIP_TRY(hipMalloc((void **)&d->d_coefq, padded.size() * sizeof(float)));
      NEEDLE_HIP_TRY(hipMemcpy(d->d_coefq, padded.data(), padded.size() * sizeof(float), hipMemcpyHostToDevice));
      // per quad: where its common window starts and which (shifted, delayed) coefficient row each output reads
      d->delta = d->M % 4 == 0 ? ((1 - d->T / 2) % 4 + 4) % 4 : 0;
      const int quads = (d->L + 3) / 4;
      std::vector<QuadInfo> info((size_t)quads);
      for (int u = 0; u < quads; u++) {
        int b0 = 0;
        for (int q = 0; q < 4; q++) {
          const int o = std::min(4 * u + q, d->L - 1);
          const long long pm = (long long)o * d->M;
          const int cp = (int)(pm / d->L), phase = (int)(pm - (long long)cp * d->L);
          const int rel = cp + d->delta;
          if (q == 0) b0 = rel >> 2;
          const int off = rel - 4 * b0, dq = off >> 2, a = off & 3;
          info[u].coef[q] = (uint32_t)(((size_t)a * d->L + phase) * d->row_len + (size_t)(d->quad_pad - dq));
        }
        info[u].b0 = (uint32_t)b0;
      }
      NEEDLE_HIP_TRY(hipMalloc(&d->d_quad_info, info.size() * sizeof(QuadInfo)));
      NEEDLE_HIP_TRY(hipMemcpy(d->d_quad_info, info.data(), info.size() * sizeof(QuadInfo), hipMemcpyHostToDevice));
      // resample_mfma_kernel: per block of sixteen outputs of a row, where the union of their windows starts and the
      // coefficient of every (sample of the union, output) pair as the MFMA's B operand
      if (d->M % 4 == 0 && d->M >= kRowModeMinM && d->L >= 16) {
        auto cp = [&](int o) { return (int)((long long)o * d->M / d->L); };
        d->nblocks = (d->L + 15) / 16;
        int need = 0;
        for (int b = 0; b < d->nblocks; b++) need = std::max(need, cp(std::min(16 * b + 15, d->L - 1)) - cp(16 * b) + d->T);
        for (int bucket : {12, 20, 36, 52})
          if (d->mfma_steps == 0 && 4 * bucket >= need) d->mfma_steps = bucket;
      }
      if (d->mfma_steps) {
        const int S = d->mfma_steps;
        std::vector<float> cb((size_t)d->nblocks * S * 64, 0.f);
        d->block_k0.assign((size_t)2 * d->nblocks, 0);  // window starts, then the steps a block needs (a multiple of 4)
        for (int b = 0; b < d->nblocks; b++) {
          const int c0 = (int)((long long)16 * b * d->M / d->L);
          d->block_k0[b] = c0 + d->delta;
          const int c_last = (int)((long long)std::min(16 * b + 15, d->L - 1) * d->M / d->L);
          d->block_k0[d->nblocks + b] = std::min(S, ((c_last - c0 + d->T + 3) / 4 + 3) & ~3);
          for (int j = 0; j < 16; j++) {
            const int o = 16 * b + j;
            if (o >= d->L) continue;
            const long long pm = (long long)o * d->M;
            const int cpo = (int)(pm / d->L), phase = (int)(pm - (long long)cpo * d->L);
            for (int t = 0; t < d->T; t++) {
              const int k = cpo - c0 + t;  // sample of the union this tap meets
              cb[((size_t)b * S + (size_t)(k >> 2)) * 64 + (size_t)(16 * (k & 3) + j)] = d->coef[(size_t)phase * d->T + t];
            }
          }
        }
        NEEDLE_HIP_TRY(hipMalloc((void **)&d->d_coef_b, cb.size() * sizeof(float)));
        NEEDLE_HIP_TRY(hipMemcpy(d->d_coef_b, cb.data(), cb.size() * sizeof(float), hipMemcpyHostToDevice));
        NEEDLE_HIP_TRY(hipMalloc((void **)&d->d_block_k0, d->block_k0.size() * sizeof(int)));
        NEEDLE_HIP_TRY(hipMemcpy(d->d_block_k0, d->block_k0.data(), d->block_k0.size() * sizeof(int), hipMemcpyHostToDevice));
      }
    }
  }
  // Decimation steps of 64 samples or more use the row layout; there the kernel with four consecutive outputs per
  // lane applies (NEEDLE_HIP_RESAMPLE_V1 forces the first kernel: tests and A/B timing).
  // a tile's rows are cut into `quad_splits` workgroups (see the kernel); the rows of a workgroup: the groups between
  // the first reads of its first and last quad, the window, two groups of slack
  // (measured at 48 kHz, 37 quads: 1 / 2 / 3 / 4 / 6 workgroups per tile 0.77 / 0.72 / 0.64 / 0.72 / 0.81 ms: about
  // 16 quads = 256 threads per workgroup)
  const int quads_all = (d->L + 3) / 4;
  int quad_splits = (quads_all + 15) / 16;
  if (const char *e = getenv("NEEDLE_HIP_RESAMPLE_SPLITS")) quad_splits = std::max(1, atoi(e));  // tuning
  quad_splits = std::min(quad_splits, quads_all);
  const int quads_per_split = (quads_all + quad_splits - 1) / quad_splits;
  const int row_groups = ((quads_per_split - 1) * 4 * d->M + d->L - 1) / d->L / 4 + 2 + d->steps + 2;
  const int quad_pitch = row_groups | 1;
  const size_t quad_lds = (size_t)kQuadRows * quad_pitch * 16;  // (the output tile reuses the start of the region)
  const int quad_threads = std::min(1024, ((16 * quads_per_split + 63) / 64) * 64);
  const bool quad = d->M >= kRowModeMinM && d->L >= 4 && quad_lds <= 160 * 1024 &&
                    (size_t)kQuadRows * 4 * quads_per_split * 2 <= quad_lds &&
                    quads_per_split <= kQuadMaxRounds * (quad_threads / 16) && getenv("NEEDLE_HIP_RESAMPLE_V1") == nullptr;
  // integer decimation by 4 or 2 (44.1 / 22.05 kHz): the kernel with scalar coefficients
  constexpr int kDecThreads = 256;
  const int dec_q = d->L == 1 && d->T == 32 * d->M ? (d->M == 4 ? 5 : d->M == 2 ? 6 : 0) : 0;
  const bool dec = dec_q != 0 && getenv("NEEDLE_HIP_RESAMPLE_V1") == nullptr;
  // matrix-core kernel (resample_mfma.h): one wave per block of sixteen outputs of a row, at most `mf_waves` per
  // workgroup; a tile's blocks are cut into `mf_splits` workgroups.  NEEDLE_HIP_RESAMPLE_QUAD forces the DPP kernel.
  const int mf_max_waves = mfma_rs::kConsumers;
  const int mf_splits = d->mfma_steps ? (d->nblocks + mf_max_waves - 1) / mf_max_waves : 1;
  const int mf_waves = d->mfma_steps ? (d->nblocks + mf_splits - 1) / mf_splits : 1;
  int mf_groups = 0;
  for (int sp = 0; d->mfma_steps && sp < mf_splits; sp++) {
    const int b0 = sp * mf_waves, b1 = std::min(b0 + mf_waves, d->nblocks);
    if (b0 >= b1) { mf_groups = 1 << 30; break; }  // an empty split: not this kernel
    const int start = d->block_k0[b0] & ~3;
    mf_groups = std::max(mf_groups, (d->block_k0[b1 - 1] + 4 * d->mfma_steps - start + 3) >> 2);
  }
  // a row of the LDS image: the groups the staging threads of the row move, or all the groups its blocks read if more
  const int mf_row_groups = std::max(mfma_rs::kCovered, mf_groups);
  const int mf_buffer_floats = mf_row_groups * 4 * mfma_rs::kRows;
  const size_t mf_lds = (size_t)2 * mf_buffer_floats * sizeof(float);  // double-buffered
  const bool mfma = quad && d->mfma_steps != 0 && (mf_groups <= mfma_rs::kCovered ||  // longer rows: the rest is the start of the next row (one split only)
                     (mf_splits == 1 && mf_groups - d->M / 4 <= mfma_rs::kCovered && mfma_rs::kCovered >= d->M / 4 &&
                      mf_groups - mfma_rs::kCovered <= 64 * mfma_rs::kProducers)) &&
                    mf_lds <= 160 * 1024 && getenv("NEEDLE_HIP_RESAMPLE_QUAD") == nullptr;
  const uint64_t tile_outputs = dec ? (uint64_t)kDecThreads * dec_q : quad ? (uint64_t)kQuadRows * d->L : (uint64_t)d->n * d->L;
  std::vector<RsStream> meta;
  uint64_t blocks = 0;
  for (const ResampleSpan &sp : spans) {
    RsStream m;
    m.in_off = sp.in_off;
    m.n_in = sp.n_in;
    m.out_off = sp.out_off;
    m.n_out = resample_out_len(sp.n_in, rate);
    m.block_base = (uint32_t)blocks;
    m.pad = 0;
    blocks += (m.n_out + tile_outputs - 1) / tile_outputs * (dec ? 1 : mfma ? (uint64_t)mf_splits : quad ? (uint64_t)quad_splits : 1);
    if (m.n_out) meta.push_back(m);
  }
  if (blocks > 0x7FFFFFFFull) return Status::Make(NeedleError_InvalidArgument, "resample: batch too large for one launch");
  if (!meta.empty()) {
    hipStream_t stream = library_stream();
    static std::map<int, std::pair<DeviceBuffer<RsStream> *, PinnedStage *>> ws;
    auto &w = ws[dev];
    if (!w.first) {
      w.first = new DeviceBuffer<RsStream>();
      w.second = new PinnedStage();
    }
    if (!(s = w.first->reserve(meta.size())).ok()) return s;
    if (!(s = w.second->acquire(meta.size() * sizeof(RsStream))).ok()) return s;
    std::memcpy(w.second->ptr, meta.data(), meta.size() * sizeof(RsStream));
    NEEDLE_HIP_TRY(hipMemcpyAsync(w.first->ptr, w.second->ptr, meta.size() * sizeof(RsStream), hipMemcpyHostToDevice, stream));
    w.second->mark(stream);
    if (dec) {
      KernelTimer timer("resample");
      auto launch = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, dim3((uint32_t)blocks), dim3(kDecThreads), 0, stream, d_in, w.first->ptr, (int)meta.size(),
                           d->d_coef_plain, d_out);
      };
      if (d->M == 4) {
        if (channels == 1) launch(resample_dec_kernel<1, 4, 5, kDecThreads>); else launch(resample_dec_kernel<2, 4, 5, kDecThreads>);
      } else {
        if (channels == 1) launch(resample_dec_kernel<1, 2, 6, kDecThreads>); else launch(resample_dec_kernel<2, 2, 6, kDecThreads>);
      }
      NEEDLE_HIP_TRY(hipGetLastError());
      if (sync) NEEDLE_HIP_TRY(hipStreamSynchronize(library_stream()));
      return Status::Ok();
    }
    if (mfma) {
      mfma_rs::Geom mg;
      mg.L = d->L; mg.M = d->M; mg.half = d->T / 2; mg.delta = d->delta;
      mg.nblocks = d->nblocks; mg.blocks_per_wg = mf_waves; mg.splits = mf_splits;
      mg.m_groups = d->M / 4;
      mg.dup_lo = mg.dup_hi = 0;
      if (mf_groups > mfma_rs::kCovered) {
        mg.dup_lo = mfma_rs::kCovered - mg.m_groups;
        mg.dup_hi = mf_groups - mg.m_groups;
      }
      int cus = 256;
      (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
      // Wave roles.  The waves of a workgroup go to the four SIMDs in turn, so waves w, w + 4, w + 8, w + 12 share one.
      // Measured (profiles/r03_resample_mfma.log): whatever a staging wave issues -- conversions or bare loads --
      // adds its time to the MFMA time of the SIMD it sits on instead of hiding under it, and on a SIMD with two or
      // three multiplying waves it barely gets a turn (blocks first, staging waves last: 0.57 ms; the staging waves on
      // SIMDs with one multiplying wave: 0.44).  Hence: three SIMDs with three multiplying waves each and nothing else,
      // the fourth with the tenth block's wave and the three staging waves; waves 12-14 leave at once.
      // NEEDLE_HIP_RESAMPLE_LAYOUT=0 (tuning): blocks first, then the staging waves.
      static const unsigned char by_simd[16] = {0, 1, 2, 9, 3, 4, 5, 0x80, 6, 7, 8, 0x81, 0xFF, 0xFF, 0xFF, 0x82};
      const bool naive = getenv("NEEDLE_HIP_RESAMPLE_LAYOUT") && atoi(getenv("NEEDLE_HIP_RESAMPLE_LAYOUT")) == 0;
      for (int w = 0; w < 16; w++) {
        unsigned char r = naive ? (w < mf_waves ? (unsigned char)w : w < mf_waves + mfma_rs::kProducers ? (unsigned char)(0x80 + w - mf_waves) : 0xFF)
                                : by_simd[w];
        if (r < 0x80 && r >= mf_waves) r = 0xFF;  // fewer blocks than multiplying waves
        mg.role[w] = r;
      }
      const int mf_total_waves = 16;
      const int mf_threads = mf_total_waves * 64;
      int per_cu = 1;
      if (const char *e = getenv("NEEDLE_HIP_RESAMPLE_WG_PER_CU")) per_cu = std::max(1, atoi(e));  // tuning
      // persistent workgroups: a multiple of the splits, so that a workgroup keeps its blocks (and B registers)
      uint64_t grid = std::min<uint64_t>(blocks, (uint64_t)cus * per_cu);
      if (const char *e = getenv("NEEDLE_HIP_RESAMPLE_GRID")) grid = std::max(1, atoi(e));  // tests: few workgroups, many tiles each
      grid = std::max<uint64_t>(mf_splits, grid / mf_splits * mf_splits);
      auto launch = [&](auto kernel) -> Status {
        static std::map<std::pair<int, const void *>, size_t> announced;  // largest dynamic LDS size per device and kernel
        size_t &have = announced[{dev, reinterpret_cast<const void *>(kernel)}];
        if (have < mf_lds) {
          NEEDLE_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             (int)mf_lds));
          have = mf_lds;
        }
        KernelTimer timer("resample");
        hipLaunchKernelGGL(kernel, dim3((uint32_t)grid), dim3(mf_threads), mf_lds, stream, d_in, w.first->ptr,
                           (int)meta.size(), d->d_coef_b, d->d_block_k0, mg, (uint32_t)blocks, mf_buffer_floats, d_out);
        return Status::Ok();
      };
      using namespace mfma_rs;
      const bool mono = channels == 1;
#ifdef NEEDLE_HIP_LAB_BUILD
      const int mlab = getenv("NEEDLE_HIP_RESAMPLE_LAB") ? atoi(getenv("NEEDLE_HIP_RESAMPLE_LAB")) : 0;
#else
      const int mlab = 0;  // the product has no wrong-result variants
#endif
      if (mlab && !mono && d->mfma_steps == 52) {
#ifdef NEEDLE_HIP_LAB_BUILD
        switch (mlab) {
          case 1: s = launch(resample_mfma_kernel<2, 52, 1>); break;
          case 2: s = launch(resample_mfma_kernel<2, 52, 2>); break;
          case 3: s = launch(resample_mfma_kernel<2, 52, 3>); break;
          case 4: s = launch(resample_mfma_kernel<2, 52, 4>); break;
          case 7: s = launch(resample_mfma_kernel<2, 52, 7>); break;
          case 8: s = launch(resample_mfma_kernel<2, 52, 8>); break;
          case 12: s = launch(resample_mfma_kernel<2, 52, 12>); break;
          case 15: s = launch(resample_mfma_kernel<2, 52, 15>); break;
          case 33: s = launch(resample_mfma_kernel<2, 52, 33>); break;
          case 64: s = launch(resample_mfma_kernel<2, 52, 64>); break;
          case 128: s = launch(resample_mfma_kernel<2, 52, 128>); break;
          case 132: s = launch(resample_mfma_kernel<2, 52, 132>); break;
          case 68: s = launch(resample_mfma_kernel<2, 52, 68>); break;
          case 37: s = launch(resample_mfma_kernel<2, 52, 37>); break;
          case 6: s = launch(resample_mfma_kernel<2, 52, 6>); break;
          default: {
            unsigned long long zero[40] = {}, got[40];
            NEEDLE_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_rs_clock), zero, sizeof zero));
            s = launch(resample_mfma_kernel<2, 52, 16>);
            NEEDLE_HIP_TRY(hipStreamSynchronize(stream));
            NEEDLE_HIP_TRY(hipMemcpyFromSymbol(got, HIP_SYMBOL(g_rs_clock), sizeof got));
            const double n = (double)std::max<unsigned long long>(got[32], 1);
            fprintf(stderr, "resample_mfma block 0, s_memtime ticks per tile (work + barrier) over %llu tiles, by wave:", got[32]);
            for (int w = 0; w < 16; w++)
              if (mg.role[w] != 0xFF)
                fprintf(stderr, " %s%d %.0f+%.0f", mg.role[w] >= 0x80 ? "stage" : "blk", mg.role[w] & 0x7F, got[2 * w] / n, got[2 * w + 1] / n);
            fprintf(stderr, "\n  block 0: prologue %.1f us, loop %.1f us; last block entered %.1f us after block 0 and ended %.1f us after it\n",
                    (got[34] - got[33]) / 100.0, (got[35] - got[34]) / 100.0, ((double)got[36] - (double)got[33]) / 100.0,
                    ((double)got[37] - (double)got[35]) / 100.0);
            break;
          }
        }
#endif
      } else {
        switch (d->mfma_steps) {
          case 12: s = mono ? launch(resample_mfma_kernel<1, 12>) : launch(resample_mfma_kernel<2, 12>); break;
          case 20: s = mono ? launch(resample_mfma_kernel<1, 20>) : launch(resample_mfma_kernel<2, 20>); break;
          case 36: s = mono ? launch(resample_mfma_kernel<1, 36>) : launch(resample_mfma_kernel<2, 36>); break;
          default: s = mono ? launch(resample_mfma_kernel<1, 52>) : launch(resample_mfma_kernel<2, 52>); break;
        }
      }
      if (!s.ok()) return s;
      NEEDLE_HIP_TRY(hipGetLastError());
      if (sync) NEEDLE_HIP_TRY(hipStreamSynchronize(library_stream()));
      return Status::Ok();
    }
    RsGeom geo;
    geo.L = d->L; geo.M = d->M; geo.T = d->T; geo.G = d->G;
    if (quad) {
      geo.n_log2 = 4;
      geo.pitch = quad_pitch;
      geo.region_slots = kQuadRows * quad_pitch;
      const bool vec4 = d->M % 4 == 0;
      geo.delta = d->delta;
      const void *variants[] = {reinterpret_cast<const void *>(resample_quad_kernel<1, false, false>),
                                  reinterpret_cast<const void *>(resample_quad_kernel<1, true, false>),
                                  reinterpret_cast<const void *>(resample_quad_kernel<2, false, false>),
                                  reinterpret_cast<const void *>(resample_quad_kernel<2, true, false>),
                                  reinterpret_cast<const void *>(resample_quad_kernel<1, false, true>),
                                  reinterpret_cast<const void *>(resample_quad_kernel<1, true, true>),
                                  reinterpret_cast<const void *>(resample_quad_kernel<2, false, true>),
                                  reinterpret_cast<const void *>(resample_quad_kernel<2, true, true>),
#ifdef NEEDLE_HIP_LAB_BUILD
                                  reinterpret_cast<const void *>(resample_quad_kernel<2, true, true, 1>),
                                  reinterpret_cast<const void *>(resample_quad_kernel<2, true, true, 2>),
                                  reinterpret_cast<const void *>(resample_quad_kernel<2, true, true, 3>),
                                  reinterpret_cast<const void *>(resample_quad_kernel<2, true, true, 4>),
                                  reinterpret_cast<const void *>(resample_quad_kernel<2, true, true, 5>),
#endif
      };
      static std::map<int, size_t> quad_attr;  // largest dynamic LDS size announced per device
      if (quad_attr[dev] < quad_lds) {
        for (const void *fn : variants)
          NEEDLE_HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)quad_lds));
        quad_attr[dev] = quad_lds;
      }
      // one DPP row (16 lanes) per quad of a row's outputs, whole waves, at most 1024 threads
      const int threads = quad_threads;
      // start skew (see the kernel): the workgroups resident at first, in units of s_sleep 127 = 8 128 cycles
      int cus = 256;
      (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
      const uint32_t skew_blocks = (uint32_t)std::max(cus, 1) * (quad_lds <= 80 * 1024 ? 2u : 1u);
      int skew_unit = blocks > 2 * (uint64_t)skew_blocks ? 1 : 0;
      if (const char *e = getenv("NEEDLE_HIP_RESAMPLE_SKEW")) skew_unit = std::max(0, atoi(e));  // tuning
      KernelTimer timer("resample");
      const float4 *coefq = reinterpret_cast<const float4 *>(d->d_coefq);
      auto launch = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, dim3((uint32_t)blocks), dim3(threads), quad_lds, stream, d_in, w.first->ptr,
                           (int)meta.size(), coefq, static_cast<const QuadInfo *>(d->d_quad_info), geo, d->steps, quad_splits,
                           skew_blocks, skew_unit, d_out);
      };
      const bool small = threads <= 640;
#ifdef NEEDLE_HIP_LAB_BUILD
      const int lab = getenv("NEEDLE_HIP_RESAMPLE_LAB") ? atoi(getenv("NEEDLE_HIP_RESAMPLE_LAB")) : 0;
#else
      const int lab = 0;  // the product has no wrong-result variants
#endif
      if (lab && channels == 2 && vec4 && small) {  // timing experiments (tools/rslab.sh): wrong results on purpose
#ifdef NEEDLE_HIP_LAB_BUILD
        if (lab == 1) launch(resample_quad_kernel<2, true, true, 1>);
        else if (lab == 2) launch(resample_quad_kernel<2, true, true, 2>);
        else if (lab == 3) launch(resample_quad_kernel<2, true, true, 3>);
        else if (lab == 4) launch(resample_quad_kernel<2, true, true, 4>);
        else launch(resample_quad_kernel<2, true, true, 5>);
#endif
      } else if (channels == 1) {
        if (vec4 && small) launch(resample_quad_kernel<1, true, true>);
        else if (vec4) launch(resample_quad_kernel<1, true, false>);
        else if (small) launch(resample_quad_kernel<1, false, true>);
        else launch(resample_quad_kernel<1, false, false>);
      } else {
        if (vec4 && small) launch(resample_quad_kernel<2, true, true>);
        else if (vec4) launch(resample_quad_kernel<2, true, false>);
        else if (small) launch(resample_quad_kernel<2, false, true>);
        else launch(resample_quad_kernel<2, false, false>);
      }
      NEEDLE_HIP_TRY(hipGetLastError());
      if (sync) NEEDLE_HIP_TRY(hipStreamSynchronize(library_stream()));
      return Status::Ok();
    }
    geo.n_log2 = 0;
    while ((1 << geo.n_log2) < d->n) geo.n_log2++;
    const bool row_mode = d->M >= kRowModeMinM && d->n <= kRowModeMaxN;
    if (row_mode) {  // n rows of M + 4G + 4 samples, pitch odd in slots
      geo.pitch = ((d->M + 4 * d->G + 4 + 3) / 4) | 1;
      geo.region_slots = d->n * geo.pitch;
    } else {         // the n M + T - 1 inputs a tile reads, plus the taps the shifted rows add at either end
      geo.pitch = 0;
      geo.region_slots = (d->n * d->M + 4 * d->G + 4 + 3) / 4;
    }
    const int rows_per_wave = d->n >= 64 ? 1 : 64 / d->n;
    // (very long filters, from rates far above 100 kHz: the per-wave row scratch would not fit; coefficients then
    // come straight from global memory)
    const bool rows_in_lds = (row_mode || d->M % 4 == 0) && (size_t)(kThreads / 64) * (d->n >= 64 ? 1 : 64 / d->n) * d->G * 16 <= 48 * 1024;
    // M % 4 == 0: a tile's first tap is a fixed number of samples past a multiple of 4; the region starts at that
    // multiple, so rows can be staged by aligned groups of four samples
    const bool vec4 = d->M % 4 == 0;
    geo.delta = vec4 ? ((1 - d->T / 2) % 4 + 4) % 4 : 0;
    const size_t lds_bytes = (size_t)geo.region_slots * 16 + ((tile_outputs * 2 + 15) & ~(size_t)15) +
                             (rows_in_lds ? (size_t)(kThreads / 64) * rows_per_wave * d->G * 16 : 0);
    if (lds_bytes > 160 * 1024) return Status::Make(NeedleError_InvalidArgument, "resample: rate ratio too large for this kernel");
    const void *variants[6] = {reinterpret_cast<const void *>(resample_kernel<1, false, false>),
                               reinterpret_cast<const void *>(resample_kernel<1, true, false>),
                               reinterpret_cast<const void *>(resample_kernel<1, true, true>),
                               reinterpret_cast<const void *>(resample_kernel<2, false, false>),
                               reinterpret_cast<const void *>(resample_kernel<2, true, false>),
                               reinterpret_cast<const void *>(resample_kernel<2, true, true>)};
    static std::map<int, size_t> lds_attr;  // largest dynamic LDS size announced per device
    if (lds_attr[dev] < lds_bytes) {
      for (const void *fn : variants)
        NEEDLE_HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
      lds_attr[dev] = lds_bytes;
    }
    KernelTimer timer("resample");
    const float4 *coef4 = reinterpret_cast<const float4 *>(d->d_coef);
    auto launch = [&](auto kernel) {
      hipLaunchKernelGGL(kernel, dim3((uint32_t)blocks), dim3(kThreads), lds_bytes, stream, d_in, w.first->ptr,
                         (int)meta.size(), coef4, geo, d_out);
    };
    if (channels == 1) {
      if (vec4 && rows_in_lds) launch(resample_kernel<1, true, true>);
      else if (rows_in_lds) launch(resample_kernel<1, true, false>);
      else launch(resample_kernel<1, false, false>);
    } else {
      if (vec4 && rows_in_lds) launch(resample_kernel<2, true, true>);
      else if (rows_in_lds) launch(resample_kernel<2, true, false>);
      else launch(resample_kernel<2, false, false>);
    }
    NEEDLE_HIP_TRY(hipGetLastError());
  }
  if (sync) NEEDLE_HIP_TRY(hipStreamSynchronize(library_stream()));
  return Status::Ok();
}

Status gpu_resample_host(const std::vector<const int16_t *> &pcm, const std::vector<size_t> &num_values, int channels,
                         int rate, std::vector<std::vector<int16_t>> *out) {
  std::lock_guard<std::recursive_mutex> gpu_lock(gpu_mutex());
  Status s = ensure_device();
  if (!s.ok()) return s;
  if (channels != 1 && channels != 2) return Status::Make(NeedleError_InvalidArgument, "resample: channels must be 1 or 2");
  const size_t n = pcm.size();
  out->assign(n, {});
  std::vector<ResampleSpan> spans(n);
  uint64_t in_total = 0, out_total = 0;
  for (size_t i = 0; i < n; i++) {
    spans[i].in_off = in_total;
    spans[i].n_in = num_values[i] / (size_t)channels;
    spans[i].out_off = out_total;
    in_total += (num_values[i] + 7) & ~(uint64_t)7;  // 16-byte aligned stream starts: staging by aligned groups
    out_total += (resample_out_len(spans[i].n_in, rate) + 1) & ~(uint64_t)1;
  }
  DeviceBuffer<int16_t> d_in, d_out;
  if (!(s = d_in.reserve(std::max<uint64_t>(in_total, 1))).ok()) return s;
  if (!(s = d_out.reserve(std::max<uint64_t>(out_total, 1))).ok()) return s;
  hipStream_t stream = library_stream();
  for (size_t i = 0; i < n; i++)
    if (num_values[i])
      NEEDLE_HIP_TRY(hipMemcpyAsync(d_in.ptr + spans[i].in_off, pcm[i], num_values[i] * sizeof(int16_t),
                                    hipMemcpyHostToDevice, stream));
  s = gpu_resample_device(d_in.ptr, spans, channels, rate, d_out.ptr, false);
  if (!s.ok()) return s;
  // measurement (tools/bench_resample.py): the kernel again on the resident input, so that the timed launch follows
  // another launch and not the copies
  if (const char *e = getenv("NEEDLE_HIP_RESAMPLE_REPEAT"))
    for (int k = 0; k < atoi(e); k++)
      if (!(s = gpu_resample_device(d_in.ptr, spans, channels, rate, d_out.ptr, false)).ok()) return s;
  std::vector<int16_t> host(std::max<uint64_t>(out_total, 1));
  NEEDLE_HIP_TRY(hipMemcpyAsync(host.data(), d_out.ptr, out_total * sizeof(int16_t), hipMemcpyDeviceToHost, stream));
  NEEDLE_HIP_TRY(hipStreamSynchronize(stream));
  for (size_t i = 0; i < n; i++) {
    const size_t k = resample_out_len(spans[i].n_in, rate);
    (*out)[i].assign(host.begin() + spans[i].out_off, host.begin() + spans[i].out_off + k);
  }
  return Status::Ok();
}

}  // namespace needle
