// Per-thread arithmetic of the fingerprint kernels, written once and shared by fingerprint.hip (device)
// and tests/cpu_emu (a g++ build that steps the same per-thread code serially, phase by phase, to
// validate indexing without a GPU; it is a test fixture, never a product path).
//
// STFT: TWO real frames per 4096-point complex FFT (z = frameA + i*frameB), radix 16,16,16, decimation in
// frequency in place, executed by 256 threads that each own one 16-point butterfly per stage and exchange
// through one padded LDS buffer; then the real-input split, |X|^2 and the 12-class chroma fold.
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#define NEEDLE_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define NEEDLE_HD inline
#endif

namespace needle {
namespace core {

struct cd {  // 8-byte aligned on purpose: LDS traffic as paired 64-bit accesses measured faster than b128 here
  double x, y;
};

NEEDLE_HD cd cadd(cd a, cd b) { return cd{a.x + b.x, a.y + b.y}; }
NEEDLE_HD cd csub(cd a, cd b) { return cd{a.x - b.x, a.y - b.y}; }
NEEDLE_HD cd cmul(cd a, cd b) { return cd{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
// complex multiply with explicit fused multiply-adds (2 mul + 2 fma): same on host (fma()) and device
NEEDLE_HD cd cmulf(cd a, cd b) {
  return cd{__builtin_fma(a.x, b.x, -(a.y * b.y)), __builtin_fma(a.x, b.y, a.y * b.x)};
}
NEEDLE_HD cd mul_neg_i(cd a) { return cd{a.y, -a.x}; }

constexpr int kThreads = 256;    // threads per frame pair
constexpr int kMinBin = 10;      // max(1, round(4096*28/11025))
constexpr int kMaxBin = 1308;    // min(2048, round(4096*3520/11025)), exclusive
constexpr int kNumBins = kMaxBin - kMinBin;
constexpr int kBinsPerThread = 6;  // registers j = 0..5 of a thread hold every bin below 1536

// One LDS buffer of 4096 complex slots whose index is padded by one slot per 16 (pidx): the stride-16 and
// stride-17 access patterns of the stages are then bank-conflict free for 16-byte elements.
constexpr int kFft2N = 4096;
constexpr int kLds2Slots = kFft2N + kFft2N / 16;  // padded complex slots

NEEDLE_HD int pidx(int i) { return i + (i >> 4); }

NEEDLE_HD void bfly4(cd &a0, cd &a1, cd &a2, cd &a3) {
  cd e0 = cadd(a0, a2), e1 = csub(a0, a2), e2 = cadd(a1, a3), e3 = mul_neg_i(csub(a1, a3));
  a0 = cadd(e0, e2);
  a1 = cadd(e1, e3);
  a2 = csub(e0, e2);
  a3 = csub(e1, e3);
}

// 16-point forward DFT in place.  Output X[k], k = k1 + 4 k2, is left in a[4 k1 + k2]; out16(j) gives the
// register holding X[j].
NEEDLE_HD int out16(int j) { return 4 * (j & 3) + (j >> 2); }

NEEDLE_HD void fft16(cd *a) {
  const double c = 0.92387953251128673848, s = 0.38268343236508977173, h = 0.70710678118654752440;
  // step 1: 4-point DFTs over n1 for each n2 (x[4 n1 + n2]); result y[n2][k1] -> a[4 k1 + n2]
#pragma unroll
  for (int n2 = 0; n2 < 4; n2++) bfly4(a[n2], a[4 + n2], a[8 + n2], a[12 + n2]);
  // step 2: twiddles W_16^{n2 k1}
  {
    cd v;
    v = a[4 * 1 + 1]; a[4 * 1 + 1] = cd{v.x * c + v.y * s, v.y * c - v.x * s};            // W^1 = (c, -s)
    v = a[4 * 2 + 1]; a[4 * 2 + 1] = cd{(v.x + v.y) * h, (v.y - v.x) * h};                // W^2 = (h, -h)
    v = a[4 * 3 + 1]; a[4 * 3 + 1] = cd{v.x * s + v.y * c, v.y * s - v.x * c};            // W^3 = (s, -c)
    v = a[4 * 1 + 2]; a[4 * 1 + 2] = cd{(v.x + v.y) * h, (v.y - v.x) * h};                // W^2
    v = a[4 * 2 + 2]; a[4 * 2 + 2] = cd{v.y, -v.x};                                        // W^4 = -i
    v = a[4 * 3 + 2]; a[4 * 3 + 2] = cd{(v.y - v.x) * h, -(v.x + v.y) * h};               // W^6 = (-h, -h)
    v = a[4 * 1 + 3]; a[4 * 1 + 3] = cd{v.x * s + v.y * c, v.y * s - v.x * c};            // W^3
    v = a[4 * 2 + 3]; a[4 * 2 + 3] = cd{(v.y - v.x) * h, -(v.x + v.y) * h};               // W^6
    v = a[4 * 3 + 3]; a[4 * 3 + 3] = cd{-(v.x * c) - v.y * s, v.x * s - v.y * c};          // W^9 = (-c, s)
  }
  // step 3: 4-point DFTs over n2 for each k1
#pragma unroll
  for (int k1 = 0; k1 < 4; k1++) bfly4(a[4 * k1], a[4 * k1 + 1], a[4 * k1 + 2], a[4 * k1 + 3]);
}

// One complex slot as a single 128-bit LDS read (slots are 16-byte aligned); stores stay paired 64-bit.
NEEDLE_HD cd lds_get(const cd *lds, int slot) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef double v2d __attribute__((ext_vector_type(2), aligned(16)));
  const v2d v = *reinterpret_cast<const v2d *>(lds + slot);
  return cd{v.x, v.y};
#else
  return lds[slot];
#endif
}

// ================================================================================================
// Schedule of the 4096-point transform: decimation in frequency IN PLACE.  With n = 256 n2 + 16 n1 + n0,
//   stage 0 (thread t = 16 n1 + n0) transforms digit n2:  slots t + 256 k      -> same slots, * W_4096^{t j}
//   stage 1 (thread t = 16 b  + n0) transforms digit n1:  slots 256 b + n0 + 16 k -> same slots, * W_4096^{16 n0 j}
//   stage 2 (thread t = 16 b  + c ) transforms digit n0:  slots 16 t + k
// so X[k0 + 16 k1 + 256 k2] ends in slot 256 k0 + 16 k1 + k2, i.e. in register j = k2 of thread t = 16 k0 + k1.
// Every thread reads and writes the SAME slots within a stage, so only two exchanges need a workgroup barrier:
//   stage 0 -> 1  crosses waves (barrier);
//   stage 1 -> 2  stays inside the 16 consecutive lanes b = t >> 4 (slots [256 b, 256 b + 256)): LDS operations of
//                 one wave execute in order, no barrier;
//   stage 2 -> publish: a thread overwrites only slots it alone has read, no barrier;
//   publish -> partner reads crosses waves (barrier).
// The last stage only publishes the six registers (j = 10..15) that other threads need as partners Z[N - k] of the
// bins 10..1307; the bins themselves (j = 0..5) never leave the registers.  The class-sorted powers then go to the
// slots of registers j = 0..7 (dif_power_index), which are dead by then and disjoint from the partner slots, so the
// partner reads and the power stores need no barrier between them either.
// ================================================================================================
NEEDLE_HD int dif_slot_of_bin(int kf) { return 256 * (kf & 15) + 16 * ((kf >> 4) & 15) + (kf >> 8); }
NEEDLE_HD int dif_bin_of(int t, int j) { return (t >> 4) + 16 * (t & 15) + 256 * j; }

// stage 0: r holds the inputs x[t + 256 k]; leaves the stage's outputs in the same slots.  Split in two so the
// kernel can keep the register-only half ahead of the barrier that frees the LDS image of the previous pair.
NEEDLE_HD void dif0_store(int t, cd base0, cd *lds, const cd *r) {
  lds[pidx(t)] = r[out16(0)];
  cd w = base0;
#pragma unroll
  for (int j = 1; j < 16; j++) {
    lds[pidx(t + 256 * j)] = cmulf(r[out16(j)], w);
    if (j < 15) w = cmulf(w, base0);
  }
}
NEEDLE_HD void dif0(int t, cd base0, cd *lds, cd *r) {
  fft16(r);
  dif0_store(t, base0, lds, r);
}

// stage 1, in place; base1 = W_4096^{16 (t & 15)}
NEEDLE_HD void dif1(int t, cd base1, cd *lds, cd *r) {
  const int o = 256 * (t >> 4) + (t & 15);
#pragma unroll
  for (int k = 0; k < 16; k++) r[k] = lds_get(lds, pidx(o + 16 * k));
  fft16(r);
  lds[pidx(o)] = r[out16(0)];
  cd w = base1;
#pragma unroll
  for (int j = 1; j < 16; j++) {
    lds[pidx(o + 16 * j)] = cmulf(r[out16(j)], w);
    if (j < 15) w = cmulf(w, base1);
  }
}

// stage 2: afterwards r[out16(j)] = Z[dif_bin_of(t, j)]
NEEDLE_HD void dif2(int t, const cd *lds, cd *r) {
#pragma unroll
  for (int k = 0; k < 16; k++) r[k] = lds_get(lds, pidx(16 * t + k));
  fft16(r);
}

// publish the registers other threads read as partners (bins >= 2789 live in j = 10..15), in place
NEEDLE_HD void dif2_publish(int t, cd *lds, const cd *r) {
#pragma unroll
  for (int j = 10; j < 16; j++) lds[pidx(16 * t + j)] = r[out16(j)];
}

// ---- Hamming window by recurrence -------------------------------------------------------------------------------
// chromaprint's window is w[n] = (0.54 - 0.46 cos(theta n)) / 32767, theta = 2 pi / 4095.  A thread needs it at
// n = t + 256 k, k = 0..15: cos(theta (t + 256 (k + 1))) = 2 cos(256 theta) cos(theta (t + 256 k)) -
// cos(theta (t + 256 (k - 1))), one fused multiply-add per step from two per-thread starting values, and one more
// for A - B cos: two instructions per sample instead of an 8-byte load per sample from a 32 KB table that every
// workgroup re-read for every frame pair (measured: 6 % of the kernel).  Over 16 steps the recurrence stays
// within 8e-15 (relative) of the table values.
struct WindowConst {
  double k2;    // 2 cos(256 theta)
  double a, b;  // w = a - b cos(theta n); both carry kPairInputScale / 32767
};
NEEDLE_HD double window_step(const WindowConst &wc, double *c, double *c_prev) {
  const double w = __builtin_fma(-wc.b, *c, wc.a);
  const double next = __builtin_fma(wc.k2, *c, -*c_prev);
  *c_prev = *c;
  *c = next;
  return w;
}

// The transform's inputs carry this factor (it is folded into the window table), so that the split of Z into the
// two real spectra needs no halving: X_A = (Z[k] + conj Z[N-k]) / 2, X_B = (Z[k] - conj Z[N-k]) / 2i.  A power of
// two, so every intermediate is the unscaled one times 2^-1 exactly and the powers are bit-identical.
constexpr double kPairInputScale = 0.5;

// powers of this thread's bin in register j for the two frames (any bin 0 < k < 4096; bin 0 reads a slot that
// holds something else, for callers that discard it)
NEEDLE_HD void dif_bin_power_any(int t, int j, const cd *lds, const cd *r, double *pa, double *pb) {
  const int kf = dif_bin_of(t, j);
  const cd z = r[out16(j)], y = lds_get(lds, pidx(dif_slot_of_bin(kFft2N - kf)));
  const double ar = z.x + y.x, ai = z.y - y.y;  // X_A
  const double br = z.y + y.y, bi = y.x - z.x;  // X_B
  *pa = ar * ar + ai * ai;
  *pb = br * br + bi * bi;
}
// the same, j = 0..5; false if the bin is outside 10..1307
NEEDLE_HD bool dif_bin_power(int t, int j, const cd *lds, const cd *r, int *kf_out, double *pa, double *pb) {
  const int kf = dif_bin_of(t, j);
  *kf_out = kf;
  if (kf < kMinBin || kf >= kMaxBin) return false;
  dif_bin_power_any(t, j, lds, r, pa, pb);
  return true;
}

// Class-sorted power image.  Position p (0..1297, the bins sorted by pitch class) holds the PAIR (power in frame
// A, power in frame B) in one 16-byte slot: slot 17 (p >> 3) + (p & 7), i.e. registers j = 0..7 of thread
// (p >> 3)'s stage-2 row (pidx(16 q + j) = 17 q + j), dead by then and disjoint from the partner slots.  One
// 128-bit store per bin, and one 128-bit load gives a fold lane both frames' powers: LDS operations, not their
// bytes, are what this kernel pays for.
constexpr int kClassLanes = 16;      // lanes that share one pitch class in the fold (one DPP row)
constexpr int kClassLaneMax = 9;     // >= ceil(largest class / kClassLanes); checked where the tables are built
NEEDLE_HD int dif_power_slot(int p) { return 17 * (p >> 3) + (p & 7); }
// Two constant slots make the stores and the fold's loads branch-free.  kPowerZeroSlot: a pad slot (17 q + 16:
// touched by no stage and no power) that holds (0, 0), read in place of positions outside a class.
// kPowerTrashSlot: a row beyond the last position, where the powers of the bins outside 10..1307 go.
constexpr int kPowerZeroSlot = 16, kPowerTrashSlot = 17 * 200;

// One lane's share of a pitch class [b0, b1): positions b0 + l, b0 + l + 16, ... (every other row of the image:
// slots 34 apart), both frames at once (x = frame A, y = frame B).  Split into the loads and the sums so the
// kernel can put other work between them.
NEEDLE_HD void class_lane_load(const cd *lds, int b0, int b1, int l, cd *v) {
#pragma unroll
  for (int i = 0; i < kClassLaneMax; i++) {
    const int b = b0 + l + kClassLanes * i;
    v[i] = lds_get(lds, b < b1 ? dif_power_slot(b) : kPowerZeroSlot);
  }
}
NEEDLE_HD cd class_lane_add(const cd *v) {
  cd acc = v[0];
#pragma unroll
  for (int i = 1; i < kClassLaneMax; i++) acc = cadd(acc, v[i]);
  return acc;
}
// The 16 lane sums of a class are then combined in four exchange steps; partner of lane l in step s:
NEEDLE_HD int class_tree_partner(int l, int s) { return s == 0 ? (l ^ 1) : s == 1 ? (l ^ 2) : s == 2 ? ((l & 8) | (7 - (l & 7))) : (15 - l); }

// ---- classifiers (chromaprint kClassifiersTest2) ---------------------------------
struct ClassifierDef {
  int type, y, h, w;  // Filter(type, y, height, width)
};

constexpr ClassifierDef kClassifiers[16] = {
    {0, 4, 3, 15}, {4, 4, 6, 15}, {1, 0, 4, 16}, {3, 8, 2, 12}, {3, 4, 4, 8}, {4, 0, 3, 5},
    {1, 2, 2, 9},  {2, 7, 3, 4},  {2, 6, 2, 16}, {2, 1, 3, 2},  {5, 10, 1, 15}, {3, 6, 2, 10},
    {2, 1, 1, 14}, {3, 5, 6, 4},  {1, 9, 2, 12}, {3, 4, 2, 14},
};

// Membership of image cell (row r = time offset within the 16-row window, column c = band) in the
// classifier's "a" (+1) or "b" (-1) region, 0 if in neither.  Pure function of constants, so the
// fully unrolled classify loop folds to straight-line adds.
constexpr int cell_sign(const ClassifierDef &f, int r, int c) {
  const int x = r, y = c;
  if (x < 0 || x >= f.w || y < f.y || y >= f.y + f.h) return 0;
  const int yy = y - f.y;
  switch (f.type) {
    case 0:
      return 1;
    case 1:  // a = high half in y, b = low half
      return yy >= f.h / 2 ? 1 : -1;
    case 2:  // a = later half in x, b = earlier half
      return x >= f.w / 2 ? 1 : -1;
    case 3: {  // a = (early x, high y) + (late x, low y)
      const bool late = x >= f.w / 2, high = yy >= f.h / 2;
      return (late != high) ? 1 : -1;
    }
    case 4: {  // a = middle third in y
      const int h3 = f.h / 3;
      return (yy >= h3 && yy < 2 * h3) ? 1 : -1;
    }
    default: {  // 5: a = middle third in x
      const int w3 = f.w / 3;
      return (x >= w3 && x < 2 * w3) ? 1 : -1;
    }
  }
}

// thresholds as exp(t): log((1+a)/(1+b)) < t  <=>  (1+a)/(1+b) < exp(t)
struct ClassifierThresholds {
  double e[16][3];
};

// Compile-time walk over the 16x12 window: cell (R, C) is added to accumulator a[I] / b[I] of every
// classifier whose region holds it; all membership tests fold away.
template <int R, int C, int I>
struct CellStep {
  static NEEDLE_HD void run(double v, double *a, double *b) {
    constexpr int s = cell_sign(kClassifiers[I], R, C);
    if (s > 0) a[I] += v;
    if (s < 0) b[I] += v;
    CellStep<R, C, I + 1>::run(v, a, b);
  }
};
template <int R, int C>
struct CellStep<R, C, 16> {
  static NEEDLE_HD void run(double, double *, double *) {}
};
template <int R, int C, int PITCH>
struct WindowStep {
  static NEEDLE_HD void run(const double *w, double *a, double *b) {
    CellStep<R, C, 0>::run(w[R * PITCH + C], a, b);
    WindowStep<(C == 11) ? R + 1 : R, (C == 11) ? 0 : C + 1, PITCH>::run(w, a, b);
  }
};
template <int PITCH>
struct WindowStep<16, 0, PITCH> {
  static NEEDLE_HD void run(const double *, double *, double *) {}
};

// One raw fingerprint item from 16 consecutive feature rows (w[16][12], row-major).
// PITCH = doubles between consecutive rows (12 = packed).
template <int PITCH = 12>
NEEDLE_HD uint32_t classify_window(const double *w, const ClassifierThresholds *thr) {
  double a[16], b[16];
#pragma unroll
  for (int i = 0; i < 16; i++) a[i] = b[i] = 0.0;
  WindowStep<0, 0, PITCH>::run(w, a, b);
  uint32_t bits = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) {
    const double ratio = (1.0 + a[i]) / (1.0 + b[i]);
    // Quantizer: v < t1 ? (v < t0 ? 0 : 1) : (v < t2 ? 2 : 3), on the exp() side of the monotone map
    const unsigned q = ratio < thr->e[i][1] ? (ratio < thr->e[i][0] ? 0u : 1u) : (ratio < thr->e[i][2] ? 2u : 3u);
    bits = (bits << 2) | (q ^ (q >> 1));  // Gray code {0,1,3,2}
  }
  return bits;
}

}  // namespace core
}  // namespace needle
