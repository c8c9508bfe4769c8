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

// One complex value.  The transform below is written once over the complex type C: cd (f64, the arithmetic the u32
// contract is defined on) and cf (f32, the certified first pass: stft32_kernel.h).
template <typename T>
struct cx {
  typedef T real;
  T x, y;
};
typedef cx<double> cd;
typedef cx<float> cf;

NEEDLE_HD double fmad(double a, double b, double c) { return __builtin_fma(a, b, c); }
NEEDLE_HD float fmad(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

template <class C> NEEDLE_HD C cadd(C a, C b) { return C{a.x + b.x, a.y + b.y}; }
template <class C> NEEDLE_HD C csub(C a, C b) { return C{a.x - b.x, a.y - b.y}; }
template <class C> NEEDLE_HD C cmul(C a, C b) { return C{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
// complex multiply with explicit fused multiply-adds (2 mul + 2 fma): same on host (fma()) and device
template <class C> NEEDLE_HD C cmulf(C a, C b) {
  return C{fmad(a.x, b.x, -(a.y * b.y)), fmad(a.x, b.y, a.y * b.x)};
}
template <class C> NEEDLE_HD C mul_neg_i(C a) { return C{a.y, -a.x}; }
// The forms the 16-point transform is made of (one definition of WHAT is computed; NEEDLE_PK_F32 below only changes the
// instructions that compute it):
//   axpy(s, v, e) = e + s v        arot(s, v, e) = e + s (-i v)        (both as fused multiply-adds per component)
//   addrot(e, v)  = e + (-i v)     subrot(e, v)  = e - (-i v)
//   rot_m(v)      = v + (-i v) = (x + y, y - x)       rot_p(v) = v + i v = (x - y, x + y)
template <class C> NEEDLE_HD C axpy(typename C::real s, C v, C e) { return C{fmad(s, v.x, e.x), fmad(s, v.y, e.y)}; }
template <class C> NEEDLE_HD C arot(typename C::real s, C v, C e) { return C{fmad(s, v.y, e.x), fmad(-s, v.x, e.y)}; }
template <class C> NEEDLE_HD C addrot(C e, C v) { return C{e.x + v.y, e.y - v.x}; }
template <class C> NEEDLE_HD C subrot(C e, C v) { return C{e.x - v.y, e.y + v.x}; }
template <class C> NEEDLE_HD C rot_m(C v) { return C{v.x + v.y, v.y - v.x}; }
template <class C> NEEDLE_HD C rot_p(C v) { return C{v.x - v.y, v.x + v.y}; }

#if defined(__HIP_DEVICE_COMPILE__) && defined(NEEDLE_PK_CMUL) && !defined(NEEDLE_PK_F32)
// Only the twiddle products in packed form (round 4): on gfx950 a v_fma_f32 with a register accumulator costs a wave 8.0
// cycles once a second wave issues on its SIMD and a v_pk_fma_f32 8.8 for TWO of them (profiles/r04_issue_rates.log), so
// the 2 mul + 2 fma of a complex product (25 cycles) become 1 v_pk_mul + 1 v_pk_fma (18); the additions stay scalar (a
// packed add costs what two plain ones do, and packing everything spills: profiles/NOTES.md round 3).
typedef float pk2f __attribute__((ext_vector_type(2)));
template <> NEEDLE_HD cf cmulf<cf>(cf a, cf b) {
  const pk2f t = pk2f{a.y, a.y} * pk2f{b.y, b.x};
  const pk2f r = __builtin_elementwise_fma(pk2f{a.x, a.x}, pk2f{b.x, b.y}, pk2f{-t.x, t.y});
  return cf{r.x, r.y};
}
#endif
#if defined(__HIP_DEVICE_COMPILE__) && defined(NEEDLE_PK_F32)
// The same per-component operations on cf as two-component vector operations, which the compiler issues as the packed
// instructions (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32: one instruction for the real and the imaginary part, swaps and
// signs in the operand modifiers).  Every component sees exactly the operation of the generic form above: same results.
typedef float pk2f __attribute__((ext_vector_type(2)));
NEEDLE_HD pk2f pk(cf a) { return pk2f{a.x, a.y}; }
NEEDLE_HD cf unpk(pk2f a) { return cf{a.x, a.y}; }
template <> NEEDLE_HD cf cadd<cf>(cf a, cf b) { return unpk(pk(a) + pk(b)); }
template <> NEEDLE_HD cf csub<cf>(cf a, cf b) { return unpk(pk(a) - pk(b)); }
template <> NEEDLE_HD cf cmulf<cf>(cf a, cf b) {
  const pk2f t = pk2f{a.y, a.y} * pk2f{b.y, b.x};
  return unpk(__builtin_elementwise_fma(pk2f{a.x, a.x}, pk(b), pk2f{-t.x, t.y}));
}
template <> NEEDLE_HD cf axpy<cf>(float s, cf v, cf e) { return unpk(__builtin_elementwise_fma(pk2f{s, s}, pk(v), pk(e))); }
template <> NEEDLE_HD cf arot<cf>(float s, cf v, cf e) { return unpk(__builtin_elementwise_fma(pk2f{s, -s}, pk2f{v.y, v.x}, pk(e))); }
template <> NEEDLE_HD cf addrot<cf>(cf e, cf v) { return unpk(pk(e) + pk2f{v.y, -v.x}); }
template <> NEEDLE_HD cf subrot<cf>(cf e, cf v) { return unpk(pk(e) - pk2f{v.y, -v.x}); }
template <> NEEDLE_HD cf rot_m<cf>(cf v) { return unpk(pk(v) + pk2f{v.y, -v.x}); }
template <> NEEDLE_HD cf rot_p<cf>(cf v) { return unpk(pk(v) + pk2f{-v.y, v.x}); }
#endif

constexpr int kThreads = 256;    // threads per frame pair
constexpr int kMinBin = 10;      // max(1, round(4096*28/11025))
constexpr int kMaxBin = 1308;    // min(2048, round(4096*3520/11025)), exclusive
constexpr int kNumBins = kMaxBin - kMinBin;
constexpr int kBinsPerThread = 6;  // registers j = 0..5 of a thread hold every bin below 1536

// One LDS buffer of 4096 complex slots whose index is padded by one slot per 16 (pidx): the stride-16 and
// stride-17 access patterns of the stages are then bank-conflict free for 16-byte elements.
constexpr int kFft2N = 4096;
constexpr int kFftSlots = kFft2N + kFft2N / 16;  // padded complex slots of the transform's image
// Behind the image: the constant zero and the trash slot of the power image.  A thread's private slot for the loop
// invariants that do not fit its registers is its pad inside the image (slot 17 t + 16, touched by no stage).
constexpr int kPowerZeroSlot = kFftSlots;               // holds (0, 0): read in place of positions beyond a lane's count
constexpr int kPowerTrashSlot = kPowerZeroSlot + 1;     // takes the powers of the bins outside 10..1307, never read
constexpr int kLds2Slots = kPowerTrashSlot + 1;         // 69 664 bytes: two workgroups per CU
NEEDLE_HD int thread_pad_slot(int t) { return 17 * t + 16; }  // window recurrence seeds

NEEDLE_HD int pidx(int i) { return i + (i >> 4); }

template <class C> NEEDLE_HD void bfly4(C &a0, C &a1, C &a2, C &a3) {
  C e0 = cadd(a0, a2), e1 = csub(a0, a2), e2 = cadd(a1, a3), d = csub(a1, a3);
  a0 = cadd(e0, e2);
  a1 = addrot(e1, d);
  a2 = csub(e0, e2);
  a3 = subrot(e1, d);
}

// 16-point forward DFT in place.  Output X[k], k = k1 + 4 k2, is left in a[4 k1 + k2]; out16(j) gives the
// register holding X[j].
NEEDLE_HD int out16(int j) { return 4 * (j & 3) + (j >> 2); }

// Last layer of the 16-point transform for one k1: bfly4 over (b0, w1 b1, w2 b2, w3 b3) with the constant
// twiddles folded into fused multiply-adds.  The caller passes E0 = b0 + w2 b2, E1 = b0 - w2 b2 and, for the odd
// pair, v = u1 + rho u3, v' = u1 - rho u3 where w1 b1 = g u1 and w3 b3 = g rho u3: then E2 = g v, E3 = -i g v'
// are never formed, the four outputs are E0 +- g v and E1 +- g (-i v') (8 FMAs in place of 8 adds).
template <class C>
NEEDLE_HD void bfly4_tail(C E0, C E1, C v, C vp, typename C::real g, C &o0, C &o1, C &o2, C &o3) {
  o0 = C{fmad(g, v.x, E0.x), fmad(g, v.y, E0.y)};
  o2 = C{fmad(-g, v.x, E0.x), fmad(-g, v.y, E0.y)};
  o1 = C{fmad(g, vp.y, E1.x), fmad(-g, vp.x, E1.y)};
  o3 = C{fmad(-g, vp.y, E1.x), fmad(g, vp.x, E1.y)};
}

// The transform in two parts, so that a caller can start storing the outputs of one k1 while the next is computed:
// fft16_head = step 1 (4-point DFTs over n1 for each n2; y[n2][k1] -> a[4 k1 + n2]); fft16_tail<K1> = the twiddles
// W_16^{n2 K1} and the 4-point DFT over n2, leaving X[K1 + 4 k2] in a[4 K1 + k2].
// W_16^1 = (c, -s), W_16^2 = (h, -h), W_16^3 = (s, -c), W_16^6 = (-h, -h), W_16^9 = (-c, s); t8 = s / c = tan(pi/8),
// ct8 = c / s.  A twiddle w = g (1, tau) costs two FMAs for u = (1, tau) x and its scale g rides on the FMAs of the
// following additions: 144 instructions per transform instead of 168.
template <class C> NEEDLE_HD void fft16_head(C *a) {
#pragma unroll
  for (int n2 = 0; n2 < 4; n2++) bfly4(a[n2], a[4 + n2], a[8 + n2], a[12 + n2]);
}
// `hook(i)`, i = 0..3, is called at four points of the tail: a caller can use it to issue one LDS store of the PREVIOUS
// tail's outputs at each, which spaces the stores out under this tail's arithmetic.  PLAIN: the round-1 arithmetic
// (constant twiddles multiplied out, then a 4-point DFT of additions: 168 instructions per transform), kept for
// tools/stft_lab.hip.
struct NoHook {
  NEEDLE_HD void operator()(int) const {}
};
template <int K1, typename HOOK = NoHook, bool PLAIN = false, class C = cd>
NEEDLE_HD void fft16_tail(C *a, HOOK hook = HOOK()) {
  typedef typename C::real T;
  const T c = (T)0.92387953251128673848, s = (T)0.38268343236508977173, h = (T)0.70710678118654752440;
  const T t8 = (T)0.41421356237309504880, ct8 = (T)2.41421356237309504880;
  C E0, E1, v, vp;
  T g = (T)1.0;
  if (K1 == 0 || PLAIN) {  // no twiddles left: v = b1 + b3, v' = b1 - b3, plain additions below
    C b0 = a[4 * K1], b1 = a[4 * K1 + 1], b2 = a[4 * K1 + 2], b3 = a[4 * K1 + 3];
    if (PLAIN && K1 == 1) {
      b1 = C{b1.x * c + b1.y * s, b1.y * c - b1.x * s};       // W1 = (c, -s)
      b2 = C{(b2.x + b2.y) * h, (b2.y - b2.x) * h};           // W2 = (h, -h)
      b3 = C{b3.x * s + b3.y * c, b3.y * s - b3.x * c};       // W3 = (s, -c)
    } else if (PLAIN && K1 == 2) {
      b1 = C{(b1.x + b1.y) * h, (b1.y - b1.x) * h};           // W2
      b2 = C{b2.y, -b2.x};                                    // W4 = -i
      b3 = C{(b3.y - b3.x) * h, -(b3.x + b3.y) * h};          // W6 = (-h, -h)
    } else if (PLAIN && K1 == 3) {
      b1 = C{b1.x * s + b1.y * c, b1.y * s - b1.x * c};       // W3
      b2 = C{(b2.y - b2.x) * h, -(b2.x + b2.y) * h};          // W6
      b3 = C{-(b3.x * c) - b3.y * s, b3.x * s - b3.y * c};    // W9 = (-c, s)
    }
    E0 = cadd(b0, b2);
    E1 = csub(b0, b2);
    hook(0);
    v = cadd(b1, b3);
    vp = csub(b1, b3);
    hook(1);
    a[4 * K1] = cadd(E0, v);
    a[4 * K1 + 2] = csub(E0, v);
    hook(2);
    a[4 * K1 + 1] = addrot(E1, vp);
    a[4 * K1 + 3] = subrot(E1, vp);
    hook(3);
    return;
  } else if (K1 == 1) {  // w = (1, W1, W2, W3)
    const C b0 = a[4], b1 = a[5], b2 = a[6], b3 = a[7];
    const C u2 = rot_m(b2);  // W2 b2 = h u2
    E0 = axpy(h, u2, b0);
    E1 = axpy(-h, u2, b0);
    hook(0);
    const C u1 = arot(t8, b1, b1);    // W1 b1 = c u1
    const C u3 = arot(ct8, b3, b3);   // W3 b3 = s u3 = c t8 u3
    v = axpy(t8, u3, u1);
    vp = axpy(-t8, u3, u1);
    g = c;
  } else if (K1 == 2) {  // w = (1, W2, W4, W6)
    const C b0 = a[8], b1 = a[9], b2 = a[10], b3 = a[11];
    E0 = addrot(b0, b2);  // W4 = -i
    E1 = subrot(b0, b2);
    hook(0);
    const C u1 = rot_m(b1);  // W2 b1 = h u1
    const C n3 = rot_p(b3);  // W6 b3 = -h n3
    v = csub(u1, n3);
    vp = cadd(u1, n3);
    g = h;
  } else {  // K1 == 3: w = (1, W3, W6, W9)
    const C b0 = a[12], b1 = a[13], b2 = a[14], b3 = a[15];
    const C n2 = rot_p(b2);  // W6 b2 = -h n2
    E0 = axpy(-h, n2, b0);
    E1 = axpy(h, n2, b0);
    hook(0);
    const C u1 = arot(ct8, b1, b1);  // W3 b1 = s u1
    const C u3 = arot(t8, b3, b3);   // W9 b3 = -c u3 = -s ct8 u3
    v = axpy(-ct8, u3, u1);
    vp = axpy(ct8, u3, u1);
    g = s;
  }
  hook(1);
  // E2 = g v and E3 = -i g v' are never formed: the outputs are E0 +- g v and E1 +- g (-i v')
  a[4 * K1] = axpy(g, v, E0);
  a[4 * K1 + 2] = axpy(-g, v, E0);
  hook(2);
  a[4 * K1 + 1] = arot(g, vp, E1);
  a[4 * K1 + 3] = arot(-g, vp, E1);
  hook(3);
}
template <class C> NEEDLE_HD void fft16(C *a) {
  fft16_head(a);
  fft16_tail<0, NoHook, false, C>(a);
  fft16_tail<1, NoHook, false, C>(a);
  fft16_tail<2, NoHook, false, C>(a);
  fft16_tail<3, NoHook, false, C>(a);
}

// One complex slot as a single LDS access (slots are naturally aligned): 128 bits for cd, 64 bits for cf.  Both take a
// 16-bit immediate byte offset, so every slot of the form base + constant costs no address arithmetic.
NEEDLE_HD cd lds_get(const cd *lds, int slot) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef double v2d __attribute__((ext_vector_type(2), aligned(16)));
  const v2d v = *reinterpret_cast<const v2d *>(lds + slot);
  return cd{v.x, v.y};
#else
  return lds[slot];
#endif
}
NEEDLE_HD void lds_put(cd *lds, int slot, cd v) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef double v2d __attribute__((ext_vector_type(2), aligned(16)));
  *reinterpret_cast<v2d *>(lds + slot) = v2d{v.x, v.y};
#else
  lds[slot] = v;
#endif
}
NEEDLE_HD cf lds_get(const cf *lds, int slot) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef float v2f __attribute__((ext_vector_type(2), aligned(8)));
  const v2f v = *reinterpret_cast<const v2f *>(lds + slot);
  return cf{v.x, v.y};
#else
  return lds[slot];
#endif
}
NEEDLE_HD void lds_put(cf *lds, int slot, cf v) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef float v2f __attribute__((ext_vector_type(2), aligned(8)));
  *reinterpret_cast<v2f *>(lds + slot) = v2f{v.x, v.y};
#else
  lds[slot] = v;
#endif
}
// the same by BYTE offset from the start of the image (offsets kept packed in registers, already scaled):
// pack_slots puts two 13-bit slots, each times 16, into one word; slot_bytes<0/1> takes them out again
struct Words4 {
  uint32_t w[4];
};
NEEDLE_HD Words4 lds_get_words(const cd *lds, int slot) {
  Words4 r;
#if defined(__HIP_DEVICE_COMPILE__)
  typedef uint32_t v4u __attribute__((ext_vector_type(4), aligned(16)));
  const v4u v = *reinterpret_cast<const v4u *>(lds + slot);
  r.w[0] = v.x; r.w[1] = v.y; r.w[2] = v.z; r.w[3] = v.w;
#else
  __builtin_memcpy(r.w, lds + slot, 16);
#endif
  return r;
}
NEEDLE_HD void lds_put_words(cd *lds, int slot, Words4 v) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef uint32_t v4u __attribute__((ext_vector_type(4), aligned(16)));
  *reinterpret_cast<v4u *>(lds + slot) = v4u{v.w[0], v.w[1], v.w[2], v.w[3]};
#else
  __builtin_memcpy(lds + slot, v.w, 16);
#endif
}
NEEDLE_HD uint32_t pack_slots(uint32_t a, uint32_t b) { return (a << 4) | (b << 17); }
template <int WHICH>
NEEDLE_HD uint32_t slot_bytes(uint32_t packed) { return WHICH == 0 ? (packed & 0x1fff0u) : ((packed >> 13) & 0x1fff0u); }
NEEDLE_HD void lds_put_bytes(cd *lds, uint32_t byte_off, cd v) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef double v2d __attribute__((ext_vector_type(2), aligned(16)));
  *reinterpret_cast<v2d *>(reinterpret_cast<char *>(lds) + byte_off) = v2d{v.x, v.y};
#else
  lds[byte_off / sizeof(cd)] = v;
#endif
}
// cf: the packed word still carries slot * 16 (one table for both widths); a cf slot is 8 bytes
NEEDLE_HD void lds_put_bytes(cf *lds, uint32_t byte_off16, cf v) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef float v2f __attribute__((ext_vector_type(2), aligned(8)));
  *reinterpret_cast<v2f *>(reinterpret_cast<char *>(lds) + (byte_off16 >> 1)) = v2f{v.x, v.y};
#else
  lds[byte_off16 / 16] = v;
#endif
}

// ================================================================================================
// Schedule of the 4096-point transform: decimation in frequency IN PLACE.  With n = 256 n2 + 16 n1 + n0,
//   stage 0 (thread t = 16 n1 + n0)      transforms digit n2:  slots t + 256 k          -> same slots, * W_4096^{t j}
//   stage 1 (thread t = 16 g + n0)       transforms digit n1:  slots 256 b + n0 + 16 k  -> same slots, * W_4096^{16 n0 j}
//   stage 2 (thread t = 16 g + c)        transforms digit n0:  slots 16 (16 b + c) + k
// where b = group_k0(g) is a fixed permutation of the sixteen 16-lane groups (below), so X[b + 16 c + 256 k2] ends in
// slot 256 b + 16 c + k2, i.e. in register j = k2 of the thread with group_k0(t >> 4) = b, t & 15 = c.
// Every thread reads and writes the SAME slots within a stage, so only ONE exchange of the transform needs a
// workgroup barrier:
//   stage 0 -> 1  crosses waves (barrier);
//   stage 1 -> 2  stays inside the 16 consecutive lanes g = t >> 4 (slots [256 b, 256 b + 256)): LDS operations of
//                 one wave execute in order, no barrier;
//   stage 2 -> publish: a thread overwrites only slots it alone has read, no barrier;
//   publish -> partner reads: the partner Z[N - k] of a bin with low digit b lives in the group with low digit
//                 16 - b, and the permutation puts b and 16 - b in the SAME WAVE: no barrier either;
//   power stores: every wave keeps the powers of its own bins in dead slots of its own rows (power layout below), so
//                 no other wave's stage-2 inputs can be overwritten: no barrier before them.
// The last stage only publishes the six registers (j = 10..15) that other threads need as partners Z[N - k] of the
// bins 10..1307; the bins themselves (j = 0..5) never leave the registers.
// ================================================================================================
// Low digits b of the four waves: {1, 2, 15, 14}, {3, 4, 13, 12}, {5, 6, 11, 10}, {7, 8, 9, 0}: b and (16 - b) & 15
// together, and each wave's rows 16 b + c form two runs of 32 consecutive rows (one of 48 + one of 16 in the last).
NEEDLE_HD int group_k0(int g) { return (int)((0x0987AB65CD43EF21ull >> (4 * g)) & 15u); }
NEEDLE_HD int thread_k0(int t) { return group_k0(t >> 4); }
NEEDLE_HD int wave_of_k0(int b) {  // the wave (t >> 6) that owns low digit b
  return (b == 1 || b == 2 || b == 15 || b == 14) ? 0 : (b == 3 || b == 4 || b == 13 || b == 12) ? 1
       : (b == 5 || b == 6 || b == 11 || b == 10) ? 2 : 3;
}

NEEDLE_HD int dif_slot_of_bin(int kf) { return 256 * (kf & 15) + 16 * ((kf >> 4) & 15) + (kf >> 8); }
NEEDLE_HD int dif_bin_of(int t, int j) { return thread_k0(t) + 16 * (t & 15) + 256 * j; }

// Padded slots of a thread's 16 values in each stage are ONE per-thread base plus a compile-time constant:
//   stage 0  pidx(t + 256 j)                 = dif0_base(t) + 272 j
//   stage 1  pidx(256 b + n0 + 16 k)         = dif1_base(t) + 17 k
//   stage 2  pidx(16 (16 b + c) + k)         = dif2_base(t) + k
//   partner  pidx(slot of bin N - (K + 256 j)) = dif_partner_base(t) + 15 - j   (K = dif_bin_of(t, 0))
NEEDLE_HD int dif0_base(int t) { return t + (t >> 4); }
NEEDLE_HD int dif1_base(int t) { return 272 * thread_k0(t) + (t & 15); }
NEEDLE_HD int dif2_base(int t) { return 17 * (16 * thread_k0(t) + (t & 15)); }
// N - (K + 256 j) = (256 - K) + 256 (15 - j) for K != 0: low byte K' = 256 - K, register 15 - j of the thread that
// owns K'.  K = 0: bins 256 j, partners 256 (16 - j) = slot 16 - j; j = 0 (bin 0, discarded) then reads a pad
// slot (16: thread 0's window seeds, harmless).
NEEDLE_HD int dif_partner_base(int t) {
  const int K = thread_k0(t) + 16 * (t & 15);
  const int Kp = (256 - K) & 255;
  return K == 0 ? 1 : 272 * (Kp & 15) + 17 * (Kp >> 4);
}

// stage 0: r holds the inputs x[t + 256 k]; leaves the stage's outputs in the same slots.  Split in two so the
// kernel can keep the register-only half ahead of the barrier that frees the LDS image of the previous pair.
template <class C> NEEDLE_HD void dif0_store(int t, C base0, C *lds, const C *r) {
  const int o = dif0_base(t);
  lds_put(lds, o, r[out16(0)]);
  C w = base0;
#pragma unroll
  for (int j = 1; j < 16; j++) {
    lds_put(lds, o + 272 * j, cmulf(r[out16(j)], w));
    if (j < 15) w = cmulf(w, base0);
  }
}
template <class C> NEEDLE_HD void dif0(int t, C base0, C *lds, C *r) {
  fft16(r);
  dif0_store(t, base0, lds, r);
}

// stage 1, in place; base1 = W_4096^{16 (t & 15)}
template <class C> NEEDLE_HD void dif1(int t, C base1, C *lds, C *r) {
  const int o = dif1_base(t);
#pragma unroll
  for (int k = 0; k < 16; k++) r[k] = lds_get(lds, o + 17 * k);
  fft16(r);
  lds_put(lds, o, r[out16(0)]);
  C w = base1;
#pragma unroll
  for (int j = 1; j < 16; j++) {
    lds_put(lds, o + 17 * j, cmulf(r[out16(j)], w));
    if (j < 15) w = cmulf(w, base1);
  }
}

// The same two stages with their stores streamed out under the butterflies instead of in a burst behind them.  pw[j] =
// base^j.  MODE bit 0 (kStoreSpaced): the four stores of tail k1 (outputs j = k1, k1 + 4, k1 + 8, k1 + 12, each with
// its twiddle multiply) go out one at a time at four points of tail k1 + 1, the last tail's own four behind it;
// otherwise each tail's four follow it directly.  MODE bit 1 (kPlainFft): round-1 arithmetic in the tails.
enum : int { kStoreSpaced = 1, kPlainFft = 2 };
template <class C> NEEDLE_HD void twiddle_powers(C base, C *pw) {
  pw[1] = base;
#pragma unroll
  for (int j = 2; j < 16; j++) pw[j] = cmulf(pw[j - 1], base);
}
template <int K1, class C = cd>
struct StoreHook {
  int o, pitch;
  C *lds;
  const C *r;
  const C *pw;
  NEEDLE_HD void operator()(int k2) const {
    const int j = K1 + 4 * k2;
    lds_put(lds, o + pitch * j, j == 0 ? r[4 * K1 + k2] : cmulf(r[4 * K1 + k2], pw[j]));
  }
};
template <int MODE, class C>
NEEDLE_HD void dif_tails_store(int o, int pitch, const C *pw, C *lds, C *r) {
  constexpr bool P = (MODE & kPlainFft) != 0;
  if (MODE & kStoreSpaced) {
    fft16_tail<0, NoHook, P>(r);
    fft16_tail<1, StoreHook<0, C>, P>(r, StoreHook<0, C>{o, pitch, lds, r, pw});
    fft16_tail<2, StoreHook<1, C>, P>(r, StoreHook<1, C>{o, pitch, lds, r, pw});
    fft16_tail<3, StoreHook<2, C>, P>(r, StoreHook<2, C>{o, pitch, lds, r, pw});
#pragma unroll
    for (int k2 = 0; k2 < 4; k2++) StoreHook<3, C>{o, pitch, lds, r, pw}(k2);
  } else {
    fft16_tail<0, NoHook, P>(r);
#pragma unroll
    for (int k2 = 0; k2 < 4; k2++) StoreHook<0, C>{o, pitch, lds, r, pw}(k2);
    fft16_tail<1, NoHook, P>(r);
#pragma unroll
    for (int k2 = 0; k2 < 4; k2++) StoreHook<1, C>{o, pitch, lds, r, pw}(k2);
    fft16_tail<2, NoHook, P>(r);
#pragma unroll
    for (int k2 = 0; k2 < 4; k2++) StoreHook<2, C>{o, pitch, lds, r, pw}(k2);
    fft16_tail<3, NoHook, P>(r);
#pragma unroll
    for (int k2 = 0; k2 < 4; k2++) StoreHook<3, C>{o, pitch, lds, r, pw}(k2);
  }
}
// (the _pw forms take the fifteen powers pw[1..15] of the stage's twiddle base ready-made: the f32 pass computes them
// in double once per thread, so that every twiddle is correctly rounded instead of a chain of f32 products)
template <int MODE = 0, class C>
NEEDLE_HD void dif0_streamed_pw(int t, const C *pw, C *lds, C *r) {
  fft16_head(r);
  dif_tails_store<MODE>(dif0_base(t), 272, pw, lds, r);
}
template <int MODE = 0, class C>
NEEDLE_HD void dif1_streamed_pw(int t, const C *pw, C *lds, C *r) {
  const int o = dif1_base(t);
#pragma unroll
  for (int k = 0; k < 16; k++) r[k] = lds_get(lds, o + 17 * k);
  fft16_head(r);
  dif_tails_store<MODE>(o, 17, pw, lds, r);
}
template <int MODE = 0, class C>
NEEDLE_HD void dif0_streamed(int t, C base0, C *lds, C *r) {
  C pw[16];
  twiddle_powers(base0, pw);
  dif0_streamed_pw<MODE>(t, pw, lds, r);
}
template <int MODE = 0, class C>
NEEDLE_HD void dif1_streamed(int t, C base1, C *lds, C *r) {
  C pw[16];
  twiddle_powers(base1, pw);
  dif1_streamed_pw<MODE>(t, pw, lds, r);
}

// ---- twiddles on the CONSUMER side, folded into the first layer (f32 first pass) -------------------------------------
// Decimation in frequency multiplies output j of a stage-0 butterfly by W^{t j} and output j of a stage-1 butterfly by
// W^{16 n0 j}.  Seen from the thread that READS a value: input n1 of the stage-1 thread (b, n0) carries
// W^{n0 b} (W^{16 b})^{n1}, and with the factor W^{n0 b}, common to all its inputs, moved through to its outputs (the
// transform is linear), input n0 of the stage-2 thread (b, c) carries (W^{b + 16 c})^{n0}:
//   stage 1, thread (b, n0):  input k times g1^k,  g1 = W_4096^{16 b};   stage 2, thread (b, c):  input k times g2^k,
//   g2 = W_4096^{b + 16 c};   NO multiplication on the producer side: a stage stores its butterfly outputs as they are.
// In the first layer (4-point DFTs over inputs n2, 4 + n2, 8 + n2, 12 + n2) a twiddle w = gamma (1, tau) costs two FMAs
// for u = (1, tau) x and its scale rides on the FMAs that replace the layer's additions: 36 instructions per stage on
// top of the 64 additions instead of 60 (15 complex multiplies).  gamma = cos is never exactly zero in the tables (cosl
// of a rounded pi / 2), only tiny; tau is then huge and gamma tau comes out as the sine it stands for.
// Per thread and stage 30 constants, built on the host in double (build_twiddle_row) and rounded once:
//   column 0: tau2 gamma2 tau1 tau3 rho gamma1;  columns 1..3: w0.re w0.im tau2 gamma2 tau1 tau3 rho gamma1.
// (The f64 kernel tried this in round 2 and ran 7 % slower -- profiles/NOTES.md §4.1; the f32 pass is VALU-bound.)
constexpr int kTwRow = 30;
template <typename T>
inline void build_twiddle_row(const cd *tw4096, int m, T *row) {  // g = W_4096^m
  int o = 0;
  for (int n2 = 0; n2 < 4; n2++) {
    const cd w0 = tw4096[(m * n2) & 4095], w1 = tw4096[(m * (4 + n2)) & 4095], w2 = tw4096[(m * (8 + n2)) & 4095],
             w3 = tw4096[(m * (12 + n2)) & 4095];
    if (n2 != 0) {
      row[o++] = (T)w0.x;
      row[o++] = (T)w0.y;
    }
    row[o++] = (T)(w2.y / w2.x);
    row[o++] = (T)w2.x;
    row[o++] = (T)(w1.y / w1.x);
    row[o++] = (T)(w3.y / w3.x);
    row[o++] = (T)(w3.x / w1.x);
    row[o++] = (T)w1.x;
  }
}
template <int N2, class C>
NEEDLE_HD void head_col_tw(C *a, const typename C::real *row) {
  typedef typename C::real T;
  const T *k = row + (N2 == 0 ? 0 : 6 + 8 * (N2 - 1));
  const C A0 = N2 == 0 ? a[0] : cmulf(a[N2], C{k[0], k[1]});
  const int o = N2 == 0 ? 0 : 2;
  const T tau2 = k[o], g2 = k[o + 1], tau1 = k[o + 2], tau3 = k[o + 3], rho = k[o + 4], g1 = k[o + 5];
  const C x1 = a[4 + N2], x2 = a[8 + N2], x3 = a[12 + N2];
  const C u2 = C{fmad(-tau2, x2.y, x2.x), fmad(tau2, x2.x, x2.y)};
  const C E0 = C{fmad(g2, u2.x, A0.x), fmad(g2, u2.y, A0.y)}, E1 = C{fmad(-g2, u2.x, A0.x), fmad(-g2, u2.y, A0.y)};
  const C u1 = C{fmad(-tau1, x1.y, x1.x), fmad(tau1, x1.x, x1.y)};
  const C u3 = C{fmad(-tau3, x3.y, x3.x), fmad(tau3, x3.x, x3.y)};
  const C v = C{fmad(rho, u3.x, u1.x), fmad(rho, u3.y, u1.y)}, vp = C{fmad(-rho, u3.x, u1.x), fmad(-rho, u3.y, u1.y)};
  bfly4_tail(E0, E1, v, vp, g1, a[N2], a[4 + N2], a[8 + N2], a[12 + N2]);
}
template <class C>
NEEDLE_HD void fft16_head_tw(C *a, const typename C::real *row) {
  head_col_tw<0>(a, row);
  head_col_tw<1>(a, row);
  head_col_tw<2>(a, row);
  head_col_tw<3>(a, row);
}
// First layer of stage 0 with the WINDOW folded in: the inputs are (sample of frame A, sample of frame B) still without
// it; the real window value of each input rides on the layer's FMAs (4 multiplies + 8 FMAs + 8 additions per column
// instead of 8 multiplies + 16 additions).
template <int N2, class C>
NEEDLE_HD void head_col_win(C *a, const typename C::real *w, C x0, C x1, C x2, C x3) {  // w = window at inputs n2, 4+n2, 8+n2, 12+n2
  typedef typename C::real T;
  const C A0 = C{w[0] * x0.x, w[0] * x0.y};
  const T w1 = w[1], w2 = w[2], w3 = w[3];
  const C E0 = C{fmad(w2, x2.x, A0.x), fmad(w2, x2.y, A0.y)}, E1 = C{fmad(-w2, x2.x, A0.x), fmad(-w2, x2.y, A0.y)};
  const C p3 = C{w3 * x3.x, w3 * x3.y};
  const C e2 = C{fmad(w1, x1.x, p3.x), fmad(w1, x1.y, p3.y)}, d = C{fmad(w1, x1.x, -p3.x), fmad(w1, x1.y, -p3.y)};
  const C e3 = C{d.y, -d.x};
  a[N2] = cadd(E0, e2);
  a[4 + N2] = cadd(E1, e3);
  a[8 + N2] = csub(E0, e2);
  a[12 + N2] = csub(E1, e3);
}
// the four tails of a stage, outputs stored UNMULTIPLIED to slots o + pitch j right behind each tail
template <class C>
NEEDLE_HD void dif_tails_store_raw(int o, int pitch, C *lds, C *r) {
  fft16_tail<0, NoHook, false>(r);
#pragma unroll
  for (int k2 = 0; k2 < 4; k2++) lds_put(lds, o + pitch * (0 + 4 * k2), r[0 + k2]);
  fft16_tail<1, NoHook, false>(r);
#pragma unroll
  for (int k2 = 0; k2 < 4; k2++) lds_put(lds, o + pitch * (1 + 4 * k2), r[4 + k2]);
  fft16_tail<2, NoHook, false>(r);
#pragma unroll
  for (int k2 = 0; k2 < 4; k2++) lds_put(lds, o + pitch * (2 + 4 * k2), r[8 + k2]);
  fft16_tail<3, NoHook, false>(r);
#pragma unroll
  for (int k2 = 0; k2 < 4; k2++) lds_put(lds, o + pitch * (3 + 4 * k2), r[12 + k2]);
}
// stage 1 and stage 2 in the consumer-side form (row1 / row2: this thread's 30 constants of the stage)
template <class C>
NEEDLE_HD void dif1_consumer(int t, const typename C::real *row1, C *lds, C *r) {
  const int o = dif1_base(t);
#pragma unroll
  for (int k = 0; k < 16; k++) r[k] = lds_get(lds, o + 17 * k);
  fft16_head_tw(r, row1);
  dif_tails_store_raw(o, 17, lds, r);
}
template <class C>
NEEDLE_HD void dif2_consumer(int t, const typename C::real *row2, C *lds, C *r) {
  const int o = dif2_base(t);
#pragma unroll
  for (int k = 0; k < 16; k++) r[k] = lds_get(lds, o + k);
  fft16_head_tw(r, row2);
  fft16_tail<0, NoHook, false>(r);
  fft16_tail<1, NoHook, false>(r);
  lds_put(lds, o + 12, r[3]);
  lds_put(lds, o + 13, r[7]);
  fft16_tail<2, NoHook, false>(r);
  lds_put(lds, o + 10, r[10]);
  lds_put(lds, o + 14, r[11]);
  fft16_tail<3, NoHook, false>(r);
  lds_put(lds, o + 11, r[14]);
  lds_put(lds, o + 15, r[15]);
}

// stage 2: afterwards r[out16(j)] = Z[dif_bin_of(t, j)]
template <class C> NEEDLE_HD void dif2(int t, const C *lds, C *r) {
  const int o = dif2_base(t);
#pragma unroll
  for (int k = 0; k < 16; k++) r[k] = lds_get(lds, o + k);
  fft16(r);
}

// publish the registers other threads read as partners (bins >= 2789 live in j = 10..15), in place
template <class C> NEEDLE_HD void dif2_publish(int t, C *lds, const C *r) {
  const int o = dif2_base(t);
#pragma unroll
  for (int j = 10; j < 16; j++) lds_put(lds, o + j, r[out16(j)]);
}

// stage 2 from values already in r (a caller that loads them itself), publish stores as in dif2_streamed
template <int MODE = 0, class C>
NEEDLE_HD void dif2_from_registers(int t, C *lds, C *r) {
  constexpr bool P = (MODE & kPlainFft) != 0;
  const int o = dif2_base(t);
  fft16_head(r);
  fft16_tail<0, NoHook, P>(r);
  fft16_tail<1, NoHook, P>(r);
  lds_put(lds, o + 12, r[3]);
  lds_put(lds, o + 13, r[7]);
  fft16_tail<2, NoHook, P>(r);
  lds_put(lds, o + 10, r[10]);
  lds_put(lds, o + 14, r[11]);
  fft16_tail<3, NoHook, P>(r);
  lds_put(lds, o + 11, r[14]);
  lds_put(lds, o + 15, r[15]);
}

// stage 2 with the publish stores streamed out the same way (j = 12, 13 are outputs of tails 0 and 1, j = 10, 14 of
// tail 2, j = 11, 15 of tail 3)
template <int MODE = 0, class C>
NEEDLE_HD void dif2_streamed(int t, C *lds, C *r) {
  constexpr bool P = (MODE & kPlainFft) != 0;
  const int o = dif2_base(t);
#pragma unroll
  for (int k = 0; k < 16; k++) r[k] = lds_get(lds, o + k);
  fft16_head(r);
  fft16_tail<0, NoHook, P>(r);
  fft16_tail<1, NoHook, P>(r);
  if (MODE & kStoreSpaced) {
    struct Hook2 {
      int o; C *lds; const C *r;
      NEEDLE_HD void operator()(int i) const {
        if (i == 1) lds_put(lds, o + 12, r[3]);
        if (i == 3) lds_put(lds, o + 13, r[7]);
      }
    };
    struct Hook3 {
      int o; C *lds; const C *r;
      NEEDLE_HD void operator()(int i) const {
        if (i == 1) lds_put(lds, o + 10, r[10]);
        if (i == 3) lds_put(lds, o + 14, r[11]);
      }
    };
    fft16_tail<2, Hook2, P>(r, Hook2{o, lds, r});
    fft16_tail<3, Hook3, P>(r, Hook3{o, lds, r});
    lds_put(lds, o + 11, r[14]);
    lds_put(lds, o + 15, r[15]);
  } else {
    lds_put(lds, o + 12, r[3]);
    lds_put(lds, o + 13, r[7]);
    fft16_tail<2, NoHook, P>(r);
    lds_put(lds, o + 10, r[10]);
    lds_put(lds, o + 14, r[11]);
    fft16_tail<3, NoHook, P>(r);
    lds_put(lds, o + 11, r[14]);
    lds_put(lds, o + 15, r[15]);
  }
}

// the partners Z[N - k] of this thread's bins in registers j = 0..kBinsPerThread-1, all reads issued together
template <class C> NEEDLE_HD void dif_partner_load(int t, const C *lds, C *y) {
  const int o = dif_partner_base(t);
#pragma unroll
  for (int j = 0; j < kBinsPerThread; j++) y[j] = lds_get(lds, o + 15 - j);
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

// powers of one bin for the two frames from Z[k] and its partner Z[N - k]
template <class C>
NEEDLE_HD void dif_power_of(C z, C y, typename C::real *pa, typename C::real *pb) {
  const typename C::real ar = z.x + y.x, ai = z.y - y.y;  // X_A
  const typename C::real br = z.y + y.y, bi = y.x - z.x;  // X_B
  *pa = fmad(ar, ar, ai * ai);
  *pb = fmad(br, br, bi * bi);
}
// powers of this thread's bin in register j (any bin 0 < k < 4096; bin 0 reads a slot that holds something else,
// for callers that discard it)
template <class C>
NEEDLE_HD void dif_bin_power_any(int t, int j, const C *lds, const C *r, typename C::real *pa, typename C::real *pb) {
  dif_power_of(r[out16(j)], lds_get(lds, dif_partner_base(t) + 15 - j), pa, pb);
}
// the same, j = 0..5; false if the bin is outside 10..1307
template <class C>
NEEDLE_HD bool dif_bin_power(int t, int j, const C *lds, const C *r, int *kf_out, typename C::real *pa, typename C::real *pb) {
  const int kf = dif_bin_of(t, j);
  *kf_out = kf;
  if (kf < kMinBin || kf >= kMaxBin) return false;
  dif_bin_power_any(t, j, lds, r, pa, pb);
  return true;
}

// Power image.  The PAIR (power in frame A, power in frame B) of a bin goes into one 16-byte slot of the wave that
// owns the bin (one 128-bit store per bin, and one 128-bit load gives a fold lane both frames' powers: LDS
// operations, not their bytes, are what this kernel pays for).  The slots are registers j = 0..7 of the wave's own
// stage-2 rows (pidx(16 q + j) = 17 q + j), dead once the wave has read its stage-2 inputs and disjoint from the
// partner slots j = 10..15 -- so no other wave's data is ever overwritten and the stores need no barrier.
// Inside a wave the bins of pitch class c (segment (w, c), n bins in ascending order, position q = 0..n-1) form a
// vertical strip: slot 17 (R + (q >> 2)) + C + (q & 3), a column group C in {0, 4} and consecutive rows from R.
// Fold lane l of class c (16 lanes = one DPP row per class) takes segment w = l >> 2 and positions u, u + 4, ...
// (u = l & 3): slots base + 17 i, one base and constants.  PowerLayout is built on the host (build_power_layout).
constexpr int kClassLanes = 16;      // lanes that share one pitch class in the fold (one DPP row)
constexpr int kClassLaneMax = 10;    // >= rows of the tallest strip; checked by build_power_layout
constexpr int kClassLaneMin = 4;     // <= positions of every fold lane; checked by build_power_layout
// Two constant slots (kPowerZeroSlot, kPowerTrashSlot, behind the image) make the stores and the fold's loads
// branch-free.

struct PowerLayout {
  uint16_t bin_slot[kNumBins];   // slot in the LDS image of the power pair of bin kMinBin + i
  uint32_t fold[12 * kClassLanes];  // fold thread 16 c + l: first slot | positions << 16
};
// class_of_bin[i] = pitch class of bin kMinBin + i.  Returns false if a strip does not fit (never for chromaprint's
// tables; the caller turns it into an error).
inline bool build_power_layout(const uint8_t *class_of_bin, PowerLayout *out) {
  for (int f = 0; f < 12 * kClassLanes; f++) out->fold[f] = (uint32_t)kPowerZeroSlot;
  for (int w = 0; w < 4; w++) {
    // this wave's rows in ascending order, as runs of consecutive rows
    int rows[64], nrows = 0;
    for (int b = 0; b < 16; b++)
      if (wave_of_k0(b) == w)
        for (int c = 0; c < 16; c++) rows[nrows++] = 16 * b + c;
    int next_row[2] = {0, 0};  // per column group: index into rows[] of the first free row
    for (int c = 0; c < 12; c++) {
      int bins[kNumBins], n = 0;
      for (int k = kMinBin; k < kMaxBin; k++)
        if (class_of_bin[k - kMinBin] == c && wave_of_k0(k & 15) == w) bins[n++] = k;
      const int h = (n + 3) / 4;
      if (h > kClassLaneMax) return false;
      // first fit: a column group with h consecutive rows left inside one run
      int R = -1, C = 0;
      for (int cg = 0; cg < 2 && R < 0; cg++) {
        int i = next_row[cg];
        while (i + h <= nrows) {
          bool run = true;
          for (int d = 1; d < h; d++) run = run && rows[i + d] == rows[i] + d;
          if (run) break;
          i++;  // skip to the next run
        }
        if (i + h <= nrows) {
          R = rows[i];
          C = 4 * cg;
          next_row[cg] = i + h;
        }
      }
      if (R < 0) return false;
      for (int q = 0; q < n; q++)
        out->bin_slot[bins[q] - kMinBin] = (uint16_t)(17 * (R + (q >> 2)) + C + (q & 3));
      for (int u = 0; u < 4; u++) {
        const int count = n > u ? (n - u + 3) / 4 : 0;
        if (count < kClassLaneMin) return false;
        out->fold[16 * c + 4 * w + u] = (uint32_t)(17 * R + C + u) | ((uint32_t)count << 16);
      }
    }
  }
  return true;
}

// f32 first pass only: every thread also leaves the sum of squares of its 16 (windowed) samples of the two frames in
// the spare column 8 of its own stage-2 row, and the fourth wave -- which owns no pitch class -- folds the 256 partials
// like a class: lane l takes rows 4 l .. 4 l + 3, the DPP tree sums each of its four rows of 16 lanes, so a frame's
// energy arrives as four partial sums.
NEEDLE_HD int energy_slot(int t) { return 17 * (16 * thread_k0(t) + (t & 15)) + 8; }
NEEDLE_HD uint32_t energy_fold_entry(int l) { return (uint32_t)(17 * 4 * l + 8) | (4u << 16); }  // l = 0..63

// One fold lane's share of its class: `count` positions from slot `base`, 17 slots apart, both frames at once (x =
// frame A, y = frame B); reads beyond the count fetch the zero slot.  Split into the loads and the sums so the kernel
// can put other work between them.
template <int FROM, int TO, class C>
NEEDLE_HD void class_lane_load_part(const C *lds, uint32_t fold_entry, C *v) {  // v[i - FROM], i = FROM..TO-1
  const int base = (int)(fold_entry & 0xffffu), count = (int)(fold_entry >> 16);
#pragma unroll
  for (int i = FROM; i < TO; i++)
    v[i - FROM] = lds_get(lds, (i < kClassLaneMin || i < count) ? base + 17 * i : kPowerZeroSlot);
}
template <class C> NEEDLE_HD void class_lane_load(const C *lds, uint32_t fold_entry, C *v) {
  class_lane_load_part<0, kClassLaneMax>(lds, fold_entry, v);
}
template <class C> NEEDLE_HD C class_lane_add(const C *v) {
  C acc = v[0];
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
// ODD > 0: the window's rows are stored de-interleaved -- even rows PITCH apart from w, odd rows PITCH apart from w + ODD
// (the certification kernel at step 2: neighbouring lanes' windows then start one row apart in each half, and a wave's
// accesses to one cell spread over all LDS banks; row-major with two rows between lanes they hit every other bank twice).
template <int R, int C, int PITCH, int ODD = 0>
struct WindowStep {
  static NEEDLE_HD void run(const double *w, double *a, double *b) {
    CellStep<R, C, 0>::run(w[ODD ? (R >> 1) * PITCH + (R & 1) * ODD + C : R * PITCH + C], a, b);
#if defined(__HIP_DEVICE_COMPILE__)
    // The 32 sums are pinned at the end of every row (an empty asm that "uses" them): without this the compiler loads all
    // 192 cells of the window first and adds afterwards -- 384 registers for the cells alone, 376 VGPRs, one wave per
    // SIMD.  The sums themselves, and their order, are unchanged.
    if (C == 11) {
      asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),
                        "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]));
      asm volatile("" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]),
                        "+v"(b[8]), "+v"(b[9]), "+v"(b[10]), "+v"(b[11]), "+v"(b[12]), "+v"(b[13]), "+v"(b[14]), "+v"(b[15]));
    }
#endif
    WindowStep<(C == 11) ? R + 1 : R, (C == 11) ? 0 : C + 1, PITCH, ODD>::run(w, a, b);
  }
};
template <int PITCH, int ODD>
struct WindowStep<16, 0, PITCH, ODD> {
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
