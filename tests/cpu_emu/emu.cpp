// TEST FIXTURE — steps the per-thread code of the fingerprint / search kernels (needle_amd/csrc/fp_core.h,
// the loop body of search.hip) serially on the CPU, phase by phase with the barriers where the kernel has
// them, so indexing mistakes are caught in the CPU test-suite before any GPU time is spent.
// It is built by tests/conftest.py with g++ into tests/cpu_emu/libemu.so and is not part of, nor
// reachable from, libneedle_capi.so.
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../needle_amd/csrc/fp_core.h"

using namespace needle::core;

namespace {
struct Tables {
  std::vector<cd> tw;
  std::vector<double> wcos;  // recurrence seeds, as the kernel's table
  std::vector<float> win32;  // f32 pass: the window itself (times 1/32767 and fp_core.h's 1/2), correctly rounded
  WindowConst wconst;
  PowerLayout layout;  // where each bin's power pair goes, what each fold lane reads
  ClassifierThresholds thr;
};

const double kThr[16][3] = {
    {1.98215, 2.35817, 2.63523},          {-1.03809, -0.651211, -0.282167},  {-0.298702, 0.119262, 0.558497},
    {-0.105439, 0.0153946, 0.135898},     {-0.142891, 0.0258736, 0.200632},  {-0.826319, -0.590612, -0.368214},
    {-0.557409, -0.233035, 0.0534525},    {-0.0646826, 0.00620476, 0.0784847}, {-0.192387, -0.029699, 0.215855},
    {-0.0397818, -0.00568076, 0.0292026}, {-0.53823, -0.369934, -0.190235},  {-0.124877, 0.0296483, 0.139239},
    {-0.101475, 0.0225617, 0.231971},     {-0.0799915, -0.00729616, 0.063262}, {-0.272556, 0.019424, 0.302559},
    {-0.164292, -0.0321188, 0.0846339},
};

const Tables &tables() {
  static Tables t;
  if (!t.tw.empty()) return t;
  t.tw.resize(4096);
  for (int k = 0; k < 4096; k++) {
    long double a = -2.0L * 3.14159265358979323846264338327950288L * k / 4096.0L;
    t.tw[k] = cd{(double)cosl(a), (double)sinl(a)};
  }
  const long double theta = 2.0L * 3.14159265358979323846264338327950288L / 4095.0L;
  t.wcos.resize(512);
  for (int i = 0; i < 512; i++) t.wcos[i] = (double)cosl(theta * (long double)(i - 256));
  t.wconst.k2 = (double)(2.0L * cosl(256.0L * theta));
  t.win32.resize(4096);
  for (int i = 0; i < 4096; i++)
    t.win32[i] = (float)((long double)kPairInputScale * (0.54L - 0.46L * cosl(theta * (long double)i)) / 32767.0L);
  t.wconst.a = kPairInputScale * (0.54 / 32767.0);
  t.wconst.b = kPairInputScale * (0.46 / 32767.0);
  std::vector<uint8_t> class_of_bin(kNumBins);
  for (int i = kMinBin; i < kMaxBin; i++) {
    double freq = (double)i * 11025 / 4096;
    double octave = std::log(freq / (440.0 / 16.0)) / std::log(2.0);
    double note = 12 * (octave - std::floor(octave));
    class_of_bin[i - kMinBin] = (uint8_t)(int)(signed char)note;
  }
  if (!build_power_layout(class_of_bin.data(), &t.layout)) std::abort();
  for (int i = 0; i < 16; i++)
    for (int j = 0; j < 3; j++) t.thr.e[i][j] = std::exp(kThr[i][j]);
  return t;
}
}  // namespace

// stft_chroma_kernel (C = cd) / stft_chroma32_kernel (C = cf), one frame pair (frame B may be NULL): 256 emulated
// threads, phases separated exactly where the kernels have their barriers.  The two kernels share the schedule, the
// LDS image and the power layout; they differ in the arithmetic width, in where the window and the twiddle powers come
// from (f64: recurrence and running products; f32: correctly rounded tables) and in the energy partials the f32 pass
// adds (slot 8 of every stage-2 row, folded by the fourth wave into four sums per frame).
template <class C>
static void emu_pair(const int16_t *fa, const int16_t *fb, int channels, double *chroma_a, double *chroma_b,
                     float *energy_a, float *energy_b) {
  typedef typename C::real T;
  constexpr bool F32 = sizeof(T) == 4;
  const Tables &Tb = tables();
  std::vector<C> lds(kLds2Slots);
  lds[kPowerZeroSlot] = C{(T)0, (T)0};  // the kernel's constant zero (a pad slot)
  std::vector<C> regs(256 * 16);
  std::vector<C> esum(256);  // f32 pass: this thread's sum of squares of its 16 samples of frame A / frame B
  auto sample = [&](const int16_t *src, int n) -> int {
    if (!src) return 0;
    if (channels == 1) return src[n];
    return ((int)src[2 * n] + (int)src[2 * n + 1]) / 2;
  };
  auto twp = [&](int e) { return C{(T)Tb.tw[e & 4095].x, (T)Tb.tw[e & 4095].y}; };  // correctly rounded W^e
  for (int t = 0; t < 256; t++) {
    double c = Tb.wcos[t + 256], c_prev = Tb.wcos[t];
    C e{(T)0, (T)0};
    for (int k = 0; k < 16; k++) {
      const int n = t + 256 * k;
      if (F32) {
        const T w = (T)Tb.win32[n];
        const C x{(T)sample(fa, n) * w, (T)sample(fb, n) * w};
        regs[t * 16 + k] = x;
        e = C{fmad(x.x, x.x, e.x), fmad(x.y, x.y, e.y)};
      } else {
        const double w = window_step(Tb.wconst, &c, &c_prev);
        regs[t * 16 + k] = C{(T)((double)sample(fa, n) * w), (T)((double)sample(fb, n) * w)};
      }
    }
    esum[t] = e;
  }
  for (int t = 0; t < 256; t++) {
    if (F32) {
      C pw[16];
      for (int j = 1; j < 16; j++) pw[j] = twp(t * j);
      dif0_streamed_pw(t, pw, lds.data(), &regs[t * 16]);
    } else {
      dif0_streamed(t, twp(t), lds.data(), &regs[t * 16]);
    }
  }
  // ---- workgroup barrier (stage-0 stores -> stage-1 reads).  From here to the next barrier (power image complete)
  // the kernel has only wave-level ordering: a wave may run through ALL of stage 1, stage 2, publish, partner reads
  // and power stores while another has not started stage 1.  The waves are therefore run to completion one after the
  // other, in both orders, and inside a wave the 16-lane groups run stage 1 -> stage 2 one group ahead of the next
  // (the order a wave is allowed to take without a fence between them); the two images must be identical.
  const std::vector<C> after_stage0 = lds;
  const std::vector<C> regs0 = regs;
  std::vector<C> image[2];
  int seen_total = 0;
  for (int dir = 0; dir < 2; dir++) {
    lds = after_stage0;
    regs = regs0;
    int seen = 0;
    for (int wi = 0; wi < 4; wi++) {
      const int w = dir ? 3 - wi : wi;
      for (int grp = 3; grp >= 0; grp--) {
        const int t0 = 64 * w + 16 * grp;
        for (int t = t0; t < t0 + 16; t++) {
          if (F32) {
            C pw[16];
            for (int j = 1; j < 16; j++) pw[j] = twp(16 * (t & 15) * j);
            dif1_streamed_pw(t, pw, lds.data(), &regs[t * 16]);
          } else {
            dif1_streamed(t, twp(16 * (t & 15)), lds.data(), &regs[t * 16]);
          }
        }
        for (int t = t0; t < t0 + 16; t++) dif2_streamed(t, lds.data(), &regs[t * 16]);
      }
      // wave fence; partner reads and power stores interleaved thread by thread inside the wave, both directions
      for (int i = 0; i < 64; i++) {
        const int t = 64 * w + (dir ? 63 - i : i);
        C y[kBinsPerThread];
        dif_partner_load(t, lds.data(), y);  // as the kernel: the six partner reads first, then the six stores
        for (int j = 0; j < kBinsPerThread; j++) {
          const int kf = dif_bin_of(t, j);
          if (kf < kMinBin || kf >= kMaxBin) {
            lds[kPowerTrashSlot] = C{(T)1e30, (T)1e30};  // what the kernel does with them: a slot nobody may read
            continue;
          }
          if (dif_partner_base(t) + 15 - j != pidx(dif_slot_of_bin(kFft2N - kf))) { chroma_a[0] = -3.0; return; }
          // the partner must have been published by THIS wave, and the power slot must be one of this wave's rows
          if (wave_of_k0((kFft2N - kf) & 15) != w) { chroma_a[0] = -4.0; return; }
          const int slot = Tb.layout.bin_slot[kf - kMinBin];
          const int row = slot / 17, col = slot % 17;
          if (wave_of_k0(row >> 4) != w || col > 7) { chroma_a[0] = -5.0; return; }
          // the kernel keeps the slots packed two to a register, already scaled to bytes
          if (slot_bytes<0>(pack_slots((uint32_t)slot, 4351u)) != (uint32_t)slot * 16u ||
              slot_bytes<1>(pack_slots(4351u, (uint32_t)slot)) != (uint32_t)slot * 16u) { chroma_a[0] = -6.0; return; }
          T a, b;
          dif_power_of(regs[t * 16 + out16(j)], y[j], &a, &b);
          lds_put_bytes(lds.data(), slot_bytes<0>(pack_slots((uint32_t)slot, 4351u)), C{a, b});
          seen++;
        }
        if (F32) {  // energy partial: spare column 8 of the thread's own stage-2 row
          const int slot = energy_slot(t);
          if (slot % 17 != 8 || wave_of_k0((slot / 17) >> 4) != w) { chroma_a[0] = -7.0; return; }
          lds[slot] = esum[t];
        }
      }
    }
    if (seen != kNumBins) { chroma_a[0] = -1.0; return; }  // every bin must be owned by exactly one (t, j)
    seen_total += seen;
    image[dir] = lds;
  }
  for (int i = 0; i < kNumBins; i++) {
    const int slot = Tb.layout.bin_slot[i];
    const C u = image[0][slot], v = image[1][slot];
    if (u.x != v.x || u.y != v.y) { chroma_a[0] = -2.0; return; }
  }
  // ---- workgroup barrier (power image complete -> fold reads)
  const int rows = F32 ? 16 : 12;  // DPP rows that fold: 12 classes (+ the fourth wave's four energy rows)
  for (int c = 0; c < rows; c++) {
    C lane[kClassLanes];
    for (int l = 0; l < kClassLanes; l++) {
      C v[kClassLaneMax];
      class_lane_load(lds.data(), c < 12 ? Tb.layout.fold[16 * c + l] : energy_fold_entry(16 * (c - 12) + l), v);
      lane[l] = class_lane_add(v);
    }
    for (int step = 0; step < 4; step++) {
      C nxt[kClassLanes];
      for (int l = 0; l < kClassLanes; l++) nxt[l] = cadd(lane[l], lane[class_tree_partner(l, step)]);
      std::memcpy(lane, nxt, sizeof(lane));
    }
    if (c < 12) {
      chroma_a[c] = (double)lane[0].x;
      if (chroma_b) chroma_b[c] = (double)lane[0].y;
    } else {
      // both frames carry the energy of the PAIR (stft32_kernel.h: one transform, one noise floor)
      const float e = fb ? (float)lane[0].x + (float)lane[0].y : (float)lane[0].x;
      if (energy_a) energy_a[c - 12] = e;
      if (energy_b) energy_b[c - 12] = fb ? e : 0.0f;
    }
  }
}

extern "C" {

void emu_stft_chroma_pair(const int16_t *fa, const int16_t *fb, int channels, double *chroma_a, double *chroma_b) {
  emu_pair<cd>(fa, fb, channels, chroma_a, chroma_b, nullptr, nullptr);
}
// the f32 first pass; energy_[ab][4]: the four partial sums of sum x^2 (x = sample * window / 2) over the frame
void emu_stft_chroma_pair_f32(const int16_t *fa, const int16_t *fb, int channels, double *chroma_a, double *chroma_b,
                              float *energy_a, float *energy_b) {
  emu_pair<cf>(fa, fb, channels, chroma_a, chroma_b, energy_a, energy_b);
}

// a whole mono stream through the f32 pass (f32 = 1) or the f64 kernel's arithmetic (f32 = 0): chroma [frames][12],
// energy [frames][4] (f32 pass only; may be NULL).  For calibrating the certification radius on the CPU.
void emu_stft_chroma_stream(const int16_t *pcm, size_t n, int f32, double *chroma, float *energy) {
  const size_t frames = n < 4096 ? 0 : (n - 4096) / 1365 + 1;
  for (size_t f = 0; f < frames; f += 2) {
    const bool has_b = f + 1 < frames;
    const int16_t *fa = pcm + f * 1365, *fb = has_b ? fa + 1365 : nullptr;
    double dummy[12];
    float edummy[4];
    if (f32)
      emu_pair<cf>(fa, fb, 1, chroma + f * 12, has_b ? chroma + (f + 1) * 12 : dummy, energy ? energy + f * 4 : edummy,
                   energy && has_b ? energy + (f + 1) * 4 : edummy);
    else
      emu_pair<cd>(fa, fb, 1, chroma + f * 12, has_b ? chroma + (f + 1) * 12 : dummy, nullptr, nullptr);
  }
}

// The STFT kernel's power layout (fp_core.h build_power_layout): every bin 10..1307 has a slot of its own among the dead
// slots (columns 0..7) of a row owned by the wave that owns the bin; the 192 fold lanes together read every bin's slot
// exactly once and nothing else but the zero slot; the two packed forms round-trip.  Returns 0, or a negative code.
int emu_power_layout_check() {
  const Tables &T = tables();
  std::vector<int> owner(kLds2Slots, -1);
  for (int k = kMinBin; k < kMaxBin; k++) {
    const int slot = T.layout.bin_slot[k - kMinBin];
    if (slot < 0 || slot >= kFftSlots) return -1;
    const int row = slot / 17, col = slot % 17;
    if (col > 7) return -2;                                  // columns 8, 9 are spare, 10..15 the partner values, 16 the pad
    if (wave_of_k0(row >> 4) != wave_of_k0(k & 15)) return -3;  // another wave's row
    if (owner[slot] != -1) return -4;                        // two bins in one slot
    owner[slot] = k;
  }
  std::vector<int> reads(kLds2Slots, 0);
  for (int c = 0; c < 12; c++)
    for (int l = 0; l < kClassLanes; l++) {
      const uint32_t e = T.layout.fold[16 * c + l];
      const int base = (int)(e & 0xffffu), count = (int)(e >> 16);
      if (count < kClassLaneMin || count > kClassLaneMax) return -5;
      for (int i = 0; i < count; i++) reads[base + 17 * i]++;
    }
  for (int s2 = 0; s2 < kLds2Slots; s2++)
    if (reads[s2] != (owner[s2] != -1 ? 1 : 0)) return -6;    // a bin not folded, folded twice, or a foreign slot read
  // every thread's six bins: inside the range iff the thread's wave owns them; threads of all four waves together own
  // each bin exactly once
  std::vector<int> seen(kNumBins, 0);
  for (int t = 0; t < kThreads; t++)
    for (int j = 0; j < kBinsPerThread; j++) {
      const int kf = dif_bin_of(t, j);
      if (kf >= kMinBin && kf < kMaxBin) {
        if (wave_of_k0(kf & 15) != (t >> 6)) return -7;
        seen[kf - kMinBin]++;
      }
    }
  for (int i = 0; i < kNumBins; i++)
    if (seen[i] != 1) return -8;
  // the sixteen lane groups are a permutation of the low digits, partners in the same wave
  int digits = 0;
  for (int g = 0; g < 16; g++) {
    digits |= 1 << group_k0(g);
    if (wave_of_k0(group_k0(g)) != (g >> 2) || wave_of_k0((16 - group_k0(g)) & 15) != (g >> 2)) return -9;
  }
  return digits == 0xffff ? 0 : -10;
}

// classify_kernel, one item: 16 feature rows in, raw u32 out
uint32_t emu_classify(const double *window16x12) { return classify_window(window16x12, &tables().thr); }

// hamming_runs_kernel, one problem: every diagonal walked exactly as a lane does it
struct EmuRun {
  uint32_t src_end, dst_end, len;
};
size_t emu_hamming_runs(const uint32_t *s, int n, const uint32_t *t, int m, uint32_t threshold, uint32_t min_len,
                        EmuRun *out, size_t cap) {
  size_t count = 0;
  if (n < 2 || m < 2) return 0;
  const int num_diags = n + m - 3;
  for (int dd = 0; dd < num_diags; dd++) {
    const int d = dd - (n - 2);
    const int i_lo = d < 0 ? 1 - d : 1;
    const int i_hi = (n - 1) < (m - 1 - d) ? (n - 1) : (m - 1 - d);
    uint32_t run = 0;
    for (int i = i_lo; i <= i_hi; i++) {
      const bool match = (uint32_t)__builtin_popcount(s[i] ^ t[i + d]) <= threshold;
      if (match) {
        run++;
      } else {
        if (run >= min_len) {
          if (count < cap) out[count] = EmuRun{(uint32_t)(i - 1), (uint32_t)(i - 1 + d), run};
          count++;
        }
        run = 0;
      }
    }
    if (run >= min_len) {
      if (count < cap) out[count] = EmuRun{(uint32_t)i_hi, (uint32_t)(i_hi + d), run};
      count++;
    }
  }
  return count;
}
}
