// TEST FIXTURE — steps the per-thread code of the fingerprint / search kernels (needle_amd/csrc/fp_core.h,
// the loop body of search.hip) serially on the CPU, phase by phase with the barriers where the kernel has
// them, so indexing mistakes are caught in the CPU test-suite before any GPU time is spent.
// It is built by tests/conftest.py with g++ into tests/cpu_emu/libemu.so and is not part of, nor
// reachable from, libneedle_capi.so.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../needle_amd/csrc/fp_core.h"

using namespace needle::core;

namespace {
struct Tables {
  std::vector<cd> tw;
  std::vector<double> wcos;  // recurrence seeds, as the kernel's table
  WindowConst wconst;
  std::vector<uint16_t> class_bins;
  uint32_t class_start[13];
  std::vector<uint16_t> bin_pos;  // bin - kMinBin -> position in the class-sorted list
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
  t.wconst.a = kPairInputScale * (0.54 / 32767.0);
  t.wconst.b = kPairInputScale * (0.46 / 32767.0);
  std::vector<std::vector<uint16_t>> by(12);
  for (int i = kMinBin; i < kMaxBin; i++) {
    double freq = (double)i * 11025 / 4096;
    double octave = std::log(freq / (440.0 / 16.0)) / std::log(2.0);
    double note = 12 * (octave - std::floor(octave));
    by[(int)(signed char)note].push_back((uint16_t)i);
  }
  for (int c = 0; c < 12; c++) {
    t.class_start[c] = (uint32_t)t.class_bins.size();
    t.class_bins.insert(t.class_bins.end(), by[c].begin(), by[c].end());
  }
  t.class_start[12] = (uint32_t)t.class_bins.size();
  t.bin_pos.resize(kNumBins);
  for (size_t pos = 0; pos < t.class_bins.size(); pos++) t.bin_pos[t.class_bins[pos] - kMinBin] = (uint16_t)pos;
  for (int i = 0; i < 16; i++)
    for (int j = 0; j < 3; j++) t.thr.e[i][j] = std::exp(kThr[i][j]);
  return t;
}
}  // namespace

extern "C" {

// stft_chroma_kernel, one frame pair (frame B may be NULL): 256 emulated threads, phases separated exactly
// where the kernel has its barriers
void emu_stft_chroma_pair(const int16_t *fa, const int16_t *fb, int channels, double *chroma_a, double *chroma_b) {
  const Tables &T = tables();
  std::vector<cd> lds(kLds2Slots);
  lds[kPowerZeroSlot] = cd{0.0, 0.0};  // the kernel's constant zero (a pad slot)
  std::vector<cd> regs(256 * 16);
  auto sample = [&](const int16_t *src, int n) -> int {
    if (!src) return 0;
    if (channels == 1) return src[n];
    return ((int)src[2 * n] + (int)src[2 * n + 1]) / 2;
  };
  for (int t = 0; t < 256; t++) {
    double c = T.wcos[t + 256], c_prev = T.wcos[t];
    for (int k = 0; k < 16; k++) {
      const int n = t + 256 * k;
      const double w = window_step(T.wconst, &c, &c_prev);
      regs[t * 16 + k] = cd{(double)sample(fa, n) * w, (double)sample(fb, n) * w};
    }
  }
  for (int t = 0; t < 256; t++) dif0(t, T.tw[t], lds.data(), &regs[t * 16]);
  // stage 1 -> stage 2 -> publish run group by group (16 consecutive lanes), the other groups still untouched:
  // this is the order the kernel is allowed to take without a workgroup barrier between these phases
  for (int grp = 15; grp >= 0; grp--) {
    for (int t = 16 * grp; t < 16 * grp + 16; t++) dif1(t, T.tw[16 * (t & 15)], lds.data(), &regs[t * 16]);
    for (int t = 16 * grp; t < 16 * grp + 16; t++) {
      dif2(t, lds.data(), &regs[t * 16]);
      dif2_publish(t, lds.data(), &regs[t * 16]);
    }
  }
  // partner reads and power stores interleaved thread by thread (no barrier between them in the kernel); walking
  // the threads in both directions must give the same image if the stores never touch a live partner slot
  std::vector<cd> snapshot = lds;
  std::vector<cd> image[2];
  for (int dir = 0; dir < 2; dir++) {
    lds = snapshot;
    int seen = 0;
    for (int i = 0; i < 256; i++) {
      const int t = dir ? 255 - i : i;
      for (int j = 0; j < 6; j++) {
        int kf;
        double a, b;
        if (dif_bin_power(t, j, lds.data(), &regs[t * 16], &kf, &a, &b)) {
          lds[dif_power_slot(T.bin_pos[kf - kMinBin])] = cd{a, b};
          seen++;
        }
      }
    }
    if (seen != kNumBins) { chroma_a[0] = -1.0; return; }  // every bin must be owned by exactly one (t, j)
    image[dir] = lds;
  }
  for (int p = 0; p < kNumBins; p++) {
    const cd u = image[0][dif_power_slot(p)], v = image[1][dif_power_slot(p)];
    if (u.x != v.x || u.y != v.y) { chroma_a[0] = -2.0; return; }
  }
  for (int c = 0; c < 12; c++) {
    cd lane[kClassLanes];
    for (int l = 0; l < kClassLanes; l++) {
      cd v[kClassLaneMax];
      class_lane_load(lds.data(), (int)T.class_start[c], (int)T.class_start[c + 1], l, v);
      lane[l] = class_lane_add(v);
    }
    for (int step = 0; step < 4; step++) {
      cd nxt[kClassLanes];
      for (int l = 0; l < kClassLanes; l++) nxt[l] = cadd(lane[l], lane[class_tree_partner(l, step)]);
      std::memcpy(lane, nxt, sizeof(lane));
    }
    chroma_a[c] = lane[0].x;
    if (chroma_b) chroma_b[c] = lane[0].y;
  }
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
