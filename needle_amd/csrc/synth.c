/*
 * Deterministic synthetic episode audio for tests and bench.py (harness tool, not part of the
 * analyze/search path; built as its own libneedle_synth.so).
 *
 * BASELINE.json's configs are "synthetic PCM": mono s16 generated directly at chromaprint's
 * 11025 Hz so no resampler sits in front of the path (the reference resamples with FFmpeg first,
 * needle/src/audio/analyzer.rs:179-187 — out of scope here).  Content is tonal (note/chord
 * sequences inside chromaprint's 28..3520 Hz chroma band) because white noise yields no stable
 * chroma and hence no Hamming<=10 runs.
 *
 * Bit-reproducible everywhere: no libm.  Oscillators are u32 phase accumulators, sine is a fixed
 * odd Taylor polynomial evaluated in double with a fixed operation order (-ffp-contract=off),
 * randomness is splitmix64.
 */
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#define RATE 11025

static uint64_t splitmix64(uint64_t *s) {
  uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

static double urand(uint64_t *s) { return (double)(splitmix64(s) >> 11) * (1.0 / 9007199254740992.0); }

/* sin(2*pi*phase/2^32), |error| < 4e-6 */
static double psin(uint32_t phase) {
  const double PI = 3.14159265358979323846;
  double x = (double)(int32_t)phase * (PI / 2147483648.0);
  if (x > 0.5 * PI) x = PI - x;
  if (x < -0.5 * PI) x = -PI - x;
  double x2 = x * x;
  double p = 1.0 / 362880.0;
  p = p * x2 - 1.0 / 5040.0;
  p = p * x2 + 1.0 / 120.0;
  p = p * x2 - 1.0 / 6.0;
  p = p * x2 + 1.0;
  return x * p;
}

/* equal-tempered semitone table 110 Hz * 2^(k/12), k = 0..48 (110..1760 Hz), by repeated multiply */
static void scale_table(double f[49]) {
  const double semi = 1.0594630943592953; /* 2^(1/12) */
  f[0] = 110.0;
  for (int k = 1; k < 49; k++) f[k] = f[k - 1] * semi;
  f[12] = 220.0;
  f[24] = 440.0;
  f[36] = 880.0;
  f[48] = 1760.0;
}

/*
 * Render n samples of a seeded note/chord sequence into acc (double, full scale = 1.0), adding to
 * what is there.  Each note lasts 0.37..1.5 s, has 3 or 4 partials drawn from the scale, a 20 ms
 * raised-cosine attack/release and a peak of about `gain`.
 */
static void render_tonal(uint64_t seed, double *acc, size_t n, double gain) {
  double scale[49];
  scale_table(scale);
  uint64_t s = seed ^ 0xA5A5A5A5DEADBEEFull;
  const size_t ramp = (size_t)(0.020 * RATE);
  size_t pos = 0;
  while (pos < n) {
    size_t len = (size_t)((0.37 + urand(&s) * (1.5 - 0.37)) * RATE);
    int np = 3 + (int)(splitmix64(&s) & 1);
    uint32_t inc[4], ph[4];
    double amp[4], total = 0.0;
    for (int k = 0; k < np; k++) {
      int note = (int)(splitmix64(&s) % 49);
      inc[k] = (uint32_t)(scale[note] * (4294967296.0 / RATE));
      ph[k] = (uint32_t)splitmix64(&s);
      amp[k] = 0.5 + 0.5 * urand(&s);
      total += amp[k];
    }
    for (int k = 0; k < np; k++) amp[k] = amp[k] * (gain / total);
    for (size_t i = 0; i < len && pos + i < n; i++) {
      double env = 1.0;
      if (i < ramp) {
        /* 0.5 - 0.5 cos(pi i / ramp) = 0.5 + 0.5 sin(pi i/ramp - pi/2) */
        uint32_t p = (uint32_t)(((uint64_t)i << 31) / ramp) - 0x40000000u;
        env = 0.5 + 0.5 * psin(p);
      } else if (len - i <= ramp) {
        uint32_t p = (uint32_t)(((uint64_t)(len - 1 - i) << 31) / ramp) - 0x40000000u;
        env = 0.5 + 0.5 * psin(p);
      }
      double v = 0.0;
      for (int k = 0; k < np; k++) {
        v += amp[k] * psin(ph[k]);
        ph[k] += inc[k];
      }
      acc[pos + i] += env * v;
    }
    pos += len;
  }
}

static void add_noise(uint64_t seed, double *acc, size_t n, double level) {
  uint64_t s = seed ^ 0x0123456789ABCDEFull;
  for (size_t i = 0; i < n; i++) acc[i] += level * (2.0 * urand(&s) - 1.0);
}

/*
 * One episode: unique tonal body (seed), optional shared intro / outro segments (seeded separately
 * so every episode that uses the same segment seed gets bit-identical segment content) overwriting
 * [intro_off, intro_off+intro_len) and [outro_off, outro_off+outro_len), then episode-specific
 * white noise at -40 dB over everything.  `scratch` must hold total doubles.
 */
void needle_synth_episode(uint64_t seed, size_t total, uint64_t intro_seed, size_t intro_off,
                          size_t intro_len, uint64_t outro_seed, size_t outro_off, size_t outro_len,
                          double *scratch, int16_t *out) {
  memset(scratch, 0, total * sizeof(double));
  render_tonal(seed, scratch, total, 0.5);
  if (intro_len && intro_off + intro_len <= total) {
    memset(scratch + intro_off, 0, intro_len * sizeof(double));
    render_tonal(intro_seed, scratch + intro_off, intro_len, 0.5);
  }
  if (outro_len && outro_off + outro_len <= total) {
    memset(scratch + outro_off, 0, outro_len * sizeof(double));
    render_tonal(outro_seed, scratch + outro_off, outro_len, 0.5);
  }
  add_noise(seed, scratch, total, 0.01);
  for (size_t i = 0; i < total; i++) {
    double v = scratch[i] * 32767.0;
    if (v > 32767.0) v = 32767.0;
    if (v < -32768.0) v = -32768.0;
    /* round half away from zero, no libm */
    out[i] = (int16_t)(v >= 0.0 ? (int32_t)(v + 0.5) : -(int32_t)(0.5 - v));
  }
}
