/*
 * ORACLE — test infrastructure only (see ora_chromaprint.h for the full header and the
 * "parity unpinned" statement).  Restates chromaprint 1.5.x's default fingerprinter, the third-party
 * code behind needle/src/audio/analyzer.rs:176-300 (Context::default/start/feed/finish/
 * get_fingerprint_raw).  Plain C, double precision, deliberately simple: the stages are written as
 * chromaprint structures them (AudioProcessor -> FFT -> Chroma -> ChromaFilter -> ChromaNormalizer ->
 * FingerprintCalculator), one frame at a time, sequential summation everywhere.
 */
#include "ora_chromaprint.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

#define MIN_FREQ 28
#define MAX_FREQ 3520
#define FFT_N ORA_FRAME_SIZE

int ora_chromaprint_delay_ms(void) {
  /* FingerprinterConfiguration::delay(): ((taps-1)+(max_filter_width-1))*hop + overlap samples,
   * reported in integer ms by chromaprint_get_delay_ms. */
  int overlap = ORA_FRAME_SIZE - ORA_FRAME_HOP;
  int delay = ((ORA_FIR_TAPS - 1) + (ORA_MAX_FILTER_WIDTH - 1)) * ORA_FRAME_HOP + overlap;
  return (int)(1000.0 * delay / ORA_SAMPLE_RATE);
}

int ora_chromaprint_item_duration_ms(void) {
  return (int)(1000.0 * ORA_FRAME_HOP / ORA_SAMPLE_RATE);
}

size_t ora_chromaprint_num_frames(size_t s) {
  /* AudioSlicer: only complete frames; nothing is padded at finish(). */
  return s < ORA_FRAME_SIZE ? 0 : (s - ORA_FRAME_SIZE) / ORA_FRAME_HOP + 1;
}

size_t ora_chromaprint_num_items(size_t s) {
  size_t f = ora_chromaprint_num_frames(s);
  size_t lat = (ORA_FIR_TAPS - 1) + (ORA_MAX_FILTER_WIDTH - 1);
  return f > lat ? f - lat : 0;
}

/* ---- classifier table (chromaprint fingerprinter_configuration.cpp, kClassifiersTest2) ---------- */
typedef struct {
  int type, y, height, width;
  double t0, t1, t2;
} classifier_t;

static const classifier_t kClassifiers[16] = {
    {0, 4, 3, 15, 1.98215, 2.35817, 2.63523},
    {4, 4, 6, 15, -1.03809, -0.651211, -0.282167},
    {1, 0, 4, 16, -0.298702, 0.119262, 0.558497},
    {3, 8, 2, 12, -0.105439, 0.0153946, 0.135898},
    {3, 4, 4, 8, -0.142891, 0.0258736, 0.200632},
    {4, 0, 3, 5, -0.826319, -0.590612, -0.368214},
    {1, 2, 2, 9, -0.557409, -0.233035, 0.0534525},
    {2, 7, 3, 4, -0.0646826, 0.00620476, 0.0784847},
    {2, 6, 2, 16, -0.192387, -0.029699, 0.215855},
    {2, 1, 3, 2, -0.0397818, -0.00568076, 0.0292026},
    {5, 10, 1, 15, -0.53823, -0.369934, -0.190235},
    {3, 6, 2, 10, -0.124877, 0.0296483, 0.139239},
    {2, 1, 1, 14, -0.101475, 0.0225617, 0.231971},
    {3, 5, 6, 4, -0.0799915, -0.00729616, 0.063262},
    {1, 9, 2, 12, -0.272556, 0.019424, 0.302559},
    {3, 4, 2, 14, -0.164292, -0.0321188, 0.0846339},
};

static const unsigned char kGray[4] = {0, 1, 3, 2};
static const double kFir[ORA_FIR_TAPS] = {0.25, 0.75, 1.0, 0.75, 0.25};

/* ---- FFT: complex double, two real frames per 4096-point transform ----------------------------- */
typedef struct {
  double re, im;
} cpx;

typedef struct {
  cpx tw[FFT_N];
  double window[FFT_N];
  signed char notes[FFT_N / 2 + 1];
  double notes_frac[FFT_N / 2 + 1];
  int min_index, max_index;
} tables_t;

static tables_t *g_tab;

/* ---- stage functions: each is what the pipeline below runs AND what tests/test_oracle.py checks against
 * libchromaprint's own unit-test vectors (tests/golden/chromaprint_unit_vectors.json) ------------------ */

/* PrepareHammingWindow(first, last, scale) of utils.h: the denominator is size - 1. */
void ora_prepare_hamming_window(double *w, int size, double scale) {
  for (int i = 0; i < size; i++) w[i] = scale * (0.54 - 0.46 * cos(i * 2.0 * M_PI / (size - 1)));
}

/* Chroma::PrepareNotes (chroma.cpp): bins [min_index, max_index) -> pitch class 0..11 counted from A
 * (27.5 Hz * 2^k), and the position inside the class (used only by the interpolating variant). */
void ora_chroma_prepare_notes(int min_freq, int max_freq, int frame_size, int sample_rate, signed char *notes,
                              double *notes_frac, int *min_index, int *max_index) {
  int lo = (int)round((double)frame_size * min_freq / sample_rate); /* FreqToIndex */
  int hi = (int)round((double)frame_size * max_freq / sample_rate);
  *min_index = lo > 1 ? lo : 1;
  *max_index = hi < frame_size / 2 ? hi : frame_size / 2;
  for (int i = *min_index; i < *max_index; i++) {
    double freq = (double)i * sample_rate / frame_size; /* IndexToFreq */
    double octave = log(freq / (440.0 / 16.0)) / log(2.0); /* FreqToOctave, base = A0 */
    double note = ORA_NUM_BANDS * (octave - floor(octave));
    notes[i] = (signed char)note;
    if (notes_frac) notes_frac[i] = note - notes[i];
  }
}

/* Chroma::Consume: energy of every bin added to its class; with `interpolate`, split between the class and
 * its nearer neighbour (the default fingerprinter uses interpolate = false). */
void ora_chroma_consume(const signed char *notes, const double *notes_frac, int min_index, int max_index,
                        int interpolate, const double *frame, double *features) {
  for (int c = 0; c < ORA_NUM_BANDS; c++) features[c] = 0.0;
  for (int i = min_index; i < max_index; i++) {
    int note = notes[i];
    double energy = frame[i];
    if (interpolate) {
      int note2 = note;
      double a = 1.0;
      if (notes_frac[i] < 0.5) {
        note2 = (note + ORA_NUM_BANDS - 1) % ORA_NUM_BANDS;
        a = 0.5 + notes_frac[i];
      }
      if (notes_frac[i] > 0.5) {
        note2 = (note + 1) % ORA_NUM_BANDS;
        a = 1.5 - notes_frac[i];
      }
      features[note] += energy * a;
      features[note2] += energy * (1.0 - a);
    } else {
      features[note] += energy;
    }
  }
}

/* ChromaFilter::Consume for one output row: rows[0] is the OLDEST of the `taps` buffered rows. */
void ora_chroma_filter_row(const double *const *rows, const double *coefficients, int taps, int bands, double *out) {
  for (int c = 0; c < bands; c++) {
    out[c] = 0.0;
    for (int j = 0; j < taps; j++) out[c] += rows[j][c] * coefficients[j];
  }
}

/* NormalizeVector(begin, end, EuclideanNorm, threshold) of utils.h as ChromaNormalizer calls it. */
void ora_normalize_vector(double *v, int n, double threshold) {
  double squares = 0.0;
  for (int c = 0; c < n; c++) squares += v[c] * v[c];
  double norm = squares > 0.0 ? sqrt(squares) : 0.0;
  if (norm < threshold) {
    for (int c = 0; c < n; c++) v[c] = 0.0;
  } else {
    for (int c = 0; c < n; c++) v[c] /= norm;
  }
}

/* Quantizer::Quantize */
int ora_quantize(double value, double t0, double t1, double t2) {
  if (value < t1) return value < t0 ? 0 : 1;
  return value < t2 ? 2 : 3;
}

static const tables_t *tables(void) {
  if (g_tab) return g_tab;
  tables_t *t = (tables_t *)calloc(1, sizeof(tables_t));
  for (int k = 0; k < FFT_N; k++) {
    long double a = -2.0L * 3.14159265358979323846264338327950288L * k / FFT_N;
    t->tw[k].re = (double)cosl(a);
    t->tw[k].im = (double)sinl(a);
  }
  ora_prepare_hamming_window(t->window, FFT_N, 1.0 / 32767.0); /* scale = 1 / INT16_MAX (fft.cpp) */
  ora_chroma_prepare_notes(MIN_FREQ, MAX_FREQ, FFT_N, ORA_SAMPLE_RATE, t->notes, t->notes_frac, &t->min_index,
                           &t->max_index);
  g_tab = t;
  return t;
}

/* Stockham autosort radix-4 DIF, forward transform (e^{-i..}); 4096 = 4^6 so six identical passes,
 * ping-ponging between a and a scratch buffer; result ends in `a` in natural order. */
static void fft4096(cpx *a, cpx *scratch, const tables_t *t) {
  cpx *x = a, *y = scratch;
  for (int n = FFT_N, s = 1; n > 1; n >>= 2, s <<= 2) {
    const int n1 = n / 4, n2 = n / 2, n3 = n1 + n2, tstep = FFT_N / n;
    for (int p = 0; p < n1; p++) {
      const cpx w1 = t->tw[p * tstep], w2 = t->tw[2 * p * tstep], w3 = t->tw[3 * p * tstep];
      for (int q = 0; q < s; q++) {
        const cpx A = x[q + s * p], B = x[q + s * (p + n1)], C = x[q + s * (p + n2)], D = x[q + s * (p + n3)];
        const double apc_r = A.re + C.re, apc_i = A.im + C.im;
        const double amc_r = A.re - C.re, amc_i = A.im - C.im;
        const double bpd_r = B.re + D.re, bpd_i = B.im + D.im;
        /* j*(b-d) */
        const double jbmd_r = -(B.im - D.im), jbmd_i = B.re - D.re;
        cpx *o = y + q + s * 4 * p;
        o[0].re = apc_r + bpd_r;
        o[0].im = apc_i + bpd_i;
        double r = amc_r - jbmd_r, i = amc_i - jbmd_i;
        o[s].re = r * w1.re - i * w1.im;
        o[s].im = r * w1.im + i * w1.re;
        r = apc_r - bpd_r, i = apc_i - bpd_i;
        o[2 * s].re = r * w2.re - i * w2.im;
        o[2 * s].im = r * w2.im + i * w2.re;
        r = amc_r + jbmd_r, i = amc_i + jbmd_i;
        o[3 * s].re = r * w3.re - i * w3.im;
        o[3 * s].im = r * w3.im + i * w3.re;
      }
    }
    cpx *tmp = x;
    x = y;
    y = tmp;
  }
  /* six passes: data is back in `a` */
}

/* ---- rolling integral image (chromaprint RollingIntegralImage): running 2-D prefix sums in double,
 * never re-based; Area() differences rows r1-1 / r2-1 of the running table. ------------------------ */
#define RING 64 /* >= max_filter_width + 1 rows are ever addressed */

typedef struct {
  double row[RING][ORA_NUM_BANDS];
  size_t num_rows;
} integral_t;

static void image_add_row(integral_t *im, const double *f) {
  double *cur = im->row[im->num_rows % RING];
  double acc = 0.0;
  for (int c = 0; c < ORA_NUM_BANDS; c++) { /* std::partial_sum */
    acc = (c == 0) ? f[0] : acc + f[c];
    cur[c] = acc;
  }
  if (im->num_rows > 0) {
    const double *last = im->row[(im->num_rows - 1) % RING];
    for (int c = 0; c < ORA_NUM_BANDS; c++) cur[c] += last[c];
  }
  im->num_rows++;
}

static double area(const integral_t *im, size_t r1, size_t c1, size_t r2, size_t c2) {
  if (r1 == r2 || c1 == c2) return 0.0;
  if (r1 == 0) {
    const double *row = im->row[(r2 - 1) % RING];
    if (c1 == 0) return row[c2 - 1];
    return row[c2 - 1] - row[c1 - 1];
  } else {
    const double *row1 = im->row[(r1 - 1) % RING];
    const double *row2 = im->row[(r2 - 1) % RING];
    if (c1 == 0) return row2[c2 - 1] - row1[c2 - 1];
    return row2[c2 - 1] - row1[c2 - 1] - row2[c1 - 1] + row1[c1 - 1];
  }
}

static double subtract_log(double a, double b) { return log((1.0 + a) / (1.0 + b)); }

/* chromaprint filter_utils.h Filter0..Filter5: x = first row (time), y = first column (band),
 * w = rows, h = columns. */
static double filter_apply(const classifier_t *c, const integral_t *im, size_t x) {
  size_t y = (size_t)c->y, w = (size_t)c->width, h = (size_t)c->height;
  double a, b;
  switch (c->type) {
    case 0:
      a = area(im, x, y, x + w, y + h);
      b = 0;
      break;
    case 1: {
      size_t h2 = h / 2;
      a = area(im, x, y + h2, x + w, y + h);
      b = area(im, x, y, x + w, y + h2);
      break;
    }
    case 2: {
      size_t w2 = w / 2;
      a = area(im, x + w2, y, x + w, y + h);
      b = area(im, x, y, x + w2, y + h);
      break;
    }
    case 3: {
      size_t w2 = w / 2, h2 = h / 2;
      a = area(im, x, y + h2, x + w2, y + h) + area(im, x + w2, y, x + w, y + h2);
      b = area(im, x, y, x + w2, y + h2) + area(im, x + w2, y + h2, x + w, y + h);
      break;
    }
    case 4: {
      size_t h3 = h / 3;
      a = area(im, x, y + h3, x + w, y + 2 * h3);
      b = area(im, x, y, x + w, y + h3) + area(im, x, y + 2 * h3, x + w, y + h);
      break;
    }
    default: {
      size_t w3 = w / 3;
      a = area(im, x + w3, y, x + 2 * w3, y + h);
      b = area(im, x, y, x + w3, y + h) + area(im, x + 2 * w3, y, x + w, y + h);
      break;
    }
  }
  return subtract_log(a, b);
}

static int quantize(const classifier_t *c, double v) { return ora_quantize(v, c->t0, c->t1, c->t2); }

size_t ora_chromaprint_fingerprint(const int16_t *pcm, size_t num_values, int channels,
                                   uint32_t *items, size_t cap, double *chroma_out,
                                   double *feature_out, double *min_margin) {
  const tables_t *t = tables();
  if (channels < 1) channels = 1;
  size_t samples = num_values / (size_t)channels;
  size_t frames = ora_chromaprint_num_frames(samples);
  size_t n_items = ora_chromaprint_num_items(samples);
  double margin = INFINITY;

  /* AudioProcessor::LoadMono / LoadStereo / LoadMultiChannel: integer mean, C truncation. */
  int16_t *mono = (int16_t *)malloc((samples ? samples : 1) * sizeof(int16_t));
  if (channels == 1) {
    memcpy(mono, pcm, samples * sizeof(int16_t));
  } else if (channels == 2) {
    for (size_t i = 0; i < samples; i++) mono[i] = (int16_t)(((int)pcm[2 * i] + (int)pcm[2 * i + 1]) / 2);
  } else {
    for (size_t i = 0; i < samples; i++) {
      long sum = 0;
      for (int c = 0; c < channels; c++) sum += pcm[i * (size_t)channels + c];
      mono[i] = (int16_t)(sum / channels);
    }
  }

  cpx *buf = (cpx *)malloc(2 * FFT_N * sizeof(cpx));
  double(*power)[FFT_N / 2 + 1] = malloc(2 * sizeof(*power));
  double ring[8][ORA_NUM_BANDS]; /* ChromaFilter: 8-slot ring, first output with the 5th row */
  int ring_off = 0, ring_size = 1;
  integral_t *im = (integral_t *)calloc(1, sizeof(integral_t));
  size_t produced = 0, fir_rows = 0;

  for (size_t f0 = 0; f0 < frames; f0 += 2) {
    int two = (f0 + 1 < frames);
    const int16_t *x1 = mono + f0 * ORA_FRAME_HOP;
    const int16_t *x2 = two ? mono + (f0 + 1) * ORA_FRAME_HOP : NULL;
    for (int i = 0; i < FFT_N; i++) {
      buf[i].re = (double)x1[i] * t->window[i];
      buf[i].im = two ? (double)x2[i] * t->window[i] : 0.0;
    }
    fft4096(buf, buf + FFT_N, t);
    /* split Z = X1 + i X2 into the two real-input spectra; energy = re^2 + im^2 (no sqrt, no 1/N) */
    for (int k = 0; k <= FFT_N / 2; k++) {
      cpx a = buf[k], b = buf[(FFT_N - k) & (FFT_N - 1)];
      double r1 = 0.5 * (a.re + b.re), i1 = 0.5 * (a.im - b.im);
      double r2 = 0.5 * (a.im + b.im), i2 = 0.5 * (b.re - a.re);
      power[0][k] = r1 * r1 + i1 * i1;
      power[1][k] = r2 * r2 + i2 * i2;
    }
    for (int which = 0; which < 1 + two; which++) {
      size_t f = f0 + (size_t)which;
      /* Chroma::Consume (interpolate = false) */
      double feat[ORA_NUM_BANDS];
      ora_chroma_consume(t->notes, t->notes_frac, t->min_index, t->max_index, 0, power[which], feat);
      if (chroma_out) memcpy(chroma_out + f * ORA_NUM_BANDS, feat, sizeof(feat));

      /* ChromaFilter::Consume */
      memcpy(ring[ring_off], feat, sizeof(feat));
      ring_off = (ring_off + 1) % 8;
      if (ring_size < ORA_FIR_TAPS) {
        ring_size++;
        continue;
      }
      int off = (ring_off + 8 - ORA_FIR_TAPS) % 8;
      double res[ORA_NUM_BANDS];
      const double *rows[ORA_FIR_TAPS];
      for (int j = 0; j < ORA_FIR_TAPS; j++) rows[j] = ring[(off + j) % 8];
      ora_chroma_filter_row(rows, kFir, ORA_FIR_TAPS, ORA_NUM_BANDS, res);
      /* ChromaNormalizer: Euclidean norm, threshold 0.01 */
      ora_normalize_vector(res, ORA_NUM_BANDS, 0.01);
      if (feature_out) memcpy(feature_out + fir_rows * ORA_NUM_BANDS, res, sizeof(res));
      fir_rows++;

      /* FingerprintCalculator::Consume */
      image_add_row(im, res);
      if (im->num_rows >= ORA_MAX_FILTER_WIDTH) {
        size_t x = im->num_rows - ORA_MAX_FILTER_WIDTH;
        uint32_t bits = 0;
        for (int c = 0; c < 16; c++) {
          double v = filter_apply(&kClassifiers[c], im, x);
          double m0 = fabs(v - kClassifiers[c].t0), m1 = fabs(v - kClassifiers[c].t1),
                 m2 = fabs(v - kClassifiers[c].t2);
          if (m0 < margin) margin = m0;
          if (m1 < margin) margin = m1;
          if (m2 < margin) margin = m2;
          bits = (bits << 2) | kGray[quantize(&kClassifiers[c], v)];
        }
        if (items && produced < cap) items[produced] = bits;
        produced++;
      }
    }
  }
  free(im);
  free(power);
  free(buf);
  free(mono);
  if (min_margin) *min_margin = margin;
  (void)n_items;
  return produced;
}

uint32_t ora_simhash32(const uint32_t *data, size_t n) {
  int v[32];
  for (int b = 0; b < 32; b++) v[b] = 0;
  for (size_t i = 0; i < n; i++) {
    uint32_t h = data[i];
    for (int b = 0; b < 32; b++) {
      v[b] += (h & 1u) ? 1 : -1;
      h >>= 1;
    }
  }
  uint32_t out = 0;
  for (int b = 0; b < 32; b++)
    if (v[b] > 0) out |= (1u << b);
  return out;
}
