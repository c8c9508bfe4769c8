/* ORACLE — test infrastructure only (see ora_resample.h). */
#include "ora_resample.h"

#include <math.h>
#include <stdlib.h>

#define TARGET 11025
#define ZERO_CROSSINGS 16
#define ROLLOFF 0.94
#define KAISER_BETA 9.0

static int gcd_i(int a, int b) {
  while (b) {
    int t = a % b;
    a = b;
    b = t;
  }
  return a;
}

static double bessel_i0(double x) {
  double sum = 1.0, term = 1.0;
  for (int k = 1; k <= 40; k++) {
    term *= (x / (2.0 * k)) * (x / (2.0 * k));
    sum += term;
  }
  return sum;
}

void ora_resample_design(int rate, int *phases, int *taps, int *m_step, float *coef) {
  const int g = gcd_i(TARGET, rate);
  const int L = TARGET / g, M = rate / g;
  const double ratio = (double)L / (double)M;
  const double scale = ROLLOFF * (ratio < 1.0 ? ratio : 1.0); /* cutoff relative to the input Nyquist */
  const int half = (int)ceil(ZERO_CROSSINGS / (ratio < 1.0 ? ratio : 1.0));
  const int T = 2 * half;
  if (rate == TARGET) { /* nothing to resample: identity (a single unit tap), only the down-mix applies */
    *phases = 1;
    *taps = 2;
    *m_step = 1;
    if (coef) {
      coef[0] = 1.0f;
      coef[1] = 0.0f;
    }
    return;
  }
  *phases = L;
  *taps = T;
  *m_step = M;
  if (!coef) return;
  const double pi = 3.14159265358979323846;
  const double i0b = bessel_i0(KAISER_BETA);
  for (int p = 0; p < L; p++) {
    double tmp[4096];
    double sum = 0.0;
    for (int k = 0; k < T; k++) {
      const double tau = (double)(k - half + 1) - (double)p / (double)L; /* tap position minus output position */
      const double x = tau * scale;
      const double sinc = x == 0.0 ? 1.0 : sin(pi * x) / (pi * x);
      const double u = tau / (double)half;
      const double win = fabs(u) >= 1.0 ? 0.0 : bessel_i0(KAISER_BETA * sqrt(1.0 - u * u)) / i0b;
      tmp[k] = scale * sinc * win;
      sum += tmp[k];
    }
    for (int k = 0; k < T; k++) coef[(size_t)p * T + k] = (float)(tmp[k] / sum);
  }
}

size_t ora_resample_out_len(size_t n, int rate) {
  const int g = gcd_i(TARGET, rate);
  const unsigned long long L = TARGET / g, M = rate / g;
  return (size_t)(((unsigned long long)n * L + M - 1) / M);
}

size_t ora_resample(const int16_t *pcm, size_t num_values, int channels, int rate, int16_t *out, size_t cap) {
  if (channels < 1) channels = 1;
  const size_t n = num_values / (size_t)channels;
  int L, T, M;
  ora_resample_design(rate, &L, &T, &M, NULL);
  float *coef = (float *)malloc((size_t)L * T * sizeof(float));
  ora_resample_design(rate, &L, &T, &M, coef);
  const int half = T / 2;
  const size_t n_out = ora_resample_out_len(n, rate);
  for (size_t m = 0; m < n_out && m < cap; m++) {
    const unsigned long long pos = (unsigned long long)m * (unsigned long long)M;
    const long long center = (long long)(pos / (unsigned long long)L);
    const int phase = (int)(pos % (unsigned long long)L);
    const long long first = center - half + 1;
    const float *c = coef + (size_t)phase * T;
    float acc = 0.0f;
    for (int k = 0; k < T; k++) {
      const long long idx = first + k;
      int s = 0;
      if (idx >= 0 && (size_t)idx < n) {
        if (channels == 1) {
          s = pcm[idx];
        } else {
          long sum = 0;
          for (int ch = 0; ch < channels; ch++) sum += pcm[(size_t)idx * channels + ch];
          s = (int)(sum / channels);
        }
      }
      acc = fmaf(c[k], (float)s, acc);
    }
    float r = rintf(acc);
    if (r > 32767.0f) r = 32767.0f;
    if (r < -32768.0f) r = -32768.0f;
    out[m] = (int16_t)r;
  }
  free(coef);
  return n_out;
}
