/*
 * ORACLE — test infrastructure only.
 *
 * CPU statement of the needle-hip resampler (the step in front of the fingerprinter: the reference resamples
 * with FFmpeg's swresample, needle/src/audio/analyzer.rs:180-187,231-282, which is a third-party library absent
 * from this image).  swresample cannot be reproduced bit for bit here, so this front-end is OUR OWN
 * specification — "parity unpinned" against the reference by construction — pinned only by this oracle:
 *
 *   1. down-mix: mono[n] = (sum of channels) / channels, C integer division (as chromaprint's AudioProcessor)
 *   2. rational polyphase FIR to 11025 Hz: L/M = 11025/rate in lowest terms, Kaiser(beta 9)-windowed sinc,
 *      16 zero crossings each side at the lower of the two Nyquist rates, roll-off 0.94, every phase normalised
 *      to unit DC gain; coefficients rounded once to f32; samples outside the stream are zero
 *   3. out[m] = clamp(rint(sum_k coef[phase][k] * mono[first + k])), f32 fused multiply-adds in tap order
 *
 *   n_out = ceil(n_in * L / M).
 */
#ifndef ORA_RESAMPLE_H
#define ORA_RESAMPLE_H

#include <stddef.h>
#include <stdint.h>

size_t ora_resample_out_len(size_t in_samples, int rate);
/* taps per output sample and number of phases for `rate`; coef (may be NULL) receives phases*taps floats */
void ora_resample_design(int rate, int *phases, int *taps, int *m_step, float *coef);
/* interleaved s16 in (num_values values, `channels` channels at `rate` Hz) -> mono s16 at 11025 Hz */
size_t ora_resample(const int16_t *pcm, size_t num_values, int channels, int rate, int16_t *out, size_t cap);

#endif
