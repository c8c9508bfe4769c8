/*
 * ORACLE — test infrastructure only. Never linked into, imported by, or called from the product
 * (needle_amd/, libneedle_capi.so). Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may use it, and only as the checker.
 *
 * CPU restatement (plain C, double precision) of the fingerprinting arithmetic the reference reaches
 * through chromaprint::Context (needle/src/audio/analyzer.rs:176,179,218,275,286,288-289,300).
 *
 * That arithmetic lives in a THIRD-PARTY dependency that is NOT vendored under /root/reference:
 *   chromaprint-rust 0.1.3  -> chromaprint-sys-next 1.5.3 (needle/Cargo.lock:147-148,158-159),
 *   which builds libchromaprint 1.5.x (C++) with cmake.  What follows restates chromaprint's published
 *   default algorithm (CHROMAPRINT_ALGORITHM_TEST2): 11025 Hz mono, 4096-sample Hamming frames with
 *   hop 1365, power spectrum, 12-class chroma fold over 28..3520 Hz, 5-tap temporal FIR, L2 normalise,
 *   16 Haar-like classifiers on a rolling integral image, 2-bit Gray-coded quantisation -> u32/item.
 *
 * PINNING (analyze stage).  The reference holds no usable golden vector for this stage (its only analyzer
 * test is #[ignore]d with a stale snapshot of AAC media nothing here can decode, analyzer.rs:472-480) and neither
 * the Rust crate nor libchromaprint can be built in this image.  What pins this file instead are libchromaprint's
 * OWN known answers, written down from its published test-suite and replayed in tests/test_oracle.py:
 *   - tests/test_api.cpp Test2SilenceFp / Test2SilenceRawFp: the whole pipeline on silence, three items
 *     627964279 (tests/golden/chromaprint_silence.json): frame/latency arithmetic, every classifier's
 *     thresholds around 0, Gray code, bit packing;
 *   - tests/test_utils.cpp PrepareHammingWindow (the 10-point window: denominator size - 1),
 *     test_chroma.cpp (6 vectors incl. the six-digit interpolated ones), test_chroma_filter.cpp (3),
 *     test_chroma_normalizer.cpp (3), test_quantizer.cpp (8) against the stage functions below, which are the
 *     functions ora_chromaprint_fingerprint itself runs (tests/golden/chromaprint_unit_vectors.json).
 * PARITY UNPINNED for what those cannot reach: hashes of non-silent audio end to end against a real
 * libchromaprint build (window + FFT round-off feeding the quantisers) -- there this file DEFINES the expected
 * hashes.  FFT arithmetic is double (= chromaprint built against FFTW3, the backend the reference's README,
 * Dockerfile and release workflow install, README.md:168, Dockerfile:11, .github/workflows/release-needle.yml:58);
 * chromaprint's other FFT backends are single precision and differ among themselves in low-order hash bits.
 */
#ifndef ORA_CHROMAPRINT_H
#define ORA_CHROMAPRINT_H

#include <stddef.h>
#include <stdint.h>

#define ORA_SAMPLE_RATE 11025
#define ORA_FRAME_SIZE 4096
#define ORA_FRAME_HOP 1365 /* 4096 - (4096 - 4096/3) */
#define ORA_NUM_BANDS 12
#define ORA_FIR_TAPS 5
#define ORA_MAX_FILTER_WIDTH 16

/* chromaprint_get_delay_ms / chromaprint_get_item_duration_ms as the reference reads them
 * (analyzer.rs:288-289): integer milliseconds, truncated. */
int ora_chromaprint_delay_ms(void);         /* 2600 */
int ora_chromaprint_item_duration_ms(void); /* 123 */

/* Number of FFT frames / raw fingerprint items produced for `mono_samples` samples. */
size_t ora_chromaprint_num_frames(size_t mono_samples);
size_t ora_chromaprint_num_items(size_t mono_samples);

/*
 * Fingerprint one stream.  `pcm` holds `num_values` interleaved s16 values (`channels` = 1 or 2;
 * the reference always feeds 2, analyzer.rs:218).  Writes up to `cap` raw items to `items`.
 * Optional debug outputs (may be NULL):
 *   chroma_out   [frames][12]   un-normalised chroma energy per FFT frame
 *   feature_out  [frames-4][12] FIR-filtered, L2-normalised rows fed to the classifiers
 *   min_margin   smallest |filter value - threshold| over every quantiser decision taken
 * Returns the number of items the stream produces (independent of cap).
 */
size_t ora_chromaprint_fingerprint(const int16_t *pcm, size_t num_values, int channels,
                                   uint32_t *items, size_t cap, double *chroma_out,
                                   double *feature_out, double *min_margin);

/* chromaprint simhash (chromaprint-rust simhash::simhash32, called from comparator.rs:152). */
uint32_t ora_simhash32(const uint32_t *data, size_t n);

/* The stages of the pipeline one by one, with libchromaprint's constructor parameters, so that its own unit-test
 * vectors (tests/test_utils.cpp, test_chroma.cpp, test_chroma_filter.cpp, test_chroma_normalizer.cpp, test_quantizer.cpp) can be
 * replayed against exactly the code ora_chromaprint_fingerprint runs. */
void ora_prepare_hamming_window(double *w, int size, double scale);
void ora_chroma_prepare_notes(int min_freq, int max_freq, int frame_size, int sample_rate, signed char *notes,
                              double *notes_frac, int *min_index, int *max_index);
void ora_chroma_consume(const signed char *notes, const double *notes_frac, int min_index, int max_index,
                        int interpolate, const double *frame, double *features /* 12 */);
void ora_chroma_filter_row(const double *const *rows /* oldest first */, const double *coefficients, int taps,
                           int bands, double *out);
void ora_normalize_vector(double *v, int n, double threshold);
int ora_quantize(double value, double t0, double t1, double t2);

#endif
