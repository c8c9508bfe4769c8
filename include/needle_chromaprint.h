/*
 * needle_chromaprint.h — the subset of libchromaprint's C API that needle reaches through
 * chromaprint-rust 0.1.3 / chromaprint-sys-next 1.5.3 (needle/Cargo.lock:147-161), served by the MI355X
 * fingerprinter.  Call sites in the reference: needle/src/audio/analyzer.rs:176 (chromaprint_new), :179
 * (chromaprint_get_sample_rate), :218 (chromaprint_start), :275 (chromaprint_feed), :286 (chromaprint_finish),
 * :288 (chromaprint_get_delay_ms), :289 (chromaprint_get_item_duration_ms), :300
 * (chromaprint_get_raw_fingerprint + chromaprint_dealloc).
 *
 * Exported by libneedle_chromaprint.so with libchromaprint's own symbol names, so a needle built with
 * CHROMAPRINT_SYS_DYNAMIC (README.md:209) can be pointed at it instead of libchromaprint and runs its
 * analyze step on the GPU without source changes.  feed() only buffers PCM; finish() runs the whole stream
 * through needle_hip_fingerprint_host.  Conventions are libchromaprint's: every int function returns 1 on
 * success and 0 on error.  Only CHROMAPRINT_ALGORITHM_TEST2 (the default), 11025 Hz and 1-2 channels are
 * supported — needle never asks for anything else (analyzer.rs:179-187,218).
 */
#ifndef NEEDLE_CHROMAPRINT_H
#define NEEDLE_CHROMAPRINT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ChromaprintContextPrivate ChromaprintContext;

enum ChromaprintAlgorithm {
  CHROMAPRINT_ALGORITHM_TEST1 = 0,
  CHROMAPRINT_ALGORITHM_TEST2,
  CHROMAPRINT_ALGORITHM_TEST3,
  CHROMAPRINT_ALGORITHM_TEST4,
  CHROMAPRINT_ALGORITHM_TEST5,
  CHROMAPRINT_ALGORITHM_DEFAULT = CHROMAPRINT_ALGORITHM_TEST2
};

const char *chromaprint_get_version(void);
ChromaprintContext *chromaprint_new(int algorithm);
void chromaprint_free(ChromaprintContext *ctx);
int chromaprint_get_algorithm(ChromaprintContext *ctx);
int chromaprint_get_num_channels(ChromaprintContext *ctx);
int chromaprint_get_sample_rate(ChromaprintContext *ctx);
int chromaprint_get_item_duration(ChromaprintContext *ctx);    /* samples: 1365 */
int chromaprint_get_item_duration_ms(ChromaprintContext *ctx); /* 123 */
int chromaprint_get_delay(ChromaprintContext *ctx);            /* samples: 28666 */
int chromaprint_get_delay_ms(ChromaprintContext *ctx);         /* 2600 */
int chromaprint_start(ChromaprintContext *ctx, int sample_rate, int num_channels);
int chromaprint_feed(ChromaprintContext *ctx, const int16_t *data, int size);
int chromaprint_finish(ChromaprintContext *ctx);
int chromaprint_get_raw_fingerprint(ChromaprintContext *ctx, uint32_t **fingerprint, int *size);
int chromaprint_get_raw_fingerprint_size(ChromaprintContext *ctx, int *size);
int chromaprint_clear_fingerprint(ChromaprintContext *ctx);
void chromaprint_dealloc(void *ptr);

#ifdef __cplusplus
}
#endif

#endif /* NEEDLE_CHROMAPRINT_H */
