/*
 * needle.h — the needle-capi C ABI, as exported by the MI355X-native libneedle_capi.so.
 *
 * This is the drop-in boundary: every symbol, argument order, type and enum value below is what
 * the reference's `needle_capi` cdylib exports (Rust definitions in needle-capi/src/lib.rs, generated
 * header needle-capi/needle.h:146-248), so a C program written against upstream's header links
 * against this library unchanged (needle-capi/examples/{analyzer,comparator,full}.c do).
 *
 * The implementation behind it is C++ host code driving hand-written HIP kernels for gfx950; the
 * additive, GPU-facing entry points live in needle_hip.h and do not alter anything declared here.
 *
 * Differences a caller can observe are limited to what the reference delegates to FFmpeg: media
 * decode is out of scope, so "video" paths handed to the analyzer must be RIFF/WAVE PCM s16 files
 * at chromaprint's 11025 Hz (mono or stereo).  See INTEGRATION.md.
 */
#ifndef NEEDLE_H
#define NEEDLE_H

#include <stdarg.h>
#include <stdbool.h>
#include <stdint.h>
#include <stdlib.h>

#ifdef __cplusplus
extern "C" {
#endif

/* repr(C) enum, values 0..11 in this order — needle-capi/src/lib.rs:58-85 */
typedef enum NeedleError {
  NeedleError_Ok = 0,
  NeedleError_InvalidUtf8String,           /* a path is not valid UTF-8            lib.rs:297-300 */
  NeedleError_NullArgument,                /* a pointer argument was NULL          lib.rs:216,292,383 */
  NeedleError_InvalidArgument,             /* zero count / index out of range      lib.rs:219,426 */
  NeedleError_FrameHashDataNotFound,       /* <video>.needle.dat missing           lib.rs:126 */
  NeedleError_FrameHashDataInvalidVersion, /* version / payload tag disagree       lib.rs:127 */
  NeedleError_InvalidFrameHashData,        /* bincode decode failure               lib.rs:128 */
  NeedleError_ComparatorMinimumPaths,      /* comparator needs >= 2 paths          lib.rs:569-571 */
  NeedleError_AnalyzerInvalidHashPeriod,   /* unused upstream, kept for numbering */
  NeedleError_AnalyzerInvalidHashDuration, /* hash_duration <= 0                   lib.rs:474-476 */
  NeedleError_IOError,                     /* std::io::Error                       lib.rs:130 */
  NeedleError_Unknown,                     /* everything else                      lib.rs:129,131 */
} NeedleError;

/* Opaque handles (lib.rs:308, 347-350, 531). */
typedef struct FrameHashes FrameHashes;
typedef struct NeedleAudioAnalyzer NeedleAudioAnalyzer;
typedef struct NeedleAudioComparator NeedleAudioComparator;

/* lib.rs:138 — static NUL-terminated message, never freed by the caller. */
const char *needle_error_to_str(enum NeedleError error);

/* lib.rs:208 — NULL -> NullArgument, num_paths == 0 -> InvalidArgument; result freed with the next fn. */
enum NeedleError needle_util_find_video_files(const char *const *paths, size_t num_paths, bool full, bool audio,
                                              const char *const **videos, size_t *num_videos);

/* lib.rs:259 — NULL / 0 is a no-op. */
void needle_util_video_files_free(const char *const *videos, size_t num_videos);

/* lib.rs:354 — = _new(paths, n, 0.50, 0.25, false, false, false, output). */
enum NeedleError needle_audio_analyzer_new_default(const char *const *paths, size_t num_paths,
                                                   struct NeedleAudioAnalyzer **output);

/* lib.rs:373 — paths are copied, not checked for existence. */
enum NeedleError needle_audio_analyzer_new(const char *const *paths, size_t num_paths,
                                           float opening_search_percentage, float ending_search_percentage,
                                           bool include_endings, bool threaded_decoding, bool force,
                                           struct NeedleAudioAnalyzer **output);

/* lib.rs:415 — borrowed pointer into the analyzer, valid until it is re-run or freed. */
enum NeedleError needle_audio_analyzer_get_frame_hashes(const struct NeedleAudioAnalyzer *analyzer, size_t index,
                                                        const struct FrameHashes **output);

/* lib.rs:439 */
void needle_audio_analyzer_free(const struct NeedleAudioAnalyzer *analyzer);

/* lib.rs:450 — one path per line on stdout. */
void needle_audio_analyzer_print_paths(const struct NeedleAudioAnalyzer *analyzer);

/* lib.rs:465 — hash_duration in seconds; results are kept inside the handle. */
enum NeedleError needle_audio_analyzer_run(struct NeedleAudioAnalyzer *analyzer, float hash_duration, bool persist,
                                           bool threading);

/* lib.rs:537 — = _new(paths, n, false, 10, 20, 20, 0.0, output). */
enum NeedleError needle_audio_comparator_new_default(const char *const *paths, size_t num_paths,
                                                     const struct NeedleAudioComparator **output);

/* lib.rs:556 — durations in whole seconds, time_padding in seconds. */
enum NeedleError needle_audio_comparator_new(const char *const *paths, size_t num_paths, bool include_endings,
                                             uint16_t hash_match_threshold, uint16_t min_opening_duration,
                                             uint16_t min_ending_duration, float time_padding,
                                             const struct NeedleAudioComparator **output);

/* lib.rs:601 */
void needle_audio_comparator_free(const struct NeedleAudioComparator *comparator);

/* lib.rs:612 — results are observable through stdout (display) and skip files, as upstream. */
enum NeedleError needle_audio_comparator_run(const struct NeedleAudioComparator *comparator, bool analyze,
                                             bool display, bool use_skip_files, bool write_skip_files,
                                             bool threading);

#ifdef __cplusplus
}
#endif

#endif /* NEEDLE_H */
