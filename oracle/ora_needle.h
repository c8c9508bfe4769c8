/*
 * ORACLE — test infrastructure only. Never linked into, imported by, or called from the product.
 *
 * CPU restatement (plain C) of the parts of the needle hot path whose arithmetic IS in the reference
 * tree: the Analyzer's step/timestamp rule, the FrameHashes container and its bincode file, the
 * Comparator (LCS-Hamming DP, BinaryHeap order, best-match clustering) and the header-MD5 /
 * skip-file helpers.  Every function cites the reference lines it follows.
 *
 * Pinning status: the reference holds NO test for comparator.rs / data.rs, so the
 * search stage is pinned by a line-by-line reading of the Rust only ("parity unpinned" in the strict
 * sense); the one golden value the reference's tests do hold on this path — the header MD5
 * 759c6a520c5ce70359fdff38c4be6b98 of needle/resources/sample-5s.mp4
 * (needle/src/audio/snapshots/needle__audio__analyzer__test__analyzer.snap:43) — is checked in
 * tests/test_oracle.py.
 */
#ifndef ORA_NEEDLE_H
#define ORA_NEEDLE_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

typedef uint64_t ora_ns; /* std::time::Duration as total nanoseconds */

typedef struct {
  uint32_t hash;
  ora_ns ts;
} ora_hash_ts; /* (u32, Duration), data.rs:22-23 */

typedef struct {
  ora_hash_ts *opening;
  size_t n_opening;
  ora_hash_ts *ending;
  size_t n_ending;
  ora_ns hash_duration;
  char md5[33];
} ora_frame_hashes; /* FrameHashesV1, data.rs:20-26 */

/* ---- Duration arithmetic exactly as std does it ------------------------------------------------ */
ora_ns ora_duration_from_secs_f32(float s);       /* Duration::from_secs_f32: exact value, round-to-nearest-even ns */
ora_ns ora_duration_from_secs_f64(double s);
float ora_duration_as_secs_f32(ora_ns d);         /* secs as f32 + nanos as f32 / 1e9f32 */
ora_ns ora_duration_mul_f32(ora_ns d, float rhs); /* from_secs_f32(rhs * as_secs_f32()) */

/* ---- analyzer.rs:288-323: decimate raw items, attach timestamps -------------------------------- */
/* Returns number of kept hashes written (<= cap); 0 and *err=1 if step_by would be 0 (Rust panics). */
size_t ora_step_and_timestamp(const uint32_t *raw, size_t n_raw, ora_ns hash_duration,
                              int delay_ms, int item_ms, ora_ns seek_to, int has_seek,
                              ora_hash_ts *out, size_t cap, int *err);

/* ---- util.rs:99-105: lowercase hex MD5 of the first 8192 bytes (read_exact: shorter file = error) */
void ora_md5_hex(const uint8_t *data, size_t n, char out[33]);
int ora_header_md5(const char *path, char out[33]);

/* ---- data.rs + bincode 1.3 default config -------------------------------------------------------- */
/* return 0 ok, 1 io error, 2 malformed (bincode error), 3 version mismatch (data.rs:96-101) */
int ora_frame_hashes_write(const char *path, const ora_frame_hashes *fh);
int ora_frame_hashes_read(const char *path, ora_frame_hashes *fh);
void ora_frame_hashes_free(ora_frame_hashes *fh);

/* ---- comparator.rs ----------------------------------------------------------------------------- */
typedef struct {
  size_t score;
  ora_ns src_start, src_end; /* src_longest_run */
  ora_ns dst_start, dst_end; /* dst_longest_run */
  uint32_t src_match_hash, dst_match_hash;
  bool is_src_opening, is_src_ending, is_dst_opening, is_dst_ending;
  ora_ns src_hash_duration, dst_hash_duration;
  /* not part of the Rust struct (and not part of Ord): table coordinates, for kernel-level checks */
  uint32_t src_end_idx, dst_end_idx;
} ora_entry; /* ComparatorHeapEntry, comparator.rs:22-35 */

typedef struct {
  bool include_endings;
  uint32_t hash_match_threshold;
  ora_ns min_opening_duration, min_ending_duration, time_padding;
} ora_comparator; /* Comparator fields, comparator.rs:74-81 */

void ora_comparator_default(ora_comparator *c); /* comparator.rs:83-94 + audio/mod.rs:14,29,34,45 */

/* comparator.rs:157-250, literal: full (n+1)x(m+1) table of usize, forward fill, reverse walk,
 * BinaryHeap pushes, heap.into().  Returns malloc'd entries in BinaryHeap backing-array order. */
ora_entry *ora_longest_common_hash_match(const ora_comparator *c, const ora_hash_ts *src, size_t n,
                                         const ora_hash_ts *dst, size_t m, ora_ns src_hash_duration,
                                         ora_ns dst_hash_duration, bool is_opening, size_t *n_out);

/* The same entries without the table (every diagonal walked once, validity tested at each run's end, pushes in the
 * reverse walk's order): for checks at library scale.  Compared entry by entry with the literal form in tests/. */
ora_entry *ora_longest_common_hash_match_tablefree(const ora_comparator *c, const ora_hash_ts *src, size_t n,
                                                   const ora_hash_ts *dst, size_t m, ora_ns src_hash_duration,
                                                   ora_ns dst_hash_duration, bool is_opening, size_t *n_out);

typedef struct {
  bool has_result;  /* false <=> find_best_match returned None: video is skipped, comparator.rs:608-617 */
  bool has_opening, has_ending;
  ora_ns opening_start, opening_end, ending_start, ending_end;
} ora_search_result; /* Option<SearchResult>, comparator.rs:65-69 */

/* comparator.rs:524-629 without display / skip files.  out has one slot per video.
 * Returns 0 ok; 1 = FrameHashDataNoEnding (comparator.rs:271-273); 2 = Duration underflow (Rust panic). */
int ora_run_with_frame_hashes(const ora_comparator *c, const ora_frame_hashes *fh, size_t n_videos,
                              ora_search_result *out);

/* comparator.rs:524-629 for the videos[0..n_sel) of a library of n_videos, through the table-free pair function:
 * what ora_run_with_frame_hashes returns for those videos, at a cost of n_sel x n_videos pair scans instead of
 * n_videos^2 / 2 tables.  out[k] is the result of videos[k]. */
int ora_run_selected_videos(const ora_comparator *c, const ora_frame_hashes *fh, size_t n_videos, const size_t *videos,
                            size_t n_sel, ora_search_result *out);

/* thread count used by ora_run_with_frame_hashes / ora_analyze_batch (the rayon pool stand-in) */
/* An OPTIMISED CPU variant of the pair scan, for an honest second CPU baseline: no table, every
 * diagonal walked once with a running match length, runs shorter than min_len dropped on the spot.  Returns the
 * number of runs with L >= min_len over all pairs of `n_videos` hash sequences (parallel over pairs with the
 * threads of ora_set_threads); if `runs` is not NULL the first `cap` runs are stored as (pair, src_end, dst_end,
 * len) quadruples, in no particular order.  The same runs as the table walk of comparator.rs:175-247 reports. */
size_t ora_diagonal_runs_all_pairs(const uint32_t *const *hashes, const size_t *lens, size_t n_videos,
                                   uint32_t threshold, uint32_t min_len, uint32_t *runs, size_t cap);

void ora_set_threads(int n);
int ora_get_threads(void);

/* analyzer.rs:425-455 at the PCM boundary (opening window only = default Analyzer, :101) */
int ora_analyze_batch(const int16_t *const *pcm, const size_t *num_values, int channels, size_t n_eps,
                      ora_ns hash_duration, ora_frame_hashes *out);

/* data.rs:8-13 + comparator.rs:329-351: serde_json body of a skip file; returns length, 0 if the
 * reference would not write one (both None). */
size_t ora_skip_file_json(const ora_search_result *r, const char *md5, char *buf, size_t cap);

/* util.rs:8-12 */
void ora_format_time(ora_ns t, char out[32]);

#endif
