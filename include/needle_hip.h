/*
 * needle_hip.h — additive, GPU-facing entry points of libneedle_capi.so (MI355X / gfx950).
 *
 * needle.h is the reference's C ABI verbatim; it only moves file paths.  The reference's hot path,
 * however, runs *below* that surface, inside `Analyzer::process_frames` (chromaprint feed/finish,
 * needle/src/audio/analyzer.rs:176-300) and `Comparator::longest_common_hash_match`
 * (needle/src/audio/comparator.rs:157-250).  The functions here are the C-ABI form of exactly those
 * two inner interfaces plus the in-memory calls the Rust API has and the C API lacks
 * (`Comparator::run_with_frame_hashes`, comparator.rs:524; `FrameHashes` accessors, data.rs:143-168 —
 * needle-capi/src/lib.rs:306-314 wraps FrameHashes as an opaque "TODO").  Plain pointers and sizes,
 * no C++/torch types.  All functions return NeedleError; details of the last failure on the calling
 * thread are available from needle_hip_last_error_message().
 *
 * Nothing here has a CPU fallback: without a usable HIP device the compute entry points fail with
 * NeedleError_Unknown ("no HIP device").
 */
#ifndef NEEDLE_HIP_H
#define NEEDLE_HIP_H

#include "needle.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- device / diagnostics ------------------------------------------------------------------------ */
enum NeedleError needle_hip_device_count(int *count);
enum NeedleError needle_hip_set_device(int ordinal);
enum NeedleError needle_hip_synchronize(void);
/* PCI address of the current device, "0000:c1:00.0" form (NUL-terminated): lets a host program find the device's
 * sysfs node (clocks, temperature) without HIP headers.  Diagnostics only. */
enum NeedleError needle_hip_device_pci_bus_id(char out[32]);
/* The HIP stream (a hipStream_t) every kernel and copy of this library is enqueued on, for the current device:
 * lets a caller order its own device work (collectives, framework kernels) against the library's by stream
 * order or events instead of host synchronisation.  NULL without a device. */
void *needle_hip_stream(void);
const char *needle_hip_last_error_message(void);
const char *needle_hip_version(void);

/* Thin device-memory helpers so a host program (Rust/C/Python) needs no HIP headers. */
enum NeedleError needle_hip_malloc(void **device_ptr, size_t bytes);
enum NeedleError needle_hip_free(void *device_ptr);
enum NeedleError needle_hip_memcpy_h2d(void *device_dst, const void *host_src, size_t bytes);
enum NeedleError needle_hip_memcpy_d2h(void *host_dst, const void *device_src, size_t bytes);
void needle_hip_host_free(void *ptr); /* frees arrays this library malloc'd for the caller */
/* Pinned (page-locked) host memory for PCM the caller wants uploaded at the full PCIe rate without staging. */
enum NeedleError needle_hip_host_alloc(void **host_ptr, size_t bytes);
enum NeedleError needle_hip_host_alloc_free(void *host_ptr);

/* GPU timing of the most recent COMPLETED launch of a kernel (it never waits behind queued work unless no
 * launch has finished yet), measured with HIP events on the
 * library's own stream (rocprofv3 sees the same kernels).  Names: "stft_chroma", "fir_norm",
 * "classify", "hamming_runs", "simhash_runs", "resample".  Returns milliseconds, <0 if unknown or not timed. */
double needle_hip_last_kernel_ms(const char *kernel);
/* Selects the kernels that get those events: "all", a comma-separated list of names, or NULL / "" / "none".
 * Default: none (each event record is one more packet between dependent dispatches: timing all five kernels of
 * a 28 x 24 min job costs 3 % of its time), unless the environment variable NEEDLE_HIP_KERNEL_TIMING is set.
 * With the extra item "sum" (e.g. "all,sum") every launch since this call keeps its own events and
 * needle_hip_last_kernel_ms returns their SUM (it waits for them): a kernel that one job launches several times -- the
 * first pass of a library-scale job -- then reads as the job's total, not as its last launch's. */
void needle_hip_set_kernel_timing(const char *kernels);
/* Diagnostic for the search roofline: table cells per second this device sustains on the scan's
 * per-cell instruction sequence (xor, popcount, compare, select) with operands in registers -- the integer-VALU
 * ceiling a brute-force evaluation of every cell of comparator.rs:176-187 cannot exceed.  Takes ~10 ms. */
enum NeedleError needle_hip_int_valu_ceiling(double *cells_per_second);
/* The other half of that roofline.  The fast scan does not evaluate every cell of the table: a run of >= min_len must
 * cover an aligned 8-row window, so only those rows are looked at, and most windows are abandoned after three of them.
 * With NEEDLE_HIP_SCAN_COUNT=1 in the environment every scan launch runs a COUNTING instantiation of the same kernel
 * (slower; never time it) that adds up the cell evaluations it ISSUES -- xor / popcount / compare over a wave's 64
 * lanes, inside the table or not.  This returns the sum since the last reset (waits for the library stream):
 * issued evaluations per second of the UNCOUNTED kernel against needle_hip_int_valu_ceiling() is a fraction <= 1. */
enum NeedleError needle_hip_scan_issued_evaluations(uint64_t *lane_evaluations, bool reset);
/* The same counting launches, two figures: counts[0] = the issued lane evaluations above, counts[1] = diagonals that
 * passed the head rows of a window (summed over all windows) and had to be finished -- what a cheaper head costs.  The
 * window shape itself (rows per aligned window W, head rows H; default 8, 3) can be chosen among the instantiated ones
 * with NEEDLE_HIP_SCAN_SHAPE="W,H" (W in 4, 8, 16; H in 2..4): every shape emits the same runs (tools/scan_shape_sweep.py). */
enum NeedleError needle_hip_scan_counts(uint64_t counts[2], bool reset);
/* Which form of the scan the last launch of this process took: 0 none yet, 1 one lane per diagonal (short minimum runs),
 * 2 bands, 3 aligned windows on the vector ALU, 4 aligned windows with the head rows on the matrix pipe (launches of 2048
 * sequence pairs or more, thresholds <= 15, windows that fill their row tiles; NEEDLE_HIP_SCAN_MFMA=0 / 1 forces either)
 * -- and for form 4 the v_mfma_f32_32x32x64_f8f6f4 instructions (FP4 operands, two head rows each) that launch issued
 * (131 072 operations each): the numerator of ITS roofline.  Every form emits the same runs. */
enum NeedleError needle_hip_scan_last_launch(int32_t *form, uint64_t *matrix_products);
/* How many jobs of this process asked for the per-video epilogue (comparator.rs:405-515, 583-626) on the DEVICE and were
 * handed back to the host form because one pair's bucket of runs exceeded what the device orders (since round 6 a
 * bucket beyond a lane's 24 runs is a workgroup's, up to 8192 runs: two fully silent 24-minute windows are 5 800;
 * beyond that, or with a row of 65 536 hashes or more).  Correct either way; this makes the performance cliff
 * visible (also printed under NEEDLE_HIP_TRACE).  reset: start counting again. */
enum NeedleError needle_hip_epilogue_host_fallbacks(uint64_t *jobs, bool reset);

/* ---- fingerprint: the chromaprint Context replacement -------------------------------------------
 * Replaces chromaprint::Context::{start,feed,finish,get_fingerprint_raw,get_delay,get_item_duration,
 * sample_rate} as called from analyzer.rs:176,179,218,275,286,288-289,300.  Batched: many streams
 * per call.  PCM is interleaved s16 at needle_hip_fingerprint_sample_rate() Hz, `channels` = 1 or 2
 * (the reference always feeds 2, analyzer.rs:218; stereo is down-mixed (L+R)/2 with C truncation on
 * the device).  `step` keeps raw items 0, step, 2*step, ... (analyzer.rs:293-304; step = 1 returns
 * chromaprint's full raw fingerprint). */
int needle_hip_fingerprint_sample_rate(void);      /* 11025 */
int needle_hip_fingerprint_delay_ms(void);         /* 2600  (chromaprint_get_delay_ms) */
int needle_hip_fingerprint_item_duration_ms(void); /* 123   (chromaprint_get_item_duration_ms) */
size_t needle_hip_fingerprint_num_items(size_t samples_per_channel);
size_t needle_hip_fingerprint_num_kept(size_t samples_per_channel, uint32_t step);

/* Host buffers in, host buffers out (does H2D/D2H).  items[i] must hold num_kept(...) values. */
enum NeedleError needle_hip_fingerprint_host(const int16_t *const *pcm, const size_t *num_values,
                                             size_t num_streams, int channels, uint32_t step,
                                             uint32_t *const *items);

/* PCM already resident in HBM: stream i is d_pcm[pcm_offsets[i] .. +num_values[i]) (offsets in s16
 * values, host arrays); kept items of stream i are written to d_items[item_offsets[i] ..].  Runs on
 * the library stream; returns after the kernels are enqueued unless `sync` is true. */
enum NeedleError needle_hip_fingerprint_device(const int16_t *d_pcm, const uint64_t *pcm_offsets,
                                               const uint64_t *num_values, size_t num_streams,
                                               int channels, uint32_t step, uint32_t *d_items,
                                               const uint64_t *item_offsets, bool sync);

/* Arithmetic of the fingerprinter.  The u32 items are DEFINED on a double-precision pipeline (chromaprint on an f64
 * FFT; oracle/ora_chromaprint.c).  By default the STFT runs a single-precision first pass and every item is either
 * certified -- all 48 threshold comparisons clear a data-dependent error radius, so the f64 pipeline provably takes the
 * same decisions -- or recomputed: the frames it covers go through the f64 kernel again and the item is classified
 * from those.  The emitted items are therefore the f64 pipeline's, bit for bit.  Environment: NEEDLE_HIP_STFT=f64 runs
 * the f64 kernel over everything; NEEDLE_HIP_CERT_K scales the radius (default 64; 0 accepts every first-pass item,
 * for tests).  counts = {items fingerprinted, items recomputed in f64, chunks of 2 frame pairs, chunks recomputed}
 * on the current device since the last reset; waits for the library stream. */
enum NeedleError needle_hip_fingerprint_cert_stats(uint64_t counts[4], bool reset);

/* Audit of that first pass.  The acceptance radius is an EMPIRICAL guard, not a proven bound: K = 64
 * times the error scale S, where the worst |log v32 - log v64| / S ever observed is 1.8.  This makes the claim checkable on
 * any input: both transforms -- the f32 first pass and the f64 kernel -- run over the same resident PCM into buffers of
 * their own and every kept item is examined on the device with the certification kernel's own functions and K.
 *   items                : kept items examined
 *   accepted             : items the first pass accepted (emitted from f32 chroma)
 *   accepted_mismatches  : accepted items whose f32 bits differ from the f64 pipeline's item (each one a hole: must be 0)
 *   mismatches           : items of `d_items` (what the product emitted for these streams) that differ from the f64 item
 *   max_error_over_s     : max over accepted items and their 16 classifiers of |log v32 - log v64| / S (compare with K)
 *   max_s                : largest S among accepted items
 * Same stream description as needle_hip_fingerprint_device; synchronous; allocates 2 x 96 B per frame while it runs. */
typedef struct NeedleHipCertAudit {
  uint64_t items, accepted, accepted_mismatches, mismatches;
  double max_error_over_s, max_s;
} NeedleHipCertAudit;
enum NeedleError needle_hip_fingerprint_audit_device(const int16_t *d_pcm, const uint64_t *pcm_offsets,
                                                     const uint64_t *num_values, size_t num_streams, int channels,
                                                     uint32_t step, const uint32_t *d_items, const uint64_t *item_offsets,
                                                     NeedleHipCertAudit *audit);

/* Test hook: intermediate stages of one stream, copied to the host.  chroma [frames][12] (energy per
 * pitch class per FFT frame), features [frames-4][12] (FIR-filtered, L2-normalised). NULLs allowed. */
enum NeedleError needle_hip_fingerprint_debug(const int16_t *pcm, size_t num_values, int channels,
                                              double *chroma, double *features);

/* ---- resampler + down-mix: the step in FRONT of the path ---------------------------------------------
 * The reference converts decoded audio to s16 @ 11025 Hz with FFmpeg's swresample before feeding chromaprint
 * (analyzer.rs:180-187,231-282).  That library is out of scope and not bit-reproducible; this front-end is this
 * project's own specification (integer down-mix, rational polyphase Kaiser-sinc FIR, f32 fused multiply-adds
 * in tap order, round to nearest even; oracle/ora_resample.h) so that PCM at the usual decode rates can be
 * analysed on the device.  Output: mono s16 at 11025 Hz, ceil(n * 11025 / rate) samples per stream.
 * needle_audio_analyzer_run and needle_hip_analyzer_run_pcm apply it automatically when the rate differs. */
size_t needle_hip_resample_out_len(size_t samples_per_channel, int sample_rate);
enum NeedleError needle_hip_resample_host(const int16_t *const *pcm, const size_t *num_values, size_t num_streams,
                                          int channels, int sample_rate, int16_t *const *out);

/* ---- search: the LCS-Hamming DP replacement -------------------------------------------------------
 * Replaces Comparator::longest_common_hash_match's two table sweeps (comparator.rs:175-247).  For a
 * problem (src, dst, min_len) it reports every maximal diagonal run of cells (i >= 1, j >= 1) with
 * popcount(src[i] ^ dst[j]) <= threshold whose length L >= min_len, as (src_end = i, dst_end = j, L)
 * of its last cell — exactly the cells the reference's reverse walk stops at (:196-200) with
 * table[i][j] = L — together with the two chromaprint simhashes the reference computes for such a run
 * (:226-229).  Order of the emitted runs is unspecified; the caller sorts. */
typedef struct NeedleHipSeq {
  uint32_t offset; /* first hash of the sequence inside the hash arena */
  uint32_t len;
} NeedleHipSeq;

typedef struct NeedleHipProblem {
  uint32_t src_seq;
  uint32_t dst_seq;
  uint32_t min_len; /* >= 1 */
  uint32_t tag;     /* copied to NeedleHipRun.problem */
} NeedleHipProblem;

typedef struct NeedleHipRun {
  uint32_t problem;
  uint32_t src_end;
  uint32_t dst_end;
  uint32_t len;
  uint32_t src_match_hash; /* simhash32 of src[src_end-len ..= src_end], L+1 hashes (comparator.rs:149-153,226-229) */
  uint32_t dst_match_hash; /* simhash32 of dst[dst_end-len ..= dst_end] */
} NeedleHipRun;

/* Hash arena resident in HBM; descriptors are host arrays.  Writes at most `capacity` runs to d_runs
 * and the TOTAL number found to *d_count (device).  Synchronous when `sync`. */
enum NeedleError needle_hip_hamming_runs_device(const uint32_t *d_hashes, const NeedleHipSeq *seqs,
                                                size_t num_seqs, const NeedleHipProblem *problems,
                                                size_t num_problems, uint32_t threshold,
                                                NeedleHipRun *d_runs, uint32_t capacity,
                                                uint32_t *d_count, bool sync);

/* Host arrays in, malloc'd run list out (free with needle_hip_host_free). */
enum NeedleError needle_hip_hamming_runs_host(const uint32_t *hashes, size_t num_hashes,
                                              const NeedleHipSeq *seqs, size_t num_seqs,
                                              const NeedleHipProblem *problems, size_t num_problems,
                                              uint32_t threshold, NeedleHipRun **runs, size_t *num_runs);

/* ---- FrameHashes (data.rs) ------------------------------------------------------------------------ */
enum NeedleError needle_hip_frame_hashes_new(const uint32_t *opening_hashes, const uint64_t *opening_ts_ns,
                                             size_t num_opening, const uint32_t *ending_hashes,
                                             const uint64_t *ending_ts_ns, size_t num_ending,
                                             uint64_t hash_duration_ns, const char *md5, FrameHashes **output);
void needle_hip_frame_hashes_free(FrameHashes *frame_hashes);
size_t needle_hip_frame_hashes_len(const FrameHashes *frame_hashes, bool ending);          /* data.rs:143,150 */
enum NeedleError needle_hip_frame_hashes_copy(const FrameHashes *frame_hashes, bool ending, uint32_t *hashes,
                                              uint64_t *ts_ns, size_t capacity);
uint64_t needle_hip_frame_hashes_hash_duration_ns(const FrameHashes *frame_hashes);        /* data.rs:157 */
const char *needle_hip_frame_hashes_md5(const FrameHashes *frame_hashes);                  /* data.rs:164 */
enum NeedleError needle_hip_frame_hashes_read(const char *path, FrameHashes **output);     /* data.rs:104-115 */
enum NeedleError needle_hip_frame_hashes_write(const FrameHashes *frame_hashes, const char *path);
enum NeedleError needle_hip_header_md5(const char *path, char out[33]);                    /* util.rs:99-105 */

/* ---- Analyzer at the PCM boundary -----------------------------------------------------------------
 * Analyzer::run (analyzer.rs:425) with FFmpeg's half of process_frames already done by the caller:
 * pcm[i] is the whole decoded stream of video i, interleaved s16 at `sample_rate` (anything but 11025 Hz is
 * resampled on the device first, see above).  Applies the opening / ending search windows
 * (analyzer.rs:378,390), fingerprints on the GPU, attaches timestamps (:293-318), stores the
 * FrameHashes in the handle (retrieve with needle_audio_analyzer_get_frame_hashes) and persists
 * <video>.needle.dat when `persist`. */
enum NeedleError needle_hip_analyzer_run_pcm(struct NeedleAudioAnalyzer *analyzer, const int16_t *const *pcm,
                                             const size_t *num_values, int channels, int sample_rate,
                                             float hash_duration, bool persist);

/* ---- Comparator in memory ------------------------------------------------------------------------- */
typedef struct NeedleHipSearchResult {
  bool has_result; /* false: the reference pushes no SearchResult for this video (comparator.rs:608-617) */
  bool has_opening;
  bool has_ending;
  uint64_t opening_start_ns, opening_end_ns;
  uint64_t ending_start_ns, ending_end_ns;
} NeedleHipSearchResult;

/* Comparator::run_with_frame_hashes (comparator.rs:524-629).  `results` has one slot per video. */
enum NeedleError needle_hip_comparator_run_with_frame_hashes(const struct NeedleAudioComparator *comparator,
                                                             const FrameHashes *const *frame_hashes,
                                                             size_t num_videos, bool display,
                                                             bool use_skip_files, bool write_skip_files,
                                                             NeedleHipSearchResult *results);

/* The host half of that call on its own: the order-sensitive epilogue (reverse-walk order, duration validity,
 * BinaryHeap array order, find_best_match; comparator.rs:191-249,405-515,583-626) from a COMPLETE run list of all
 * pairs (NeedleHipRun.problem = pair index * regions + region, regions = 2 with endings) to the results of videos
 * [first_video, first_video + video_count); the other slots are left empty.  No device work: this is what every rank
 * of a multi-GPU job runs for its own block of videos after the run lists have been all-gathered. */
enum NeedleError needle_hip_comparator_results_from_runs(const struct NeedleAudioComparator *comparator,
                                                         const FrameHashes *const *frame_hashes, size_t num_videos,
                                                         const NeedleHipRun *runs, size_t num_runs, size_t first_video,
                                                         size_t video_count, NeedleHipSearchResult *results);

/* ---- Library: an HBM-resident analyze+search job, shardable across GPUs ----------------------------
 * One object per process/GPU describing ALL videos of a job.  PCM of the videos this rank owns is
 * uploaded once and stays in HBM; hashes live in a padded device arena [video * rows_per_video][stride] so a
 * plain all-gather over contiguous video blocks fills the rows other ranks computed; pairs are
 * addressed by their index in the reference's lexicographic pair list (comparator.rs:534-545). */
typedef struct NeedleHipLibrary NeedleHipLibrary;

enum NeedleError needle_hip_library_new(size_t num_videos, float opening_search_percentage, float hash_duration,
                                        NeedleHipLibrary **output);
void needle_hip_library_free(NeedleHipLibrary *library);
/* Analyzer::with_include_endings + with_ending_search_percentage (analyzer.rs:130-139): also fingerprint the
 * last `ending_search_percentage` of every video.  Call before set_pcm; the arena then has two rows per video. */
enum NeedleError needle_hip_library_include_endings(NeedleHipLibrary *library, float ending_search_percentage);
size_t needle_hip_library_rows_per_video(const NeedleHipLibrary *library);
/* Lengths (values per stream, all videos) are metadata every rank holds; pcm[i] may be NULL for
 * videos this rank does not own.  Crops to the opening window and uploads. */
enum NeedleError needle_hip_library_set_pcm(NeedleHipLibrary *library, const int16_t *const *pcm,
                                            const size_t *num_values, int channels);
/* The same for PCM that is already in HBM (decoded or generated on the device): d_pcm[i] are DEVICE pointers, NULL for
 * videos this rank does not own.  The search windows are copied device to device into the library's arena; the caller's
 * buffers are free on return. */
enum NeedleError needle_hip_library_set_pcm_device(NeedleHipLibrary *library, const int16_t *const *d_pcm,
                                                   const size_t *num_values, int channels);
/* The streaming form ("analyze streamed from host-pinned PCM", BASELINE.json configs[4]): the search windows of the
 * videos with a non-NULL pointer are copied to the device in order on an upload stream and fingerprinted group by
 * group (~32 MiB of PCM each) on the library stream as they land, straight into their arena rows; nothing of the PCM
 * stays in HBM beyond a 2 GiB staging arena, and the kernels of all but the last group run underneath the copies.
 * Pinned host memory (needle_hip_host_alloc, or anything hipHostRegister'ed) is read in place by the copy engine;
 * pageable memory goes through a ring of pinned slabs filled by host threads.  On return the caller's buffers are
 * free; the last kernels may still be running (stream order: search / job_begin may follow at once).  Replaces
 * set_pcm + analyze; job_begin then skips its analyze step. */
enum NeedleError needle_hip_library_stream_pcm(NeedleHipLibrary *library, const int16_t *const *pcm,
                                               const size_t *num_values, int channels);
/* Fingerprint videos [first, first+count) into their arena rows (GPU only, no host copy). */
enum NeedleError needle_hip_library_analyze(NeedleHipLibrary *library, size_t first, size_t count, bool sync);
/* Arena geometry: device pointer to u32[num_videos * rows_per_video][stride]. */
enum NeedleError needle_hip_library_hash_arena(NeedleHipLibrary *library, uint32_t **d_arena, size_t *stride);
/* Adopt caller-owned device memory u32[rows][stride] (rows >= num_videos * rows_per_video, stride >= the library's) as the
 * arena, e.g. a buffer a collective library allocated so rows can be all-gathered in place.  Call after
 * set_pcm and before analyze; the caller keeps the memory alive and zero-initialised. */
enum NeedleError needle_hip_library_use_hash_arena(NeedleHipLibrary *library, uint32_t *d_arena, size_t rows,
                                                   size_t stride);
size_t needle_hip_library_num_pairs(const NeedleHipLibrary *library);
/* Runs of pairs [first_pair, first_pair+num_pairs) into caller-provided device buffers
 * (NeedleHipRun.problem = global pair index * comparator regions + region).  The fast scan kernels stage a pair's
 * destination window in LDS: up to ~39 000 hashes (2.7 h of audio at step 1, 5.3 h at the default step 2).  Pairs
 * with a longer window are scanned from HBM by a slower kernel in the same call; nothing fails (the reference has no
 * bound). */
enum NeedleError needle_hip_library_search(NeedleHipLibrary *library, const struct NeedleAudioComparator *comparator,
                                           size_t first_pair, size_t num_pairs, NeedleHipRun *d_runs,
                                           uint32_t capacity, uint32_t *d_count, bool sync);
/* Asynchronous download of a run list for pipelining jobs: _begin enqueues (on the library stream, behind the
 * search that fills them) the copy of *d_count and of the first max_runs runs into pinned host memory of
 * `slot` (0 or 1) and returns at once; _end waits for that copy only and hands back a pointer into the pinned
 * buffer (valid until the slot's next _begin).  *num_runs is the TOTAL found: if it exceeds max_runs the list
 * is truncated and the caller must fetch it synchronously.  Lets the host epilogue of job k overlap the
 * kernels of job k+1. */
enum NeedleError needle_hip_library_fetch_runs_begin(NeedleHipLibrary *library, int slot, const NeedleHipRun *d_runs,
                                                     const uint32_t *d_count, uint32_t max_runs);
enum NeedleError needle_hip_library_fetch_runs_end(NeedleHipLibrary *library, int slot, const NeedleHipRun **runs,
                                                   uint32_t *num_runs);
/* Host epilogue over the complete run list (all pairs): D2H of the arena, duration validity,
 * simhash32, BinaryHeap order, find_best_match.  `runs` is a host array. */
enum NeedleError needle_hip_library_finalize(NeedleHipLibrary *library, const struct NeedleAudioComparator *comparator,
                                             const NeedleHipRun *runs, size_t num_runs,
                                             NeedleHipSearchResult *results);
/* needle_hip_fingerprint_audit_device over the hashes this rank's share of the fingerprinting produced (all of them
 * without a communicator): the library's resident PCM through both transforms, compared with the arena's rows.  Call after
 * a job (or needle_hip_library_analyze) has filled the arena; needs set_pcm / set_pcm_device (resident PCM). */
enum NeedleError needle_hip_library_audit(NeedleHipLibrary *library, NeedleHipCertAudit *audit);
/* Copies one video's FrameHashes out of the library after analyze (+gather). */
enum NeedleError needle_hip_library_frame_hashes(NeedleHipLibrary *library, size_t index, FrameHashes **output);

/* ---- multi-GPU: one process per GPU, RCCL over xGMI --------------------------------------------------
 * The reference parallelises inside one process with rayon when `threading` is set: over videos in
 * Analyzer::run (analyzer.rs:437-445) and over pairs in Comparator::run_with_frame_hashes
 * (comparator.rs:549-564).  Across the GPUs of a node the same two fan-outs are: the hashes of all videos in equal
 * contiguous blocks per rank, pairs in contiguous ranges of the lexicographic pair list per rank, and two all-gathers between
 * them (hash rows after analyze, run lists after search) -- all inside this library, on its own streams, with
 * librccl loaded on demand (no link-time dependency; no torch).  A process is one rank and drives one device:
 *
 *   rank 0: needle_hip_comm_create_id(id); hand the 128 bytes to the other ranks by any means (file, socket, MPI)
 *   all   : needle_hip_set_device(local_rank); needle_hip_comm_init(id, rank, world_size);
 *           library_new / set_pcm (PCM pointers for the videos of needle_hip_library_rank_videos(...) only, NULL
 *           for the rest) / job_begin / job_end ...; needle_hip_comm_finalize()
 *
 * Results are identical for every world size: the run set is a union over disjoint pair ranges and the epilogue
 * orders it.  NEEDLE_HIP_COMM=host selects a host-staged transport over POSIX shared memory instead of RCCL (ranks of
 * one node, any device assignment, e.g. two ranks on one GPU) -- for exercising the N-rank path where RCCL cannot run
 * and as a fallback; it moves data only and is not a compute fallback. */
#define NEEDLE_HIP_COMM_ID_BYTES 128
enum NeedleError needle_hip_comm_create_id(uint8_t id[NEEDLE_HIP_COMM_ID_BYTES]);
enum NeedleError needle_hip_comm_init(const uint8_t id[NEEDLE_HIP_COMM_ID_BYTES], int rank, int world_size); /* collective */
void needle_hip_comm_finalize(void);
int needle_hip_comm_rank(void);             /* 0 without a communicator */
int needle_hip_comm_world_size(void);       /* 1 without a communicator */
const char *needle_hip_comm_backend(void);  /* "none", "rccl", "host" */
enum NeedleError needle_hip_comm_barrier(void);
/* Host buffers: rank r's `bytes_per_rank` bytes land at recv + r * bytes_per_rank on every rank. */
enum NeedleError needle_hip_comm_all_gather_host(const void *send, void *recv, size_t bytes_per_rank);
/* The sharding plan: `units` (videos, pairs) in world_size contiguous blocks of ceil(units / world_size). */
void needle_hip_comm_shard(size_t units, int world_size, int rank, size_t *first, size_t *count);

/* Which videos' PCM a rank has to hold.  The fingerprinting is sharded by HASHES, not by videos: the arena
 * u32[rows][stride] is cut, as one flat array, into world_size equal blocks and a rank computes the hashes of its block
 * from the stretch of PCM they depend on -- 28 episodes on 8 ranks are 3.5 episodes' worth of frames each, no rank
 * idles (whole videos would give 7 x 4 + 0).  For the stream lengths `num_values` (all videos; what set_pcm will be
 * given) this names the videos [first_video, first_video + video_count) whose rows rank `rank`'s block meets: pass
 * their PCM to set_pcm, NULL for the others.  A pure function of the library's parameters and the lengths. */
enum NeedleError needle_hip_library_rank_videos(const NeedleHipLibrary *library, const size_t *num_values, int channels,
                                                int world_size, int rank, size_t *first_video, size_t *video_count);

/* One analyze+search job of the library across the communicator (or on one GPU without one), in two halves so
 * that two jobs can be in flight (slot 0 / 1): _begin enqueues this rank's fingerprinting, the all-gather of hash
 * rows, the scan of this rank's pair range, the all-gather of run lists and their download, and returns at once;
 * _end waits for that download, runs the per-video epilogue (find_best_match, comparator.rs:583-626; sharded by video
 * across ranks with one more all-gather once it is large enough to pay for it) and fills results[num_videos] on
 * every rank.  *num_runs (optional): runs found over all pairs. */
enum NeedleError needle_hip_library_job_begin(NeedleHipLibrary *library, const struct NeedleAudioComparator *comparator,
                                              int slot);
enum NeedleError needle_hip_library_job_end(NeedleHipLibrary *library, const struct NeedleAudioComparator *comparator,
                                            int slot, NeedleHipSearchResult *results, size_t *num_runs);
/* The complete run list (all pairs, all ranks' shares in rank order) the epilogue of the job that finished last in
 * `slot` worked from: host memory of the library, valid until the slot's next job_begin.  For checkers and callers
 * that want the segments themselves ("the final cross-shard pair list"), not only the per-video results.
 * With more than one rank and the sharded device epilogue (library scale) the runs travel OWNER-DIRECTED from the
 * library's third job on -- a run of pair (i, j) to the ranks that own videos i and j (the epilogue of a video needs its
 * own pairs and nothing else, comparator.rs:583-588), sizes from the previous job's count matrix -- and a rank then holds,
 * and this call returns (downloaded on demand), the runs of ITS videos' pairs only; *num_runs of job_end stays the
 * total over all ranks.  NEEDLE_HIP_DIRECTED_RUNS=0: every job gathers every rank's runs on every rank, as before. */
enum NeedleError needle_hip_library_job_runs(const NeedleHipLibrary *library, int slot, const NeedleHipRun **runs,
                                             size_t *num_runs);
/* What the collectives of that job moved, as received per rank: bytes[0] hash rows (all-gather of the arena's blocks),
 * bytes[1] run lists (the gathered heads, or the owner-directed blocks + the count matrix; repeats included), bytes[2] per-video results of a sharded
 * epilogue; bytes[3] = scans repeated because a slab or a head overflowed (0 in the steady state). */
enum NeedleError needle_hip_library_job_comm_bytes(const NeedleHipLibrary *library, int slot, uint64_t bytes[4]);
/* Which forms the job that finished last in `slot` took: form[0] 1 = per-video epilogue on the device (0: host threads;
 * a device epilogue handed back to the host counts in needle_hip_epilogue_host_fallbacks), form[1] 1 = epilogue sharded by
 * video across ranks, form[2] 1 = runs exchanged owner-directed, form[3] the scan's form (needle_hip_scan_last_launch).
 * The epilogue goes to the device from 16 384 sequence pairs -- or, whatever the pair count, once the library's last
 * finished job found 16 384 runs or more per rank (stretches of silence, sustained chords: the run list, not the pair
 * count, is what the host form pays for; NEEDLE_HIP_DEVICE_EPILOGUE=0 / 1 forces either). */
enum NeedleError needle_hip_library_job_form(const NeedleHipLibrary *library, int slot, uint32_t form[4]);
/* Host threads this process uses for its parallel host phases (epilogue, file reads, upload staging): the CPUs usable
 * by the process (affinity, cgroup quota) divided by the rank processes of the node once a communicator is up
 * (LOCAL_WORLD_SIZE if the launcher exports it, else the world size); NEEDLE_HOST_THREADS overrides. */
int needle_hip_host_threads(void);

#ifdef __cplusplus
}
#endif

#endif /* NEEDLE_HIP_H */
