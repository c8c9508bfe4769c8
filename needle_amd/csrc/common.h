// Shared host-side declarations for libneedle_capi.so (MI355X-native needle analyze/search path).
#pragma once

#include <cstddef>
#include <cstdint>
#include <functional>
#include <string>
#include <vector>

#include "../../include/needle_hip.h"

struct ihipStream_t;  // hipStream_t = ihipStream_t * (hip_runtime_api.h), without pulling HIP into host-only files

namespace needle {

// ---- errors -------------------------------------------------------------------------------------------
// Mirrors needle::Error (needle/src/lib.rs:117-149) closely enough to reproduce the mapping to
// NeedleError done in needle-capi/src/lib.rs:121-134.
struct Status {
  NeedleError code = NeedleError_Ok;
  std::string message;
  bool ok() const { return code == NeedleError_Ok; }
  static Status Ok() { return Status{}; }
  static Status Make(NeedleError c, std::string m) { return Status{c, std::move(m)}; }
};

// Records the message for needle_hip_last_error_message() and, like lib.rs:124, prints
// "needle error: ..." on stderr; returns the code.
NeedleError report(const Status &s);
void set_last_error(const std::string &m);
const char *last_error();

// ---- std::time::Duration, kept as total nanoseconds ----------------------------------------------------
using ns_t = uint64_t;
constexpr ns_t kNanosPerSec = 1000000000ull;

ns_t duration_from_secs_f32(float s, bool *ok = nullptr);   // Duration::from_secs_f32 (round to nearest ns, ties even)
ns_t duration_from_secs_f64(double s, bool *ok = nullptr);  // Duration::from_secs_f64
float duration_as_secs_f32(ns_t d);                         // Duration::as_secs_f32
double duration_as_secs_f64(ns_t d);
ns_t duration_mul_f32(ns_t d, float rhs, bool *ok = nullptr);  // Duration::mul_f32

// ---- FrameHashes (needle/src/audio/data.rs:15-26,74-80) -------------------------------------------------
struct HashTs {
  uint32_t hash;
  ns_t ts;
};

struct FrameHashesData {
  std::vector<HashTs> opening;
  std::vector<HashTs> ending;
  ns_t hash_duration = 0;
  std::string md5;
};

Status frame_hashes_read(const std::string &path, FrameHashesData *out);         // data.rs:104-115
Status frame_hashes_write(const std::string &path, const FrameHashesData &fh);   // analyzer.rs:414-417
Status header_md5(const std::string &path, std::string *out);                    // util.rs:99-105
std::string md5_hex(const uint8_t *data, size_t n);
std::string with_extension(const std::string &path, const std::string &ext);     // Path::with_extension
std::string format_time(ns_t t);                                                  // util.rs:8-12
std::string format_f32_json(float v);                                             // serde_json/ryu f32

// Host CPUs this process may use: hardware threads capped by its affinity mask and by the cgroup CPU quota (a
// container can see every core of the machine and be granted a few CPUs of time).
unsigned usable_cpus();
// This process's share of them: usable_cpus() divided by the rank processes of this node (set_node_ranks(), called by
// the communicator with max(world, LOCAL_WORLD_SIZE); 1 without a communicator), at least 1.  Eight ranks on a 16-CPU
// share get two epilogue threads each instead of 8 x 16 threads on 16 CPUs.  NEEDLE_HOST_THREADS overrides it.
unsigned host_threads();
void set_node_ranks(unsigned ranks);

// ---- chromaprint-facing constants -------------------------------------------------
constexpr int kSampleRate = 11025;
constexpr int kFrameSize = 4096;
constexpr int kHop = 1365;
constexpr int kBands = 12;
constexpr int kFirTaps = 5;
constexpr int kMaxFilterWidth = 16;
constexpr int kItemLatency = (kFirTaps - 1) + (kMaxFilterWidth - 1);  // frames - items
constexpr int kDelayMs = 2600;        // chromaprint_get_delay_ms
constexpr int kItemDurationMs = 123;  // chromaprint_get_item_duration_ms

inline size_t num_frames(size_t samples) { return samples < (size_t)kFrameSize ? 0 : (samples - kFrameSize) / kHop + 1; }
inline size_t num_items(size_t samples) {
  size_t f = num_frames(samples);
  return f > (size_t)kItemLatency ? f - kItemLatency : 0;
}
inline size_t num_kept(size_t samples, uint32_t step) {
  size_t n = num_items(samples);
  return step ? (n + step - 1) / step : 0;
}

// analyzer.rs:293-318: timestamps of the kept items.  Returns false if step_by would be 0.
bool step_for_hash_duration(ns_t hash_duration, uint32_t *step);
void attach_timestamps(const uint32_t *kept, size_t n_kept, uint32_t step, bool has_seek, ns_t seek_to,
                       std::vector<HashTs> *out);

// ---- WAV (the only container this build decodes; FFmpeg is out of scope) ---------------------------------
struct WavData {
  int channels = 0;
  int sample_rate = 0;
  std::vector<int16_t> pcm;  // interleaved
};
Status wav_read(const std::string &path, WavData *out);  // the whole data chunk
struct WavInfo {
  int channels = 0, sample_rate = 0;
  int format = 0, bits = 0;  // 1 = integer PCM, 3 = IEEE float
  uint64_t data_offset = 0;  // byte offset of the first sample in the file
  uint64_t frames = 0;       // samples per channel in the data chunk
};
Status wav_probe(const std::string &path, WavInfo *out);  // reads chunk headers only
// frames [first, first+count) as interleaved s16 (count * channels values)
Status wav_read_frames(const std::string &path, const WavInfo &info, uint64_t first, uint64_t count, int16_t *dst);

// ---- GPU entry points used by the host classes (fingerprint.hip / search.hip) ----------------------------
struct StreamSpan {
  uint64_t pcm_off;     // offset into the PCM arena, in s16 values
  uint64_t num_values;  // interleaved values
  uint64_t item_off;    // offset into the item arena where kept items go
};

Status gpu_fingerprint_device(const int16_t *d_pcm, const std::vector<StreamSpan> &streams, int channels,
                              uint32_t step, uint32_t *d_items, bool sync, double *d_chroma_dbg = nullptr,
                              double *d_feat_dbg = nullptr, size_t descriptor_slot = 0, int pipe = -1,
                              uint32_t *zero_word = nullptr, bool *zeroed = nullptr);
// (zero_word: a device word the caller wants cleared behind the fingerprinting -- the run counter of the scan that
// follows -- by the last kernel of the call instead of a memset dispatch; *zeroed says whether that happened.)
// (pipe 0 / 1: the caller keeps two calls in flight, e.g. the two job slots of a library.  The f32 STFT of such a call
// runs on the CU-masked stream of hipctx.hip into workspace `pipe`, so that it overlaps whatever the previous call still
// has queued on the library stream; everything behind it stays on the library stream.  -1: everything on the library
// stream, one workspace.)
// Host PCM -> kept items in DEVICE memory (d_items + item_off[i]), uploads and kernels overlapped: the streams are
// uploaded in order on the upload stream and fingerprinted group by group (about NEEDLE_HIP_LAUNCH_GROUP_BYTES of
// PCM each) on the library stream as they land.  The PCM is not kept.  On return every copy out of host memory has
// executed; the kernels may still be running (library stream order).
Status gpu_fingerprint_streamed_device(const std::vector<const int16_t *> &pcm, const std::vector<size_t> &num_values,
                                       int channels, uint32_t step, uint32_t *d_items,
                                       const std::vector<uint64_t> &item_off);
// `rate` != 11025 routes the streams through the device resampler first (resample.hip).
Status gpu_fingerprint_host(const std::vector<const int16_t *> &pcm, const std::vector<size_t> &num_values,
                            int channels, uint32_t step, std::vector<std::vector<uint32_t>> *items,
                            int rate = kSampleRate);
// Same with the PCM produced by `read` (stream index, first value, value count -> interleaved s16 at dst, which is
// pinned host memory): up to `readers` threads call it concurrently for different segments while earlier segments
// are on their way to the device.  Host memory stays a fixed ring of slabs whatever the streams' lengths.
using PcmReader = std::function<Status(size_t stream, uint64_t first_value, uint64_t num_values, int16_t *dst)>;
// The upload alone (hipctx.hip): stream i goes to d_pcm + dev_off[i] (offsets in values, multiples of 8), copied on
// `stream` (NULL: the library stream).  On return the copies have executed.  `issued` (optional) is called on the
// issuing thread right after the last copy of a stream has been enqueued, streams in order (zero-length streams of
// the ring path are skipped): the caller's place to queue work behind an event of `stream`.
using StreamIssued = std::function<Status(size_t stream_index)>;
Status gpu_upload_pcm_streamed(const std::vector<size_t> &num_values, const std::vector<uint64_t> &dev_off,
                               const PcmReader &read, unsigned readers, int16_t *d_pcm, ::ihipStream_t *stream = nullptr,
                               const StreamIssued &issued = nullptr);
// Same from host pointers.  Pinned host memory is copied in place and small totals go as plain asynchronous copies
// (in both cases nothing need have executed on return, as with hipMemcpyAsync); large pageable totals go through the ring.
Status gpu_upload_pcm(const std::vector<const int16_t *> &pcm, const std::vector<size_t> &num_values,
                      const std::vector<uint64_t> &dev_off, int16_t *d_pcm, ::ihipStream_t *stream = nullptr,
                      const StreamIssued &issued = nullptr);
Status gpu_fingerprint_streamed(const std::vector<size_t> &num_values, const PcmReader &read, unsigned readers,
                                int channels, uint32_t step, std::vector<std::vector<uint32_t>> *items,
                                int rate = kSampleRate);
Status gpu_hamming_runs_device(const uint32_t *d_hashes, const NeedleHipSeq *seqs, size_t num_seqs,
                               const NeedleHipProblem *problems, size_t num_problems, uint32_t threshold,
                               NeedleHipRun *d_runs, uint32_t capacity, uint32_t *d_count, bool sync,
                               bool count_is_zero = false);
Status gpu_hamming_runs_host(const uint32_t *hashes, size_t num_hashes, const NeedleHipSeq *seqs, size_t num_seqs,
                             const NeedleHipProblem *problems, size_t num_problems, uint32_t threshold,
                             std::vector<NeedleHipRun> *runs);

// the hash arena of a search call in pinned host memory, and its upload enqueued ahead of the call (search.hip)
uint32_t *gpu_pinned_arena_acquire(size_t words);  // nullptr: none to be had (no device, or one is handed out already)
void gpu_pinned_arena_release(uint32_t *arena);
void gpu_prefetch_hashes(const uint32_t *hashes, size_t num_hashes);

// Sequence pairs from which the per-video epilogue runs on the device (epilogue.hip) instead of on host threads
// (NEEDLE_HIP_DEVICE_EPILOGUE=1 / 0 forces either): the library job and Comparator::run_with_frame_hashes alike.
constexpr uint64_t kDeviceEpiloguePairs = 1u << 14;
constexpr uint32_t kDeviceEpilogueRuns = 1u << 14;  // ... or from this many runs per rank in the library's last finished job

// Diagnostic (search.hip): cells/s of the band scan's 4-instruction cell on registers only, measured on this device.
Status gpu_int_valu_ceiling(double *cells_per_second);
Status gpu_scan_issued_evaluations(uint64_t *lane_evaluations, bool reset, uint64_t *head_survivors = nullptr);
void gpu_scan_last_launch(int32_t *form, uint64_t *matrix_products);  // needle_hip_scan_last_launch
// certified f32 first pass of the fingerprinter: {items, items recomputed in f64, chunks of frame pairs, chunks recomputed}
Status gpu_fingerprint_cert_stats(uint64_t out[4], bool reset);
// audit of that first pass: both transforms over the same PCM, every kept item compared on the device (fingerprint.hip)
Status gpu_fingerprint_audit_device(const int16_t *d_pcm, const std::vector<StreamSpan> &spans, int channels, uint32_t step,
                                    const uint32_t *d_items, uint64_t out[4], double *max_ratio, double *max_sigma);

// ---- resampler front-end (resample.hip) -------------------------------------------------------------------
struct ResampleSpan {
  uint64_t in_off;   // offset into the input arena, in s16 values
  uint64_t n_in;     // samples per channel
  uint64_t out_off;  // offset into the output arena, in samples
};
size_t resample_out_len(size_t n_in, int rate);
Status gpu_resample_device(const int16_t *d_in, const std::vector<ResampleSpan> &spans, int channels, int rate,
                           int16_t *d_out, bool sync);
Status gpu_resample_host(const std::vector<const int16_t *> &pcm, const std::vector<size_t> &num_values, int channels,
                         int rate, std::vector<std::vector<int16_t>> *out);

}  // namespace needle
