// NeedleHipLibrary: an HBM-resident analyze+search job (include/needle_hip.h, last section).
//
// The reference's library path is Analyzer::run -> Comparator::run_with_frame_hashes
// (needle/src/audio/analyzer.rs:425, comparator.rs:524): analyze every video, then search every pair.
// Here the PCM of the videos a rank owns stays in HBM, hashes are written straight into a padded device
// arena u32[rows][stride] (one row per video and search window: row v*R = opening, v*R+1 = ending when
// endings are enabled), the pair search reads that arena in place, and only the short run list crosses
// PCIe — the runs carry their simhashes, so the host epilogue needs timestamps only.  Rows computed by
// other ranks are filled by the caller with one all-gather over contiguous row blocks (RCCL over xGMI);
// pairs are sharded by index in the lexicographic pair list.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <new>
#include <tuple>

#include "comm.h"
#include "epilogue.h"
#include "hipctx.h"
#include "needle_core.h"

struct FrameHashes;
struct NeedleAudioComparator;

namespace needle {
const Comparator &comparator_of(const NeedleAudioComparator *c);
FrameHashes *make_frame_hashes(FrameHashesData &&d);
void fill_c_result(const VideoResult &v, NeedleHipSearchResult *r);
}  // namespace needle

using namespace needle;

namespace {
struct Window {              // one search window of one video
  size_t values = 0;         // interleaved s16 values
  uint64_t pcm_off = ~0ull;  // offset into d_pcm; ~0 when this rank does not hold the PCM
  uint32_t kept = 0;         // hashes kept (every step-th raw item)
  ns_t seek = 0;             // added to every timestamp (ending window, analyzer.rs:314-318)
};
}  // namespace

struct NeedleHipLibrary {
  size_t n = 0;
  float opening_pct = DEFAULT_OPENING_SEARCH_PERCENTAGE;
  float ending_pct = DEFAULT_ENDING_SEARCH_PERCENTAGE;
  bool endings = false;
  ns_t hash_duration = 0;
  uint32_t step = 0;
  int channels = 1;
  bool have_pcm = false;      // windows and arena are set up (set_pcm or stream_pcm)
  bool pcm_resident = false;  // set_pcm: the PCM stays in HBM and analyze can be repeated
  std::vector<Window> win;  // [video * regions() + region]
  size_t stride = 0;
  DeviceBuffer<int16_t> d_pcm;
  DeviceBuffer<uint32_t> d_arena;
  uint32_t *arena = nullptr;  // d_arena.ptr, or caller-owned memory adopted with needle_hip_library_use_hash_arena
  std::vector<std::vector<HashTs>> ts_cache;  // un-seeked timestamps by kept length
  std::vector<uint32_t> min_len;              // per row, for the durations below
  ns_t min_len_for[2] = {~0ull, ~0ull};
  // the pair table of the last search, reused while (first pair, pair count, regions, min_len) stay the same: a
  // library has ~n^2 / 2 pairs and a job that is repeated should not rebuild millions of descriptors
  std::vector<NeedleHipProblem> problems;
  size_t problems_for[3] = {~(size_t)0, 0, 0};
  // Timestamps for the epilogue (the hashes stay in HBM).  A window's timestamps are a function of its length and seek
  // only, so videos of equal geometry SHARE one shell: at library scale the epilogue's four timestamp reads per run
  // then hit one 87 KB array instead of 2000 copies of it (174 MB of cache misses: 58 -> 15 ms of the heap-entry phase).
  std::vector<FrameHashesData> shells;        // distinct geometries
  std::vector<uint32_t> shell_of;             // video -> shells[]
  // double-buffered asynchronous run-list download (needle_hip_library_fetch_runs_begin / _end)
  struct Fetch {
    void *host = nullptr;  // pinned: u32 count, then max_runs NeedleHipRun
    uint32_t max_runs = 0;
    hipEvent_t ready = nullptr, done = nullptr;  // search enqueued (library stream) / copies finished (download stream)
    const NeedleHipRun *d_runs = nullptr;        // what the pending download reads
    const uint32_t *d_count = nullptr;
    bool pending = false;
  } fetch[2];
  // One slot of the job pipeline (needle_hip_library_job_begin / _end): the run slabs of all ranks, device side and
  // pinned host side.  Rank r's slab = 32-byte header (word 0: runs found) + slab_runs runs; the scan writes straight
  // into this rank's slab and the all-gather fills in the others in place.
  struct Job {
    DeviceBuffer<uint8_t> d_slabs;
    DeviceBuffer<uint8_t> d_heads;  // world > 1: [world][head_bytes], what the all-gather of run lists fills
    uint32_t slab_runs = 0, head_runs = 0;
    int world = 0;
    void *host = nullptr;
    size_t host_bytes = 0;
    hipEvent_t searched = nullptr, done = nullptr;
    bool pending = false;
    std::vector<NeedleHipRun> merged;
    // device epilogue (epilogue.hip): per-video results + failure count land here (pinned) behind the run download
    void *host_results = nullptr;
    size_t host_results_bytes = 0;
    bool device_epilogue = false, sharded = false;
    bool shape_fixed = false;  // `sharded` was decided when the job was enqueued (device epilogue requested): job_end keeps it even if the
                               // device form then failed on THIS rank alone -- the other ranks enter the collective that decision implies
    const NeedleHipRun *last_runs = nullptr;  // the complete run list of the job that finished last in this slot
    size_t last_total = 0;
    // Owner-directed run exchange (round 6; world > 1 with the sharded device epilogue): a run of pair (i, j) travels to
    // the owners of videos i and j only (comparator.rs:583-588: a video's epilogue needs its own pairs), not to every rank.
    // d_dir_send = this rank's runs sorted into one block per destination (epilogue.h DirectPlan), d_dir_recv = the blocks
    // received, in rank order, = the device epilogue's segments.  Block sizes are host values on every rank: cap[r * world + q]
    // comes from the count matrix of the library's last finished job (every rank has all of it: d_dir_counts, one row of
    // [slab count, runs directed to rank 0, 1, ...] per rank, all-gathered) plus a margin; a count beyond its block is the
    // head overflow's case (job_end: sizes grow, the exchange is repeated).  A library's first job has no matrix yet: it
    // counts, and travels as heads.
    DeviceBuffer<uint8_t> d_dir_send, d_dir_recv;
    DeviceBuffer<uint32_t> d_dir_counts;
    void *host_counts = nullptr;
    size_t host_counts_bytes = 0;
    std::vector<uint32_t> cap;      // of this job's exchange (empty: this job travels as heads)
    bool ran_device_epilogue = false, ran_sharded = false;  // what job_end found (needle_hip_library_job_form)
    uint32_t scan_form = 0;
    bool counted = false;           // host_counts holds this job's count matrix
    bool directed = false;          // this job's runs travelled owner-directed
    std::vector<NeedleHipRun> fetched;  // directed: the received runs, downloaded on demand (job_runs, a host fallback)
    bool fetched_valid = false;
    static size_t count_words(int world) { return ((size_t)world + 1 + 3) & ~(size_t)3; }
    uint64_t comm_bytes[4] = {0, 0, 0, 0};  // received per rank in this job: hash rows, run heads, results; scans repeated
    size_t slab_bytes() const { return kSlabHeader + (size_t)slab_runs * sizeof(NeedleHipRun); }
    size_t head_bytes() const { return kSlabHeader + (size_t)head_runs * sizeof(NeedleHipRun); }
  } job[2];
  static constexpr size_t kSlabHeader = 32;
  uint32_t slab_runs = 0;      // per-rank run capacity of the next job (grows on overflow)
  uint32_t last_max_count = 0;  // largest per-rank run count of the last finished job: sizes the one-trip download
  std::vector<uint32_t> dir_counts;  // [world][world]: runs rank r directed to rank q in the last finished job (Job::cap's source)
  int dir_world = 0;
  bool dir_unsupported = false;      // the communicator has no point-to-point transfers: every job travels as heads
  bool dir_cap_forced = false;       // (tests: NEEDLE_HIP_TEST_DIRECTED_CAP applied once)
  size_t arena_rows = 0;       // rows the arena was allocated with
  const uint32_t *count_zeroed = nullptr;  // a run counter the last kernel of this job's analyze has just cleared (job_begin)
  // what the device epilogue needs to know about the arena's rows (built once per geometry: plan_windows clears them)
  std::vector<uint32_t> row_len, row_ts;
  std::vector<uint64_t> row_seek, ts_tables;
  void ensure_row_tables() {
    if (row_len.size() == rows()) return;
    row_len.assign(rows(), 0);
    row_ts.assign(rows(), 0);
    row_seek.assign(rows(), 0);
    ts_tables.clear();
    std::map<uint32_t, uint32_t> offset_of;  // kept length -> offset of its un-seeked timestamps
    for (size_t row = 0; row < rows(); row++) {
      const Window &w = win[row];
      auto it = offset_of.find(w.kept);
      if (it == offset_of.end()) {
        it = offset_of.emplace(w.kept, (uint32_t)ts_tables.size()).first;
        for (const HashTs &h : timestamps(w.kept)) ts_tables.push_back(h.ts);
      }
      row_len[row] = w.kept;
      row_ts[row] = it->second;
      row_seek[row] = w.seek;
    }
  }
  ~NeedleHipLibrary() {
    for (Fetch &f : fetch) {
      if (f.host) (void)hipHostFree(f.host);
      if (f.done) (void)hipEventDestroy(f.done);
      if (f.ready) (void)hipEventDestroy(f.ready);
    }
    for (Job &j : job) {
      if (j.host_results) (void)hipHostFree(j.host_results);
      if (j.host_counts) (void)hipHostFree(j.host_counts);
      if (j.host) (void)hipHostFree(j.host);
      if (j.done) (void)hipEventDestroy(j.done);
      if (j.searched) (void)hipEventDestroy(j.searched);
    }
  }

  size_t regions() const { return endings ? 2 : 1; }
  size_t rows() const { return n * regions(); }
  // Sharding of the fingerprinting across ranks (analyzer.rs:437-445 across GPUs).  The unit is not the video but the
  // HASH: the arena u32[rows][stride] is cut, as one flat array, into world equal blocks -- stride is a multiple of 64
  // chosen so that rows * stride divides by 64 * world -- and a rank fingerprints exactly the hashes of its block:
  // for every row the block meets, the columns it meets, computed from the SUB-WINDOW of that row's PCM those hashes
  // depend on (hash k of a row is a function of frames k * step .. k * step + 19 only; blocks start on multiples of 64
  // hashes, so a sub-window starts on an even frame and the two-frames-per-transform pairing is the whole window's).
  // 28 episodes on 8 ranks are then 3.5 rows each instead of 7 x 4 + 0; one in-place all-gather of equal blocks
  // fills the arena as before.
  size_t flat_block(int world) const { return rows() * stride / (size_t)std::max(world, 1); }
  bool flat_shardable(int world) const { return world <= 1 || (stride % 64 == 0 && (rows() * (stride / 64)) % (size_t)world == 0); }
  void ensure_shells() {
    if (shell_of.size() == n) return;
    shells.clear();
    shell_of.assign(n, 0);
    const size_t R = regions();
    std::map<std::tuple<uint32_t, ns_t, uint32_t, ns_t>, uint32_t> seen;
    for (size_t v = 0; v < n; v++) {
      const Window &wo = win[v * R];
      const Window we = endings ? win[v * R + 1] : Window{};
      const auto key = std::make_tuple(wo.kept, wo.seek, we.kept, we.seek);
      auto it = seen.find(key);
      if (it == seen.end()) {
        FrameHashesData d;
        d.opening = window_timestamps(wo);
        if (endings) d.ending = window_timestamps(we);
        d.hash_duration = hash_duration;
        it = seen.emplace(key, (uint32_t)shells.size()).first;
        shells.push_back(std::move(d));
      }
      shell_of[v] = it->second;
    }
  }
  std::vector<const FrameHashesData *> shell_pointers() {
    ensure_shells();
    std::vector<const FrameHashesData *> fh(n);
    for (size_t v = 0; v < n; v++) fh[v] = &shells[shell_of[v]];
    return fh;
  }

  const std::vector<HashTs> &timestamps(uint32_t k) {
    if (ts_cache.size() <= k) ts_cache.resize(k + 1);
    if (ts_cache[k].empty() && k) {
      std::vector<uint32_t> zeros(k, 0);
      attach_timestamps(zeros.data(), k, step, false, 0, &ts_cache[k]);
    }
    return ts_cache[k];
  }
  std::vector<HashTs> window_timestamps(const Window &w) {
    std::vector<HashTs> ts = timestamps(w.kept);
    if (w.seek)
      for (HashTs &h : ts) h.ts += w.seek;
    return ts;
  }
};

namespace {
template <typename F>
NeedleError guarded(F &&f) {
  try {
    return f();
  } catch (const std::bad_alloc &) {
    return report(Status::Make(NeedleError_Unknown, "out of memory"));
  } catch (...) {
    return report(Status::Make(NeedleError_Unknown, "internal error"));
  }
}
}  // namespace

extern "C" {

enum NeedleError needle_hip_library_new(size_t num_videos, float opening_search_percentage, float hash_duration,
                                        NeedleHipLibrary **output) {
  if (!output) return NeedleError_NullArgument;
  if (num_videos == 0) return NeedleError_InvalidArgument;
  if (!(hash_duration > 0.0f)) return NeedleError_AnalyzerInvalidHashDuration;
  return guarded([&]() -> NeedleError {
    bool ok = true;
    const ns_t hd = duration_from_secs_f32(hash_duration, &ok);
    uint32_t step = 0;
    if (!ok || !step_for_hash_duration(hd, &step)) return NeedleError_AnalyzerInvalidHashDuration;
    auto *lib = new NeedleHipLibrary();
    lib->n = num_videos;
    lib->opening_pct = opening_search_percentage;
    lib->hash_duration = hd;
    lib->step = step;
    *output = lib;
    return NeedleError_Ok;
  });
}

void needle_hip_library_free(NeedleHipLibrary *library) { delete library; }

enum NeedleError needle_hip_library_include_endings(NeedleHipLibrary *lib, float ending_search_percentage) {
  if (!lib) return NeedleError_NullArgument;
  if (lib->have_pcm) return NeedleError_InvalidArgument;  // must precede set_pcm
  lib->endings = true;
  lib->ending_pct = ending_search_percentage;
  return NeedleError_Ok;
}

}  // extern "C"

namespace {
// Search windows of every video (analyzer.rs:378,390) from the stream lengths, arena geometry, and -- for the videos
// whose PCM this rank holds -- where each window starts in the caller's buffer.  `resident`: the windows get offsets
// into the device PCM arena (set_pcm); otherwise nothing of the PCM is kept (stream_pcm).
Status plan_windows(NeedleHipLibrary *lib, const int16_t *const *pcm, const size_t *num_values, int channels, bool resident,
                    std::vector<const int16_t *> *src, std::vector<size_t> *len, std::vector<uint64_t> *dst,
                    std::vector<uint64_t> *rows_of_src, uint64_t *total_values) {
  Status s = ensure_device();
  if (!s.ok()) return s;
  lib->channels = channels;
  const size_t R = lib->regions();
  lib->win.assign(lib->rows(), Window{});
  uint64_t total = 0;
  uint32_t max_kept = 0;
  std::vector<size_t> first_sample(lib->rows(), 0);
  for (size_t v = 0; v < lib->n; v++) {
    const size_t samples = num_values[v] / (size_t)channels;
    size_t open_samples = 0, end_first = 0;
    ns_t seek = 0;
    s = Analyzer::windows(samples, kSampleRate, lib->opening_pct, lib->ending_pct, &open_samples, &end_first, &seek);
    if (!s.ok()) return s;
    for (size_t r = 0; r < R; r++) {
      Window &w = lib->win[v * R + r];
      const size_t count = r == 0 ? open_samples : samples - end_first;
      first_sample[v * R + r] = r == 0 ? 0 : end_first;
      w.values = count * (size_t)channels;
      w.kept = (uint32_t)num_kept(count, lib->step);
      w.seek = r == 0 ? 0 : seek;
      max_kept = std::max(max_kept, w.kept);
      if (pcm[v] && resident) {
        w.pcm_off = total;
        total += (w.values + 1) & ~(uint64_t)1;
      }
    }
  }
  // rows start 256-byte aligned; with a communicator, rows * stride also divides into world blocks of whole 64-hash tiles
  size_t tiles_per_row = std::max<size_t>(1, ((size_t)max_kept + 63) / 64);
  while (comm_world() > 1 && (lib->rows() * tiles_per_row) % (size_t)comm_world()) tiles_per_row++;
  size_t stride = 64 * tiles_per_row;
  const size_t arena_rows = lib->rows();
  // NeedleHipSeq.offset and NeedleHipProblem.tag are 32-bit on the device
  if ((uint64_t)arena_rows * stride > UINT32_MAX || (uint64_t)pair_count(lib->n) * R > UINT32_MAX)
    return Status::Make(NeedleError_InvalidArgument, "library too large for one job: more than 2^32 arena hashes or sequence pairs");
  const bool own_arena = lib->arena == lib->d_arena.ptr;
  if (!lib->arena || (own_arena && (stride != lib->stride || arena_rows != lib->arena_rows))) {
    if (!(s = lib->d_arena.reserve(arena_rows * stride)).ok()) return s;
    lib->arena = lib->d_arena.ptr;
    lib->arena_rows = arena_rows;
    lib->stride = stride;
    if (hipMemsetAsync(lib->arena, 0, arena_rows * stride * sizeof(uint32_t), library_stream()) != hipSuccess)
      return Status::Make(NeedleError_Unknown, "hipMemset failed");
  } else if (!own_arena && (stride > lib->stride || arena_rows > lib->arena_rows)) {
    return Status::Make(NeedleError_InvalidArgument, "the adopted hash arena is too small for these videos");
  }
  for (size_t v = 0; v < lib->n; v++) {
    for (size_t r = 0; r < R; r++) {
      const Window &w = lib->win[v * R + r];
      if (!pcm[v] || !w.values) continue;
      src->push_back(pcm[v] + first_sample[v * R + r] * (size_t)channels);
      len->push_back(w.values);
      if (dst) dst->push_back(w.pcm_off);
      if (rows_of_src) rows_of_src->push_back(v * R + r);
    }
  }
  if (total_values) *total_values = total;
  lib->min_len.clear();
  lib->row_len.clear();
  lib->shells.clear();
  lib->shell_of.clear();
  lib->problems_for[0] = ~(size_t)0;
  return Status::Ok();
}
}  // namespace

extern "C" {

enum NeedleError needle_hip_library_set_pcm(NeedleHipLibrary *lib, const int16_t *const *pcm, const size_t *num_values,
                                            int channels) {
  if (!lib || !pcm || !num_values) return NeedleError_NullArgument;
  if (channels != 1 && channels != 2) return NeedleError_InvalidArgument;
  return guarded([&]() -> NeedleError {
    std::vector<const int16_t *> src;
    std::vector<size_t> len;
    std::vector<uint64_t> dst;
    uint64_t total = 0;
    Status s = plan_windows(lib, pcm, num_values, channels, true, &src, &len, &dst, nullptr, &total);
    if (!s.ok()) return report(s);
    if (!(s = lib->d_pcm.reserve(std::max<uint64_t>(total, 1))).ok()) return report(s);
    s = gpu_upload_pcm(src, len, dst, lib->d_pcm.ptr);
    // also on the error path: copies already enqueued read the caller's buffers asynchronously
    const bool drained = hipStreamSynchronize(library_stream()) == hipSuccess;
    if (!s.ok()) return report(s);
    if (!drained) return report(Status::Make(NeedleError_Unknown, "PCM upload failed"));
    lib->have_pcm = true;
    lib->pcm_resident = true;
    return NeedleError_Ok;
  });
}

enum NeedleError needle_hip_library_set_pcm_device(NeedleHipLibrary *lib, const int16_t *const *d_pcm,
                                                   const size_t *num_values, int channels) {
  if (!lib || !d_pcm || !num_values) return NeedleError_NullArgument;
  if (channels != 1 && channels != 2) return NeedleError_InvalidArgument;
  return guarded([&]() -> NeedleError {
    std::vector<const int16_t *> src;
    std::vector<size_t> len;
    std::vector<uint64_t> dst;
    uint64_t total = 0;
    Status s = plan_windows(lib, d_pcm, num_values, channels, true, &src, &len, &dst, nullptr, &total);
    if (!s.ok()) return report(s);
    if (!(s = lib->d_pcm.reserve(std::max<uint64_t>(total, 1))).ok()) return report(s);
    hipStream_t stream = library_stream();
    for (size_t i = 0; i < src.size(); i++)  // the search windows only, device to device, in stream order
      if (hipMemcpyAsync(lib->d_pcm.ptr + dst[i], src[i], len[i] * sizeof(int16_t), hipMemcpyDeviceToDevice, stream) != hipSuccess)
        return report(Status::Make(NeedleError_Unknown, "device-to-device PCM copy failed"));
    if (hipStreamSynchronize(stream) != hipSuccess) return report(Status::Make(NeedleError_Unknown, "PCM copy failed"));
    lib->have_pcm = true;
    lib->pcm_resident = true;
    return NeedleError_Ok;
  });
}

enum NeedleError needle_hip_library_stream_pcm(NeedleHipLibrary *lib, const int16_t *const *pcm, const size_t *num_values,
                                               int channels) {
  if (!lib || !pcm || !num_values) return NeedleError_NullArgument;
  if (channels != 1 && channels != 2) return NeedleError_InvalidArgument;
  return guarded([&]() -> NeedleError {
    std::vector<const int16_t *> src;
    std::vector<size_t> len;
    std::vector<uint64_t> rows;
    const auto t0 = std::chrono::steady_clock::now();
    Status s = plan_windows(lib, pcm, num_values, channels, false, &src, &len, nullptr, &rows, nullptr);
    if (!s.ok()) return report(s);
    for (uint64_t &r : rows) r *= lib->stride;  // kept items of a window go straight to its arena row
    const auto t1 = std::chrono::steady_clock::now();
    if (!(s = gpu_fingerprint_streamed_device(src, len, channels, lib->step, lib->arena, rows)).ok()) return report(s);
    if (getenv("NEEDLE_HIP_TRACE"))
      std::fprintf(stderr, "[needle_hip] stream_pcm: windows planned in %.2f ms, %zu windows streamed in %.2f ms\n",
                   std::chrono::duration<double, std::milli>(t1 - t0).count(), src.size(),
                   std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count());
    lib->have_pcm = true;
    lib->pcm_resident = false;
    return NeedleError_Ok;
  });
}

enum NeedleError needle_hip_library_analyze(NeedleHipLibrary *lib, size_t first, size_t count, bool sync) {
  if (!lib) return NeedleError_NullArgument;
  if (!lib->have_pcm || first > lib->n || count > lib->n - first) return NeedleError_InvalidArgument;
  return guarded([&]() -> NeedleError {
    std::vector<StreamSpan> spans;
    const size_t R = lib->regions();
    for (size_t row = first * R; row < (first + count) * R; row++) {
      const Window &w = lib->win[row];
      if (w.pcm_off == ~0ull)
        return report(Status::Make(NeedleError_InvalidArgument, "video " + std::to_string(row / R) + " has no PCM on this rank"));
      spans.push_back(StreamSpan{w.pcm_off, w.values, (uint64_t)row * lib->stride});
    }
    Status s = gpu_fingerprint_device(lib->d_pcm.ptr, spans, lib->channels, lib->step, lib->arena, sync);
    return s.ok() ? NeedleError_Ok : report(s);
  });
}

enum NeedleError needle_hip_library_hash_arena(NeedleHipLibrary *lib, uint32_t **d_arena, size_t *stride) {
  if (!lib || !d_arena || !stride) return NeedleError_NullArgument;
  if (!lib->have_pcm) return NeedleError_InvalidArgument;
  *d_arena = lib->arena;
  *stride = lib->stride;
  return NeedleError_Ok;
}

size_t needle_hip_library_rows_per_video(const NeedleHipLibrary *lib) { return lib ? lib->regions() : 0; }

enum NeedleError needle_hip_library_use_hash_arena(NeedleHipLibrary *lib, uint32_t *d_arena, size_t rows, size_t stride) {
  if (!lib || !d_arena) return NeedleError_NullArgument;
  if (!lib->have_pcm || rows < lib->rows() || stride < lib->stride) return NeedleError_InvalidArgument;
  if ((uint64_t)rows * stride > UINT32_MAX) return NeedleError_InvalidArgument;
  lib->arena = d_arena;
  lib->arena_rows = rows;
  lib->stride = stride;
  lib->d_arena.release();
  return NeedleError_Ok;
}

size_t needle_hip_library_num_pairs(const NeedleHipLibrary *lib) { return lib ? pair_count(lib->n) : 0; }

enum NeedleError needle_hip_library_search(NeedleHipLibrary *lib, const struct NeedleAudioComparator *comparator,
                                           size_t first_pair, size_t num_pairs, NeedleHipRun *d_runs,
                                           uint32_t capacity, uint32_t *d_count, bool sync) {
  if (!lib || !comparator || !d_runs || !d_count) return NeedleError_NullArgument;
  const size_t np = pair_count(lib->n);
  if (!lib->have_pcm || first_pair > np || num_pairs > np - first_pair) return NeedleError_InvalidArgument;
  return guarded([&]() -> NeedleError {
    const auto t_enter = std::chrono::steady_clock::now();
    const Comparator &cmp = comparator_of(comparator);
    if (cmp.include_endings() && !lib->endings)  // comparator.rs:271-273
      return report(Status::Make(NeedleError_Unknown, "no ending hash data present"));
    const size_t R = lib->regions(), Rc = cmp.include_endings() ? 2 : 1;
    std::vector<NeedleHipSeq> seqs(lib->rows());
    for (size_t row = 0; row < lib->rows(); row++)
      seqs[row] = NeedleHipSeq{(uint32_t)(row * lib->stride), lib->win[row].kept};
    // per-row minimum run length for the duration tests (it depends on the kept length only: the seek offset
    // of an ending window cancels in ts[i] - ts[i-L])
    if (lib->min_len.size() != lib->rows() || lib->min_len_for[0] != cmp.min_opening_duration() ||
        lib->min_len_for[1] != cmp.min_ending_duration()) {
      lib->min_len.assign(lib->rows(), 0);
      for (size_t row = 0; row < lib->rows(); row++) {
        const bool opening = row % R == 0;
        lib->min_len[row] = row >= R && lib->win[row].kept == lib->win[row - R].kept
                                ? lib->min_len[row - R]
                                : cmp.min_run_length_for(lib->timestamps(lib->win[row].kept), opening);
      }
      lib->min_len_for[0] = cmp.min_opening_duration();
      lib->min_len_for[1] = cmp.min_ending_duration();
      lib->problems_for[0] = ~(size_t)0;
    }
    std::vector<NeedleHipProblem> &problems = lib->problems;
    const bool cached = lib->problems_for[0] == first_pair && lib->problems_for[1] == num_pairs && lib->problems_for[2] == Rc;
    if (!cached) {
      lib->problems_for[0] = ~(size_t)0;  // not valid until it is complete
      problems.clear();
      problems.reserve(num_pairs * Rc);
      size_t i = 0, j = 0;
      if (num_pairs) pair_at(lib->n, first_pair, &i, &j);
      for (size_t p = first_pair; p < first_pair + num_pairs; p++) {
        for (size_t r = 0; r < Rc; r++) {
          if (r == 1 && (lib->win[i * R + 1].kept == 0 || lib->win[j * R + 1].kept == 0))  // comparator.rs:271-273
            return report(Status::Make(NeedleError_Unknown, "no ending hash data present"));
          const uint32_t a = lib->min_len[i * R + r], b = lib->min_len[j * R + r];
          if (a == 0 || b == 0) continue;
          problems.push_back(NeedleHipProblem{(uint32_t)(i * R + r), (uint32_t)(j * R + r), std::max(a, b),
                                              (uint32_t)(p * Rc + r)});
        }
        if (++j == lib->n) j = ++i + 1;  // next pair in i-major order
      }
      lib->problems_for[0] = first_pair;
      lib->problems_for[1] = num_pairs;
      lib->problems_for[2] = Rc;
    }
    // a download that still reads this run buffer (on the download stream) has to finish before it is overwritten
    for (NeedleHipLibrary::Fetch &f : lib->fetch)
      if (f.pending && (f.d_runs == d_runs || f.d_count == d_count) &&
          hipStreamWaitEvent(library_stream(), f.done, 0) != hipSuccess)
        return report(Status::Make(NeedleError_Unknown, "stream wait failed"));
    const auto t_built = std::chrono::steady_clock::now();
    const bool count_is_zero = lib->count_zeroed == d_count;
    lib->count_zeroed = nullptr;
    Status s = gpu_hamming_runs_device(lib->arena, seqs.data(), seqs.size(), problems.data(), problems.size(),
                                       cmp.hash_match_threshold(), d_runs, capacity, d_count, sync, count_is_zero);
    if (getenv("NEEDLE_HIP_TRACE"))
      std::fprintf(stderr, "[needle_hip] search enqueue: %zu problems built in %.2f ms, launched in %.2f ms\n",
                   problems.size(), std::chrono::duration<double, std::milli>(t_built - t_enter).count(),
                   std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_built).count());
    return s.ok() ? NeedleError_Ok : report(s);
  });
}

enum NeedleError needle_hip_library_fetch_runs_begin(NeedleHipLibrary *lib, int slot, const NeedleHipRun *d_runs,
                                                     const uint32_t *d_count, uint32_t max_runs) {
  if (!lib || !d_runs || !d_count) return NeedleError_NullArgument;
  if (slot < 0 || slot > 1 || max_runs == 0) return NeedleError_InvalidArgument;
  return guarded([&]() -> NeedleError {
    NeedleHipLibrary::Fetch &f = lib->fetch[slot];
    if (f.pending) return report(Status::Make(NeedleError_InvalidArgument, "fetch slot still pending"));
    const size_t bytes = 16 + (size_t)max_runs * sizeof(NeedleHipRun);
    if (f.max_runs < max_runs) {
      if (f.host) (void)hipHostFree(f.host);
      f.host = nullptr;
      if (hipHostMalloc(&f.host, bytes, hipHostMallocDefault) != hipSuccess)
        return report(Status::Make(NeedleError_Unknown, "pinned allocation failed"));
      f.max_runs = max_runs;
    }
    if (!f.done && hipEventCreateWithFlags(&f.done, hipEventDisableTiming) != hipSuccess)
      return report(Status::Make(NeedleError_Unknown, "event creation failed"));
    if (!f.ready && hipEventCreateWithFlags(&f.ready, hipEventDisableTiming) != hipSuccess)
      return report(Status::Make(NeedleError_Unknown, "event creation failed"));
    // the copies run on the download stream, behind the search that was just enqueued on the library stream
    hipStream_t stream = library_stream(), down = download_stream();
    if (hipEventRecord(f.ready, stream) != hipSuccess || hipStreamWaitEvent(down, f.ready, 0) != hipSuccess ||
        hipMemcpyAsync(f.host, d_count, sizeof(uint32_t), hipMemcpyDeviceToHost, down) != hipSuccess ||
        hipMemcpyAsync(static_cast<char *>(f.host) + 16, d_runs, (size_t)max_runs * sizeof(NeedleHipRun),
                       hipMemcpyDeviceToHost, down) != hipSuccess ||
        hipEventRecord(f.done, down) != hipSuccess)
      return report(Status::Make(NeedleError_Unknown, "asynchronous run download failed"));
    f.d_runs = d_runs;
    f.d_count = d_count;
    f.pending = true;
    return NeedleError_Ok;
  });
}

enum NeedleError needle_hip_library_fetch_runs_end(NeedleHipLibrary *lib, int slot, const NeedleHipRun **runs,
                                                   uint32_t *num_runs) {
  if (!lib || !runs || !num_runs) return NeedleError_NullArgument;
  if (slot < 0 || slot > 1 || !lib->fetch[slot].pending) return NeedleError_InvalidArgument;
  return guarded([&]() -> NeedleError {
    NeedleHipLibrary::Fetch &f = lib->fetch[slot];
    if (hipEventSynchronize(f.done) != hipSuccess) return report(Status::Make(NeedleError_Unknown, "run download failed"));
    f.pending = false;
    *num_runs = *static_cast<const uint32_t *>(f.host);  // total found; > max_runs means the list was truncated
    *runs = reinterpret_cast<const NeedleHipRun *>(static_cast<const char *>(f.host) + 16);
    return NeedleError_Ok;
  });
}

enum NeedleError needle_hip_library_finalize(NeedleHipLibrary *lib, const struct NeedleAudioComparator *comparator,
                                             const NeedleHipRun *runs, size_t num_runs,
                                             NeedleHipSearchResult *results) {
  if (!lib || !comparator || (!runs && num_runs) || !results) return NeedleError_NullArgument;
  if (!lib->have_pcm) return NeedleError_InvalidArgument;
  return guarded([&]() -> NeedleError {
    const Comparator &cmp = comparator_of(comparator);
    const std::vector<const FrameHashesData *> fh = lib->shell_pointers();
    std::vector<VideoResult> res;
    const auto t0 = std::chrono::steady_clock::now();
    Status s = cmp.results_from_runs(fh, runs, num_runs, false, false, false, &res);
    if (getenv("NEEDLE_HIP_TRACE"))
      std::fprintf(stderr, "[needle_hip] epilogue %zu runs: %.1f us\n", num_runs,
                   std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
    if (!s.ok()) return report(s);
    for (size_t v = 0; v < lib->n; v++) fill_c_result(res[v], &results[v]);
    return NeedleError_Ok;
  });
}

enum NeedleError needle_hip_library_frame_hashes(NeedleHipLibrary *lib, size_t index, FrameHashes **output) {
  if (!lib || !output) return NeedleError_NullArgument;
  if (!lib->have_pcm || index >= lib->n) return NeedleError_InvalidArgument;
  return guarded([&]() -> NeedleError {
    const size_t R = lib->regions();
    FrameHashesData fh;
    hipStream_t stream = library_stream();
    for (size_t r = 0; r < R; r++) {
      const Window &w = lib->win[index * R + r];
      std::vector<uint32_t> host(w.kept);
      if (!host.empty() &&
          (hipMemcpyAsync(host.data(), lib->arena + (index * R + r) * lib->stride, host.size() * sizeof(uint32_t),
                          hipMemcpyDeviceToHost, stream) != hipSuccess ||
           hipStreamSynchronize(stream) != hipSuccess))
        return report(Status::Make(NeedleError_Unknown, "hash arena download failed"));
      std::vector<HashTs> ts = lib->window_timestamps(w);
      for (size_t k = 0; k < host.size(); k++) ts[k].hash = host[k];
      (r == 0 ? fh.opening : fh.ending) = std::move(ts);
    }
    fh.hash_duration = lib->hash_duration;
    *output = make_frame_hashes(std::move(fh));
    return NeedleError_Ok;
  });
}

}  // extern "C"

// ---- the whole job, across the communicator ------------------------------------------------------------------------
namespace {

// The rows and columns of the arena that rank `rank`'s flat block meets: f(row, first column, end column), columns
// clipped to the row's kept hashes (the rest of a row is padding).
template <typename F>
void for_rows_of_block(const NeedleHipLibrary *lib, int world, int rank, F &&f) {
  const size_t B = lib->flat_block(world), begin = (size_t)rank * B, end = begin + B;
  if (!lib->stride) return;
  for (size_t row = begin / lib->stride; row < lib->rows() && row * lib->stride < end; row++) {
    const size_t r0 = row * lib->stride;
    const size_t c0 = begin > r0 ? begin - r0 : 0, c1 = std::min<size_t>(std::min(end - r0, lib->stride), lib->win[row].kept);
    if (c0 < c1) f(row, c0, c1);
  }
}

// The streams this rank's share of the fingerprinting is made of: whole windows without a communicator, otherwise per row
// of its block the sub-window of PCM the block's columns depend on, hashes straight to their places.
Status block_spans(const NeedleHipLibrary *lib, int world, int rank, std::vector<StreamSpan> *spans) {
  if (world <= 1) {
    for (size_t row = 0; row < lib->rows(); row++) {
      const Window &w = lib->win[row];
      if (w.pcm_off == ~0ull)
        return Status::Make(NeedleError_InvalidArgument, "video " + std::to_string(row / lib->regions()) + " has no PCM on this rank");
      spans->push_back(StreamSpan{w.pcm_off, w.values, (uint64_t)row * lib->stride});
    }
    return Status::Ok();
  }
  Status bad;
  for_rows_of_block(lib, world, rank, [&](size_t row, size_t c0, size_t c1) {
    const Window &w = lib->win[row];
    if (w.pcm_off == ~0ull) {
      bad = Status::Make(NeedleError_InvalidArgument, "video " + std::to_string(row / lib->regions()) +
                                                          " has no PCM on this rank (needle_hip_library_rank_videos names the videos a rank needs)");
      return;
    }
    // hashes c0 .. c1-1 = raw items c0 step .. (c1 - 1) step = frames c0 step .. (c1 - 1) step + 19
    const uint64_t x0 = (uint64_t)c0 * lib->step, frames = (uint64_t)(c1 - 1 - c0) * lib->step + 20;
    const uint64_t samples = (frames - 1) * (uint64_t)kHop + (uint64_t)kFrameSize;
    spans->push_back(StreamSpan{w.pcm_off + x0 * (uint64_t)kHop * (uint64_t)lib->channels, samples * (uint64_t)lib->channels,
                                (uint64_t)row * lib->stride + c0});
  });
  return bad;
}

// Fingerprints this rank's block.
NeedleError analyze_flat_block(NeedleHipLibrary *lib, int world, int rank, int slot, uint32_t *zero_word) {
  std::vector<StreamSpan> spans;
  bool zeroed = false;
  lib->count_zeroed = nullptr;
  Status s = block_spans(lib, world, rank, &spans);
  if (!s.ok()) return report(s);
  if (spans.empty()) return NeedleError_Ok;
  // `slot`: the job slot is the pipeline depth -- job k + 1's STFT may overlap the tail of job k (common.h)
  s = gpu_fingerprint_device(lib->d_pcm.ptr, spans, lib->channels, lib->step, lib->arena, false, nullptr, nullptr, 0, slot,
                             zero_word, &zeroed);
  if (s.ok() && zeroed) lib->count_zeroed = zero_word;
  return s.ok() ? NeedleError_Ok : report(s);
}

uint32_t round_up4(uint64_t v) { return (uint32_t)std::min<uint64_t>((v + 3) & ~(uint64_t)3, 0xfffffffcu); }

Status job_buffers(NeedleHipLibrary *lib, NeedleHipLibrary::Job &j, int world) {
  if (!j.searched) NEEDLE_HIP_TRY(hipEventCreateWithFlags(&j.searched, hipEventDisableTiming));
  if (!j.done) NEEDLE_HIP_TRY(hipEventCreateWithFlags(&j.done, hipEventDisableTiming));
  if (j.slab_runs != lib->slab_runs || j.world != world) {
    j.slab_runs = lib->slab_runs;
    j.world = world;
    j.d_slabs.release();
    Status s = j.d_slabs.reserve(j.slab_bytes());  // this rank's slab only: the other ranks' runs arrive as heads
    if (!s.ok()) return s;
  }
  // the one-trip download: everything the last job needed plus a margin, the whole slab while that is small
  uint32_t head = lib->last_max_count ? round_up4((uint64_t)lib->last_max_count + lib->last_max_count / 8 + 64) : 4096u;
  const char *forced = lib->last_max_count ? nullptr : getenv("NEEDLE_HIP_HEAD_RUNS");  // tests: a first head that overflows
  if (forced) head = round_up4((uint64_t)std::max(4, atoi(forced)));
  head = std::min(head, j.slab_runs);
  if (!forced && j.slab_bytes() * (size_t)world <= (1u << 20)) head = j.slab_runs;
  j.head_runs = head;
  const size_t want = j.head_bytes() * (size_t)world;
  if (world > 1) {
    Status s = j.d_heads.reserve(want);
    if (!s.ok()) return s;
  }
  if (want > j.host_bytes) {
    if (j.host) (void)hipHostFree(j.host);
    j.host = nullptr;
    j.host_bytes = 0;
    NEEDLE_HIP_TRY(hipHostMalloc(&j.host, want + want / 4, hipHostMallocDefault));
    j.host_bytes = want + want / 4;
  }
  return Status::Ok();
}

// The per-video epilogue on the device (epilogue.hip) instead of on host threads: at library scale, where the host form
// takes tens of milliseconds and every rank of a node has only its share of the CPUs.  Decided when the job is enqueued
// (the run count is not known yet): by the number of sequence pairs, or NEEDLE_HIP_DEVICE_EPILOGUE=1 / 0.
bool device_epilogue_wanted(const NeedleHipLibrary *lib, size_t comparator_regions) {
  if (const char *e = getenv("NEEDLE_HIP_DEVICE_EPILOGUE")) return atoi(e) != 0;
  // ... or by what the library's last finished job found (round 6): content decides the run count -- 28 episodes with
  // stretches of silence give their 378 pairs 41 528 runs instead of 852, and the host form 2.5 ms of a 0.5 ms job.
  // (Every rank sees every rank's count: the same decision everywhere.)
  return (uint64_t)pair_count(lib->n) * comparator_regions >= kDeviceEpiloguePairs || lib->last_max_count >= kDeviceEpilogueRuns;
}

// Owner-directed job: the runs this rank received (its own videos' pairs), downloaded once, on demand.
Status fetch_directed(NeedleHipLibrary::Job &j, int world, int rank) {
  if (j.fetched_valid) return Status::Ok();
  const size_t W = (size_t)world, cw = NeedleHipLibrary::Job::count_words(world);
  const uint32_t *m = static_cast<const uint32_t *>(j.host_counts);
  size_t total = 0, off = 0;
  for (size_t r = 0; r < W; r++) total += std::min(m[r * cw + 1 + (size_t)rank], j.cap[r * W + (size_t)rank]);
  j.fetched.resize(total);
  size_t at = 0;
  for (size_t r = 0; r < W; r++) {
    const uint32_t cap = j.cap[r * W + (size_t)rank], have = std::min(m[r * cw + 1 + (size_t)rank], cap);
    if (have)
      NEEDLE_HIP_TRY(hipMemcpyAsync(j.fetched.data() + at, j.d_dir_recv.ptr + off + NeedleHipLibrary::kSlabHeader,
                                    (size_t)have * sizeof(NeedleHipRun), hipMemcpyDeviceToHost, download_stream()));
    at += have;
    off += NeedleHipLibrary::kSlabHeader + (size_t)cap * sizeof(NeedleHipRun);
  }
  NEEDLE_HIP_TRY(hipStreamSynchronize(download_stream()));
  j.fetched_valid = true;
  return Status::Ok();
}

// scan of this rank's pair range into its slab, gather of the slabs, download of their heads: all asynchronous
NeedleError job_search_and_gather(NeedleHipLibrary *lib, const NeedleAudioComparator *comparator, NeedleHipLibrary::Job &j) {
  const int world = comm_world(), rank = comm_rank();
  size_t pfirst = 0, pcount = 0;
  shard_range(pair_count(lib->n), world, rank, &pfirst, &pcount);
  uint8_t *mine = j.d_slabs.ptr;
  NeedleError e = needle_hip_library_search(lib, comparator, pfirst, pcount,
                                            reinterpret_cast<NeedleHipRun *>(mine + NeedleHipLibrary::kSlabHeader), j.slab_runs,
                                            reinterpret_cast<uint32_t *>(mine), false);
  if (e != NeedleError_Ok) return e;
  hipStream_t stream = library_stream(), down = download_stream();
  if (hipEventRecord(j.searched, stream) != hipSuccess || hipStreamWaitEvent(down, j.searched, 0) != hipSuccess)
    return report(Status::Make(NeedleError_Unknown, "stream ordering failed"));
  // The epilogue's form first (the exchange depends on it): on the device or on host threads, every rank for all videos
  // or -- more than one rank and a library large enough to pay for one more collective -- each for its own block (the
  // decision cannot wait for the run count: the pairs stand in for it).
  const Comparator &cmp = comparator_of(comparator);
  const size_t Rc = cmp.include_endings() ? 2 : 1;
  j.device_epilogue = device_epilogue_wanted(lib, Rc) && lib->n >= 2;
  j.shape_fixed = j.device_epilogue;
  if (j.device_epilogue) {
    uint64_t shard_from = 1u << 16;
    if (const char *e = getenv("NEEDLE_HIP_SHARD_EPILOGUE_PAIRS")) shard_from = (uint64_t)std::max(1, atoi(e));  // tests
    j.sharded = world > 1 && (getenv("NEEDLE_HIP_SHARD_EPILOGUE") ? atoi(getenv("NEEDLE_HIP_SHARD_EPILOGUE")) != 0
                                                                   : (uint64_t)pair_count(lib->n) * Rc >= shard_from);
  }
  // Owner-directed exchange (Job): with the sharded device epilogue a rank needs the runs of its own videos' pairs only.
  j.counted = j.directed = false;
  j.cap.clear();
  j.fetched_valid = false;
  const char *dir_env = getenv("NEEDLE_HIP_DIRECTED_RUNS");  // 0: every job travels as heads (measurements, tests)
  const bool want_directed = world > 1 && world <= 64 && j.device_epilogue && j.sharded && !lib->dir_unsupported &&
                             !(dir_env && atoi(dir_env) == 0);
  std::vector<size_t> recv_off((size_t)world, 0), recv_bytes((size_t)world, 0);
  if (want_directed) {
    const size_t W = (size_t)world;
    const bool have = lib->dir_world == world && lib->dir_counts.size() == W * W;
    std::vector<uint32_t> cap(W * W, 0u);
    if (have)
      for (size_t k = 0; k < W * W; k++) cap[k] = round_up4((uint64_t)lib->dir_counts[k] + lib->dir_counts[k] / 8 + 64);
    if (const char *e = getenv("NEEDLE_HIP_TEST_DIRECTED_CAP"))  // tests: the library's first directed job gets blocks that overflow
      if (have && !lib->dir_cap_forced) {
        std::fill(cap.begin(), cap.end(), round_up4((uint64_t)std::max(4, atoi(e))));
        lib->dir_cap_forced = true;
      }
    DirectPlan plan;
    std::memset(&plan, 0, sizeof(plan));
    std::vector<size_t> send_off(W), send_bytes(W);
    size_t off = 0, largest = 0;
    for (size_t q = 0; q < W; q++) {
      plan.offset[q] = (uint32_t)off;
      plan.capacity[q] = cap[(size_t)rank * W + q];
      send_off[q] = off;
      send_bytes[q] = NeedleHipLibrary::kSlabHeader + (size_t)plan.capacity[q] * sizeof(NeedleHipRun);
      off += send_bytes[q];
    }
    for (size_t k = 0; k < W * W; k++) largest = std::max(largest, NeedleHipLibrary::kSlabHeader + (size_t)cap[k] * sizeof(NeedleHipRun));
    if (off >= 0xFFFFFFF0ull) return report(Status::Make(NeedleError_InvalidArgument, "directed run exchange: blocks beyond 4 GiB"));
    const size_t cw = NeedleHipLibrary::Job::count_words(world);
    Status s = j.d_dir_send.reserve(off);
    if (s.ok()) s = j.d_dir_counts.reserve(cw * W);
    if (s.ok() && cw * W * sizeof(uint32_t) > j.host_counts_bytes) {
      if (j.host_counts) (void)hipHostFree(j.host_counts);
      j.host_counts = nullptr;
      j.host_counts_bytes = 0;
      if (hipHostMalloc(&j.host_counts, cw * W * sizeof(uint32_t), hipHostMallocDefault) != hipSuccess)
        s = Status::Make(NeedleError_Unknown, "pinned allocation failed");
      else
        j.host_counts_bytes = cw * W * sizeof(uint32_t);
    }
    if (s.ok())
      s = gpu_direct_runs(reinterpret_cast<const uint32_t *>(mine), reinterpret_cast<const NeedleHipRun *>(mine + NeedleHipLibrary::kSlabHeader),
                          j.slab_runs, (uint32_t)lib->n, (uint32_t)Rc, (uint32_t)shard_block(lib->n, world), world, j.d_dir_send.ptr, plan, down);
    if (!s.ok()) return report(s);
    // this rank's row of the count matrix: [runs in its slab, runs directed to rank 0, 1, ...]; every rank gets every row
    uint32_t *row = j.d_dir_counts.ptr + (size_t)rank * cw;
    bool ok = hipMemsetAsync(row, 0, cw * sizeof(uint32_t), down) == hipSuccess &&
              hipMemcpyAsync(row, mine, sizeof(uint32_t), hipMemcpyDeviceToDevice, down) == hipSuccess;
    for (size_t q = 0; q < W && ok; q++)
      ok = hipMemcpyAsync(row + 1 + q, j.d_dir_send.ptr + send_off[q], sizeof(uint32_t), hipMemcpyDeviceToDevice, down) == hipSuccess;
    if (!ok) return report(Status::Make(NeedleError_Unknown, "directed run exchange: count row failed"));
    s = comm_all_gather(kSide, row, j.d_dir_counts.ptr, cw * sizeof(uint32_t), down);
    if (!s.ok()) return report(s);
    if (hipMemcpyAsync(j.host_counts, j.d_dir_counts.ptr, cw * W * sizeof(uint32_t), hipMemcpyDeviceToHost, down) != hipSuccess)
      return report(Status::Make(NeedleError_Unknown, "directed run exchange: count download failed"));
    j.counted = true;
    j.comm_bytes[1] += cw * W * sizeof(uint32_t);
    if (have) {
      size_t roff = 0;
      for (size_t r = 0; r < W; r++) {
        recv_off[r] = roff;
        recv_bytes[r] = NeedleHipLibrary::kSlabHeader + (size_t)cap[r * W + (size_t)rank] * sizeof(NeedleHipRun);
        roff += recv_bytes[r];
      }
      if (!(s = j.d_dir_recv.reserve(roff)).ok()) return report(s);
      s = comm_all_to_all_v(kSide, j.d_dir_send.ptr, send_off.data(), send_bytes.data(), j.d_dir_recv.ptr, recv_off.data(), recv_bytes.data(),
                            largest, down);
      if (s.ok()) {
        j.directed = true;
        j.cap = cap;
        j.comm_bytes[1] += roff;
      } else if (s.message.find("unsupported") != std::string::npos) {
        lib->dir_unsupported = true;  // (every rank loads the same librccl: every rank lands here, before any transfer)
        (void)hipGetLastError();
      } else {
        return report(s);
      }
    }
  }
  // What travels otherwise is the HEAD of every rank's slab -- its count and as many runs as the last job needed plus a margin
  // (job_buffers) -- not the slab's capacity: at BASELINE.json configs[4] on 8 GPUs a slab is 24 MB per rank and the
  // runs found 13 MB.  The heads land side by side in d_heads and come down in ONE copy; a rank whose count exceeds the
  // head is handled like a slab overflow in job_end (every rank sees every count): the head grows, the job's scan
  // and gather are repeated, the size sticks.  One rank: the head is copied straight out of the slab.
  const uint8_t *src = mine;
  if (!j.directed) {
    if (world > 1) {
      Status s = comm_all_gather(kSide, mine, j.d_heads.ptr, j.head_bytes(), down);
      if (!s.ok()) return report(s);
      src = j.d_heads.ptr;
      j.comm_bytes[1] += j.head_bytes() * (uint64_t)world;
    }
    if (hipMemcpyAsync(j.host, src, j.head_bytes() * (size_t)world, hipMemcpyDeviceToHost, down) != hipSuccess)
      return report(Status::Make(NeedleError_Unknown, "asynchronous run download failed"));
  }
  if (j.device_epilogue) {
    const size_t want = (lib->n + 1) * sizeof(NeedleHipSearchResult);
    if (want > j.host_results_bytes) {
      if (j.host_results) (void)hipHostFree(j.host_results);
      j.host_results = nullptr;
      j.host_results_bytes = 0;
      if (hipHostMalloc(&j.host_results, want, hipHostMallocDefault) != hipSuccess)
        return report(Status::Make(NeedleError_Unknown, "pinned allocation failed"));
      j.host_results_bytes = want;
    }
    lib->ensure_row_tables();
    EpilogueJob ej;
    ej.slot = (int)(&j - lib->job);
    ej.n = (uint32_t)lib->n;
    ej.regions = (uint32_t)Rc;
    ej.rows_per_video = (uint32_t)lib->regions();
    size_t v0 = 0, vcount = lib->n;
    if (j.sharded) shard_range(lib->n, world, rank, &v0, &vcount);
    ej.v0 = (uint32_t)v0;
    ej.v1 = (uint32_t)(v0 + vcount);
    ej.threshold = cmp.hash_match_threshold();
    ej.include_endings = cmp.include_endings();
    ej.min_opening_duration = cmp.min_opening_duration();
    ej.min_ending_duration = cmp.min_ending_duration();
    ej.time_padding = cmp.time_padding();
    ej.hash_duration = lib->hash_duration;
    ej.num_segments = world;
    auto segment = [&](int k, const uint8_t *slab) {  // a slab: 32-byte header (word 0 = runs found), then the runs
      ej.segment_count[k] = reinterpret_cast<const uint32_t *>(slab);
      ej.segment_runs[k] = reinterpret_cast<const NeedleHipRun *>(slab + NeedleHipLibrary::kSlabHeader);
    };
    if (j.directed) {  // the blocks received: one segment per source rank, each of its own capacity
      ej.max_runs = 0;
      for (int r = 0; r < world; r++) {
        segment(r, j.d_dir_recv.ptr + recv_off[(size_t)r]);
        ej.segment_capacities[r] = j.cap[(size_t)r * (size_t)world + (size_t)rank];
        ej.max_runs += ej.segment_capacities[r];
      }
      ej.segment_capacity = 0;
    } else if (world > 1) {
      for (int r = 0; r < world && r < 64; r++) segment(r, j.d_heads.ptr + (size_t)r * j.head_bytes());
      ej.segment_capacity = j.head_runs;
      ej.max_runs = (uint64_t)ej.segment_capacity * (uint64_t)world;
    } else {
      segment(0, mine);
      ej.segment_capacity = j.slab_runs;
      ej.max_runs = (uint64_t)ej.segment_capacity * (uint64_t)world;
    }
    ej.row_len = &lib->row_len;
    ej.row_ts = &lib->row_ts;
    ej.row_seek = &lib->row_seek;
    ej.ts = &lib->ts_tables;
    NeedleHipSearchResult *hr = static_cast<NeedleHipSearchResult *>(j.host_results);
    Status s = world <= 64 ? gpu_epilogue_enqueue(ej, down, hr, reinterpret_cast<uint32_t *>(hr + lib->n))
                           : Status::Make(NeedleError_InvalidArgument, "device epilogue: more than 64 ranks");
    if (const char *e = getenv("NEEDLE_HIP_TEST_EPILOGUE_FAIL_RANK"))  // tests: the device form "fails" on one rank alone
      if (atoi(e) == rank) s = Status::Make(NeedleError_Unknown, "device epilogue: failure injected by the test");
    if (!s.ok()) {  // (workspaces that do not fit: the host form computes the same results from the downloaded run list --
      j.device_epilogue = false;  // for the same block of videos, j.sharded stands: a rank that fell back alone must still
      (void)hipGetLastError();    // meet the others in the collective they enter, or not enter one they skip)
    }
  }
  if (hipEventRecord(j.done, down) != hipSuccess)
    return report(Status::Make(NeedleError_Unknown, "asynchronous run download failed"));
  return NeedleError_Ok;
}

bool shard_epilogue(size_t total_runs) {
  // Sharding the per-video epilogue costs one more (small) collective; it pays once the epilogue is milliseconds.
  if (const char *e = getenv("NEEDLE_HIP_SHARD_EPILOGUE")) return atoi(e) != 0;
  return total_runs >= (1u << 17);
}

}  // namespace

extern "C" {

int needle_hip_comm_rank(void) { return comm_rank(); }
int needle_hip_comm_world_size(void) { return comm_world(); }
const char *needle_hip_comm_backend(void) { return comm_backend(); }

enum NeedleError needle_hip_comm_create_id(uint8_t id[NEEDLE_HIP_COMM_ID_BYTES]) {
  if (!id) return NeedleError_NullArgument;
  return guarded([&]() -> NeedleError {
    Status s = comm_create_id(id);
    return s.ok() ? NeedleError_Ok : report(s);
  });
}

enum NeedleError needle_hip_comm_init(const uint8_t id[NEEDLE_HIP_COMM_ID_BYTES], int rank, int world_size) {
  if (!id) return NeedleError_NullArgument;
  return guarded([&]() -> NeedleError {
    Status s = comm_init(id, rank, world_size);
    return s.ok() ? NeedleError_Ok : report(s);
  });
}

void needle_hip_comm_finalize(void) { comm_finalize(); }

enum NeedleError needle_hip_comm_barrier(void) {
  return guarded([&]() -> NeedleError {
    Status s = comm_barrier();
    return s.ok() ? NeedleError_Ok : report(s);
  });
}

enum NeedleError needle_hip_comm_all_gather_host(const void *send, void *recv, size_t bytes_per_rank) {
  if ((!send || !recv) && bytes_per_rank) return NeedleError_NullArgument;
  return guarded([&]() -> NeedleError {
    Status s = comm_all_gather_host(send, recv, bytes_per_rank);
    return s.ok() ? NeedleError_Ok : report(s);
  });
}

void needle_hip_comm_shard(size_t units, int world_size, int rank, size_t *first, size_t *count) {
  size_t f = 0, c = 0;
  if (world_size >= 1 && rank >= 0 && rank < world_size) shard_range(units, world_size, rank, &f, &c);
  if (first) *first = f;
  if (count) *count = c;
}

enum NeedleError needle_hip_library_rank_videos(const NeedleHipLibrary *lib, const size_t *num_values, int channels, int world_size,
                                                int rank, size_t *first_video, size_t *video_count) {
  if (!lib || !num_values || !first_video || !video_count) return NeedleError_NullArgument;
  if ((channels != 1 && channels != 2) || world_size < 1 || rank < 0 || rank >= world_size) return NeedleError_InvalidArgument;
  return guarded([&]() -> NeedleError {
    // the geometry plan_windows will arrive at for these lengths, without touching the library
    NeedleHipLibrary plan;
    plan.n = lib->n;
    plan.endings = lib->endings;
    plan.step = lib->step;
    const size_t R = plan.regions();
    plan.win.assign(plan.rows(), Window{});
    uint32_t max_kept = 0;
    for (size_t v = 0; v < plan.n; v++) {
      size_t open_samples = 0, end_first = 0;
      ns_t seek = 0;
      const size_t samples = num_values[v] / (size_t)channels;
      Status s = Analyzer::windows(samples, kSampleRate, lib->opening_pct, lib->ending_pct, &open_samples, &end_first, &seek);
      if (!s.ok()) return report(s);
      for (size_t r = 0; r < R; r++) {
        plan.win[v * R + r].kept = (uint32_t)num_kept(r == 0 ? open_samples : samples - end_first, lib->step);
        max_kept = std::max(max_kept, plan.win[v * R + r].kept);
      }
    }
    size_t tiles_per_row = std::max<size_t>(1, ((size_t)max_kept + 63) / 64);
    while (world_size > 1 && (plan.rows() * tiles_per_row) % (size_t)world_size) tiles_per_row++;
    plan.stride = 64 * tiles_per_row;
    size_t lo = plan.n, hi = 0;
    if (world_size == 1) {
      lo = 0;
      hi = plan.n;
    } else {
      for_rows_of_block(&plan, world_size, rank, [&](size_t row, size_t, size_t) {
        lo = std::min(lo, row / R);
        hi = std::max(hi, row / R + 1);
      });
    }
    *first_video = lo < hi ? lo : 0;
    *video_count = lo < hi ? hi - lo : 0;
    return NeedleError_Ok;
  });
}

enum NeedleError needle_hip_library_job_begin(NeedleHipLibrary *lib, const struct NeedleAudioComparator *comparator, int slot) {
  if (!lib || !comparator) return NeedleError_NullArgument;
  if (slot < 0 || slot > 1 || !lib->have_pcm) return NeedleError_InvalidArgument;
  return guarded([&]() -> NeedleError {
    NeedleHipLibrary::Job &j = lib->job[slot];
    if (j.pending) return report(Status::Make(NeedleError_InvalidArgument, "job slot still pending"));
    const int world = comm_world(), rank = comm_rank();
    j.last_runs = nullptr;
    j.last_total = 0;
    std::fill(j.comm_bytes, j.comm_bytes + 4, 0);
    if (!lib->flat_shardable(world))
      return report(Status::Make(NeedleError_InvalidArgument,
                                 "the hash arena does not divide into this communicator's blocks: call needle_hip_library_set_pcm "
                                 "after needle_hip_comm_init (or adopt an arena of rows x stride with rows * stride / 64 a multiple of the world size)"));
    // (the run slabs first: the analyze step's last kernel clears this job's run counter)
    if (lib->slab_runs == 0 && getenv("NEEDLE_HIP_SLAB_RUNS"))  // tests: a slab that overflows
      lib->slab_runs = round_up4((uint64_t)std::max(4, atoi(getenv("NEEDLE_HIP_SLAB_RUNS"))));
    if (lib->slab_runs == 0)
      lib->slab_runs = round_up4(std::max<uint64_t>(1024, 4 * (uint64_t)shard_block(pair_count(lib->n), world) * lib->regions()));
    Status sb = job_buffers(lib, j, world);
    if (!sb.ok()) return report(sb);
    // 1. fingerprint this rank's block of HASHES (NeedleHipLibrary::flat_block) into the arena (analyzer.rs:437-445
    // across GPUs).  (After needle_hip_library_stream_pcm the rows are already there: the PCM was fingerprinted as it
    // was uploaded.)
    lib->count_zeroed = nullptr;
    NeedleError e = lib->pcm_resident ? analyze_flat_block(lib, world, rank, slot, reinterpret_cast<uint32_t *>(j.d_slabs.ptr))
                                      : NeedleError_Ok;
    if (e != NeedleError_Ok) return e;
    // 2. every rank gets every hash: one in-place all-gather of the equal blocks, in stream order
    if (comm_get()) {
      const size_t block_bytes = lib->flat_block(world) * sizeof(uint32_t);
      Status s = comm_all_gather(kData, reinterpret_cast<const uint8_t *>(lib->arena) + (size_t)rank * block_bytes, lib->arena,
                                 block_bytes, library_stream());
      if (!s.ok()) return report(s);
      j.comm_bytes[0] += block_bytes * (uint64_t)world;
    }
    // 3. + 4. scan this rank's pair range (comparator.rs:549-564 across GPUs), gather the run lists
    if ((e = job_search_and_gather(lib, comparator, j)) != NeedleError_Ok) return e;
    j.pending = true;
    return NeedleError_Ok;
  });
}

enum NeedleError needle_hip_library_job_end(NeedleHipLibrary *lib, const struct NeedleAudioComparator *comparator, int slot,
                                            NeedleHipSearchResult *results, size_t *num_runs) {
  if (!lib || !comparator || !results) return NeedleError_NullArgument;
  if (slot < 0 || slot > 1 || !lib->job[slot].pending) return NeedleError_InvalidArgument;
  return guarded([&]() -> NeedleError {
    NeedleHipLibrary::Job &j = lib->job[slot];
    const int world = comm_world(), rank = comm_rank();
    const Comparator &cmp = comparator_of(comparator);
    std::vector<uint32_t> counts((size_t)world);
    const size_t W = (size_t)world, cw = NeedleHipLibrary::Job::count_words(world);
    auto take_matrix = [&]() {  // the job's count matrix becomes the next job's block sizes (every rank holds the same one)
      const uint32_t *m = static_cast<const uint32_t *>(j.host_counts);
      lib->dir_counts.assign(W * W, 0u);
      for (size_t r = 0; r < W; r++)
        for (size_t q = 0; q < W; q++) lib->dir_counts[r * W + q] = m[r * cw + 1 + q];
      lib->dir_world = world;
    };
    for (;;) {
      if (hipEventSynchronize(j.done) != hipSuccess) return report(Status::Make(NeedleError_Unknown, "run download failed"));
      uint32_t most = 0;
      if (j.directed) {  // nothing but the counts came down
        const uint32_t *m = static_cast<const uint32_t *>(j.host_counts);
        for (size_t r = 0; r < W; r++) {
          counts[r] = m[r * cw];
          most = std::max(most, counts[r]);
        }
        if (most <= j.slab_runs) {
          bool fits = true;
          for (size_t r = 0; r < W && fits; r++)
            for (size_t q = 0; q < W && fits; q++) fits = m[r * cw + 1 + q] <= j.cap[r * W + q];
          take_matrix();
          lib->last_max_count = most;
          if (fits) break;
          // a block overflowed (every rank sees it): the sizes follow the matrix just taken, scan and exchange are repeated
          NeedleError e = job_search_and_gather(lib, comparator, j);
          if (e != NeedleError_Ok) return e;
          j.comm_bytes[3]++;
          continue;
        }
      } else {
        for (int r = 0; r < world; r++) {
          counts[r] = *reinterpret_cast<const uint32_t *>(static_cast<const char *>(j.host) + (size_t)r * j.head_bytes());
          most = std::max(most, counts[r]);
        }
        if (most <= j.slab_runs && (world == 1 || most <= j.head_runs)) {
          lib->last_max_count = most;
          if (j.counted) take_matrix();  // (a job that travelled as heads and counted: the next one can be directed)
          break;
        }
        if (most <= j.slab_runs) {  // world > 1 and some rank's list is longer than the head that was gathered
          lib->last_max_count = most;
          Status s = job_buffers(lib, j, world);  // head_runs from last_max_count
          if (!s.ok()) return report(s);
          NeedleError e = job_search_and_gather(lib, comparator, j);
          if (e != NeedleError_Ok) return e;
          j.comm_bytes[3]++;
          continue;
        }
      }
      // Some rank found more runs than a slab holds (every rank sees the same counts, so every rank takes this
      // branch): grow and repeat the scan of this job -- the scan is deterministic and the arena still holds the
      // hashes, whether or not the next job's analyze has run in between (same PCM, same rows).
      lib->slab_runs = round_up4((uint64_t)most + most / 4 + 64);
      lib->last_max_count = most;
      lib->dir_counts.clear();  // (counted over a clamped slab)
      Status s = job_buffers(lib, j, world);
      if (!s.ok()) return report(s);
      NeedleError e = job_search_and_gather(lib, comparator, j);
      if (e != NeedleError_Ok) return e;
      j.comm_bytes[3]++;
    }
    j.pending = false;
    // run list of all ranks: heads from the pinned buffer, tails (rare: the first job of a library) straight from HBM
    size_t total = 0;
    for (int r = 0; r < world; r++) total += counts[r];
    const NeedleHipRun *run_list = nullptr;
    if (j.directed)  // the runs stayed on the devices: each rank holds its own videos' (fetch_directed, on demand)
      j.merged.clear();
    else if (world == 1 && counts[0] <= j.head_runs)  // one rank, everything in the pinned head: no copy of ~100 MB at library scale
      run_list = reinterpret_cast<const NeedleHipRun *>(static_cast<const char *>(j.host) + NeedleHipLibrary::kSlabHeader);
    else
      j.merged.resize(total);
    size_t at = 0;
    bool tails = false;
    for (int r = 0; r < world && !run_list && !j.directed; r++) {  // (tails beyond the head: one rank only, out of its own slab)
      const uint32_t head = std::min(counts[r], j.head_runs);
      std::memcpy(j.merged.data() + at, static_cast<const char *>(j.host) + (size_t)r * j.head_bytes() + NeedleHipLibrary::kSlabHeader,
                  (size_t)head * sizeof(NeedleHipRun));
      if (counts[r] > head) {
        tails = true;
        if (hipMemcpyAsync(j.merged.data() + at + head,
                           j.d_slabs.ptr + (size_t)r * j.slab_bytes() + NeedleHipLibrary::kSlabHeader + (size_t)head * sizeof(NeedleHipRun),
                           (size_t)(counts[r] - head) * sizeof(NeedleHipRun), hipMemcpyDeviceToHost, download_stream()) != hipSuccess)
          return report(Status::Make(NeedleError_Unknown, "run download failed"));
      }
      at += counts[r];
    }
    if (tails && hipStreamSynchronize(download_stream()) != hipSuccess)
      return report(Status::Make(NeedleError_Unknown, "run download failed"));
    if (num_runs) *num_runs = total;

    // 5. the order-sensitive per-video epilogue (comparator.rs:583-626): every rank for all videos while that is
    // cheaper than another collective, otherwise each rank for its own block of videos + one all-gather of results
    std::vector<VideoResult> res;
    const bool sharded = j.shape_fixed ? j.sharded : (world > 1 && shard_epilogue(total));
    size_t v0 = 0, vcount = lib->n;
    if (sharded) shard_range(lib->n, world, rank, &v0, &vcount);
    const auto t0 = std::chrono::steady_clock::now();
    if (!run_list && !j.directed) run_list = j.merged.data();
    j.last_runs = run_list;
    j.last_total = total;
    Status s;
    const NeedleHipSearchResult *device_results = nullptr;
    if (j.device_epilogue) {  // computed on the device behind the gather (epilogue.hip); j.done covers its copies
      device_results = static_cast<const NeedleHipSearchResult *>(j.host_results);
      const uint32_t fail = *reinterpret_cast<const uint32_t *>(device_results + lib->n);
      if (fail & kEpilogueBucketTooLarge) {
        device_results = nullptr;  // a pair with more runs than one lane should order: the host form below, same block of videos
        note_epilogue_host_fallback("library job", total, vcount);
      }
      else if (fail != 0)
        s = Status::Make(NeedleError_Unknown, "overflow when subtracting durations (time_padding / hash_duration exceed the match end)");
    }
    j.ran_device_epilogue = device_results != nullptr;
    j.ran_sharded = sharded;
    {
      int32_t form = 0;
      uint64_t products = 0;
      gpu_scan_last_launch(&form, &products);  // (of this process's last scan launch: this job's, or the next one's -- the same shape)
      j.scan_form = (uint32_t)form;
    }
    if (!device_results) {
      const std::vector<const FrameHashesData *> fh = lib->shell_pointers();
      if (j.directed) {  // this rank's videos' runs are what it received: down they come, once
        Status f = fetch_directed(j, world, rank);
        if (!f.ok()) return report(f);
        s = cmp.results_from_runs(fh, j.fetched.data(), j.fetched.size(), false, false, false, &res, v0, v0 + vcount);
      } else {
        s = cmp.results_from_runs(fh, run_list, total, false, false, false, &res, v0, v0 + vcount);
      }
    }
    if (getenv("NEEDLE_HIP_TRACE"))
      std::fprintf(stderr, "[needle_hip] rank %d epilogue %zu runs, videos [%zu, %zu): %.1f us\n", rank, total, v0, v0 + vcount,
                   std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
    // an error on one rank must not leave the others waiting in the collective below: it travels with the results
    if (!sharded) {
      if (!s.ok()) return report(s);
      if (device_results)
        std::memcpy(results, device_results, lib->n * sizeof(NeedleHipSearchResult));
      else
        for (size_t v = 0; v < lib->n; v++) fill_c_result(res[v], &results[v]);
      return NeedleError_Ok;
    }
    const size_t b = shard_block(lib->n, world);
    struct Block {
      uint64_t failed;
    };
    const size_t block_bytes = sizeof(Block) + b * sizeof(NeedleHipSearchResult);
    std::vector<uint8_t> mine(block_bytes, 0), all(block_bytes * (size_t)world, 0);
    reinterpret_cast<Block *>(mine.data())->failed = s.ok() ? 0 : 1;
    if (s.ok() && device_results)
      std::memcpy(mine.data() + sizeof(Block), device_results + v0, vcount * sizeof(NeedleHipSearchResult));
    else if (s.ok())
      for (size_t k = 0; k < vcount; k++)
        fill_c_result(res[v0 + k], reinterpret_cast<NeedleHipSearchResult *>(mine.data() + sizeof(Block)) + k);
    Status g = comm_all_gather_host(mine.data(), all.data(), block_bytes);
    j.comm_bytes[2] += block_bytes * (uint64_t)world;
    if (!s.ok()) return report(s);
    if (!g.ok()) return report(g);
    for (int r = 0; r < world; r++) {
      const uint8_t *blk = all.data() + (size_t)r * block_bytes;
      if (reinterpret_cast<const Block *>(blk)->failed)
        return report(Status::Make(NeedleError_Unknown, "the epilogue failed on rank " + std::to_string(r)));
      size_t f = 0, c = 0;
      shard_range(lib->n, world, r, &f, &c);
      std::memcpy(results + f, blk + sizeof(Block), c * sizeof(NeedleHipSearchResult));
    }
    return NeedleError_Ok;
  });
}

enum NeedleError needle_hip_library_job_runs(const NeedleHipLibrary *lib, int slot, const NeedleHipRun **runs, size_t *num_runs) {
  if (!lib || !runs || !num_runs) return NeedleError_NullArgument;
  if (slot < 0 || slot > 1 || lib->job[slot].pending) return NeedleError_InvalidArgument;
  if (lib->job[slot].directed) {  // owner-directed exchange: a rank holds the runs of its own videos' pairs, nothing else
    NeedleHipLibrary::Job &j = const_cast<NeedleHipLibrary *>(lib)->job[slot];
    Status s = fetch_directed(j, comm_world(), comm_rank());
    if (!s.ok()) return report(s);
    *runs = j.fetched.data();
    *num_runs = j.fetched.size();
    return NeedleError_Ok;
  }
  if (!lib->job[slot].last_runs && lib->job[slot].last_total) return NeedleError_InvalidArgument;
  *runs = lib->job[slot].last_runs;
  *num_runs = lib->job[slot].last_total;
  return NeedleError_Ok;
}

enum NeedleError needle_hip_library_job_form(const NeedleHipLibrary *lib, int slot, uint32_t form[4]) {
  if (!lib || !form) return NeedleError_NullArgument;
  if (slot < 0 || slot > 1 || lib->job[slot].pending) return NeedleError_InvalidArgument;
  const NeedleHipLibrary::Job &j = lib->job[slot];
  form[0] = j.ran_device_epilogue ? 1u : 0u;
  form[1] = j.ran_sharded ? 1u : 0u;
  form[2] = j.directed ? 1u : 0u;
  form[3] = j.scan_form;
  return NeedleError_Ok;
}

enum NeedleError needle_hip_library_job_comm_bytes(const NeedleHipLibrary *lib, int slot, uint64_t bytes[4]) {
  if (!lib || !bytes) return NeedleError_NullArgument;
  if (slot < 0 || slot > 1) return NeedleError_InvalidArgument;
  std::copy(lib->job[slot].comm_bytes, lib->job[slot].comm_bytes + 4, bytes);
  return NeedleError_Ok;
}

int needle_hip_host_threads(void) { return (int)host_threads(); }

enum NeedleError needle_hip_library_audit(NeedleHipLibrary *lib, NeedleHipCertAudit *audit) {
  if (!lib || !audit) return NeedleError_NullArgument;
  if (!lib->have_pcm || !lib->pcm_resident) return NeedleError_InvalidArgument;
  return guarded([&]() -> NeedleError {
    std::vector<StreamSpan> spans;
    Status s = block_spans(lib, comm_world(), comm_rank(), &spans);
    if (!s.ok()) return report(s);
    uint64_t c[4] = {0, 0, 0, 0};
    *audit = NeedleHipCertAudit{};
    if (spans.empty()) return NeedleError_Ok;
    s = gpu_fingerprint_audit_device(lib->d_pcm.ptr, spans, lib->channels, lib->step, lib->arena, c, &audit->max_error_over_s,
                                     &audit->max_s);
    if (!s.ok()) return report(s);
    audit->items = c[0];
    audit->accepted = c[1];
    audit->accepted_mismatches = c[2];
    audit->mismatches = c[3];
    return NeedleError_Ok;
  });
}

}  // extern "C"
