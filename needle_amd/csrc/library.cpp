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
#include <new>

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
  bool have_pcm = false;
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
  std::vector<FrameHashesData> shells;        // per-video timestamps for the epilogue (hashes stay in HBM)
  // double-buffered asynchronous run-list download (needle_hip_library_fetch_runs_begin / _end)
  struct Fetch {
    void *host = nullptr;  // pinned: u32 count, then max_runs NeedleHipRun
    uint32_t max_runs = 0;
    hipEvent_t ready = nullptr, done = nullptr;  // search enqueued (library stream) / copies finished (download stream)
    const NeedleHipRun *d_runs = nullptr;        // what the pending download reads
    bool pending = false;
  } fetch[2];
  ~NeedleHipLibrary() {
    for (Fetch &f : fetch) {
      if (f.host) (void)hipHostFree(f.host);
      if (f.done) (void)hipEventDestroy(f.done);
      if (f.ready) (void)hipEventDestroy(f.ready);
    }
  }

  size_t regions() const { return endings ? 2 : 1; }
  size_t rows() const { return n * regions(); }

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

enum NeedleError needle_hip_library_set_pcm(NeedleHipLibrary *lib, const int16_t *const *pcm, const size_t *num_values,
                                            int channels) {
  if (!lib || !pcm || !num_values) return NeedleError_NullArgument;
  if (channels != 1 && channels != 2) return NeedleError_InvalidArgument;
  return guarded([&]() -> NeedleError {
    Status s = ensure_device();
    if (!s.ok()) return report(s);
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
      if (!s.ok()) return report(s);
      for (size_t r = 0; r < R; r++) {
        Window &w = lib->win[v * R + r];
        const size_t count = r == 0 ? open_samples : samples - end_first;
        first_sample[v * R + r] = r == 0 ? 0 : end_first;
        w.values = count * (size_t)channels;
        w.kept = (uint32_t)num_kept(count, lib->step);
        w.seek = r == 0 ? 0 : seek;
        max_kept = std::max(max_kept, w.kept);
        if (pcm[v]) {
          w.pcm_off = total;
          total += (w.values + 1) & ~(uint64_t)1;
        }
      }
    }
    lib->stride = ((size_t)max_kept + 63) & ~(size_t)63;  // rows start 256-byte aligned
    if (lib->stride == 0) lib->stride = 64;
    if (!(s = lib->d_pcm.reserve(std::max<uint64_t>(total, 1))).ok()) return report(s);
    if (!(s = lib->d_arena.reserve(lib->rows() * lib->stride)).ok()) return report(s);
    hipStream_t stream = library_stream();
    lib->arena = lib->d_arena.ptr;
    if (hipMemsetAsync(lib->arena, 0, lib->rows() * lib->stride * sizeof(uint32_t), stream) != hipSuccess)
      return report(Status::Make(NeedleError_Unknown, "hipMemset failed"));
    std::vector<const int16_t *> src;
    std::vector<size_t> len;
    std::vector<uint64_t> dst;
    for (size_t v = 0; v < lib->n; v++) {
      for (size_t r = 0; r < R; r++) {
        const Window &w = lib->win[v * R + r];
        if (!pcm[v] || !w.values) continue;
        src.push_back(pcm[v] + first_sample[v * R + r] * (size_t)channels);
        len.push_back(w.values);
        dst.push_back(w.pcm_off);
      }
    }
    if (!(s = gpu_upload_pcm(src, len, dst, lib->d_pcm.ptr)).ok()) return report(s);
    if (hipStreamSynchronize(stream) != hipSuccess) return report(Status::Make(NeedleError_Unknown, "PCM upload failed"));
    lib->have_pcm = true;
    lib->min_len.clear();
    lib->shells.clear();
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
  lib->arena = d_arena;
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
      if (f.pending && f.d_runs == d_runs && hipStreamWaitEvent(library_stream(), f.done, 0) != hipSuccess)
        return report(Status::Make(NeedleError_Unknown, "stream wait failed"));
    const auto t_built = std::chrono::steady_clock::now();
    Status s = gpu_hamming_runs_device(lib->arena, seqs.data(), seqs.size(), problems.data(), problems.size(),
                                       cmp.hash_match_threshold(), d_runs, capacity, d_count, sync);
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
    // the epilogue needs timestamps only: the runs carry their simhashes, so the hash arena stays in HBM
    if (lib->shells.size() != lib->n) {
      lib->shells.assign(lib->n, {});
      const size_t R = lib->regions();
      for (size_t v = 0; v < lib->n; v++) {
        lib->shells[v].opening = lib->window_timestamps(lib->win[v * R]);
        if (lib->endings) lib->shells[v].ending = lib->window_timestamps(lib->win[v * R + 1]);
        lib->shells[v].hash_duration = lib->hash_duration;
      }
    }
    std::vector<const FrameHashesData *> fh;
    for (const FrameHashesData &d : lib->shells) fh.push_back(&d);
    std::vector<NeedleHipRun> run_vec(runs, runs + num_runs);
    std::vector<VideoResult> res;
    const auto t0 = std::chrono::steady_clock::now();
    Status s = cmp.results_from_runs(fh, run_vec, false, false, false, &res);
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
