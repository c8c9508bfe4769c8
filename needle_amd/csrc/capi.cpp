// extern "C" surface of libneedle_capi.so.
//   * the 13 needle-capi symbols (include/needle.h; reference: needle-capi/src/lib.rs)
//   * the additive needle_hip_* entry points (include/needle_hip.h)
// Nothing throws across this boundary: every body is wrapped, and failures become NeedleError codes
// with the detail printed as "needle error: ..." on stderr (lib.rs:124).
#include <dirent.h>
#include <hip/hip_runtime_api.h>
#include <sys/stat.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <fstream>
#include <new>

#include "epilogue.h"
#include "hipctx.h"
#include "needle_core.h"

using namespace needle;

// ---- opaque handle types (lib.rs:308,347-350,531) ----------------------------------------------------------
struct FrameHashes {
  FrameHashesData d;
};
struct NeedleAudioAnalyzer {
  Analyzer inner;
  std::vector<FrameHashes> frame_hashes;
};
struct NeedleAudioComparator {
  Comparator inner;
};

namespace {

template <typename F>
NeedleError guarded(F &&f) {
  try {
    return f();
  } catch (const std::bad_alloc &) {
    return report(Status::Make(NeedleError_Unknown, "out of memory"));
  } catch (const std::exception &e) {
    return report(Status::Make(NeedleError_Unknown, std::string("internal error: ") + e.what()));
  } catch (...) {
    return report(Status::Make(NeedleError_Unknown, "internal error"));
  }
}

bool valid_utf8(const char *s) {
  const unsigned char *p = reinterpret_cast<const unsigned char *>(s);
  while (*p) {
    int extra;
    unsigned cp;
    if (*p < 0x80) {
      p++;
      continue;
    } else if ((*p & 0xE0) == 0xC0) {
      extra = 1;
      cp = *p & 0x1F;
    } else if ((*p & 0xF0) == 0xE0) {
      extra = 2;
      cp = *p & 0x0F;
    } else if ((*p & 0xF8) == 0xF0) {
      extra = 3;
      cp = *p & 0x07;
    } else {
      return false;
    }
    p++;
    for (int i = 0; i < extra; i++, p++) {
      if ((*p & 0xC0) != 0x80) return false;
      cp = (cp << 6) | (*p & 0x3F);
    }
    if ((extra == 1 && cp < 0x80) || (extra == 2 && cp < 0x800) || (extra == 3 && cp < 0x10000) || cp > 0x10FFFF ||
        (cp >= 0xD800 && cp <= 0xDFFF))
      return false;
  }
  return true;
}

// get_paths_from_raw, lib.rs:283-304
NeedleError paths_from_raw(const char *const *raw, size_t n, std::vector<std::string> *out) {
  for (size_t i = 0; i < n; i++) {
    if (!raw[i]) return NeedleError_NullArgument;
    if (!valid_utf8(raw[i])) return NeedleError_InvalidUtf8String;
    out->emplace_back(raw[i]);
  }
  return NeedleError_Ok;
}

bool ends_with(const std::string &s, const std::string &suffix) {
  return s.size() >= suffix.size() && s.compare(s.size() - suffix.size(), suffix.size(), suffix) == 0;
}

// util.rs:22-53 with this build's notion of a decodable file: RIFF/WAVE.  (Upstream sniffs video
// containers with `infer` / FFmpeg; media decode is outside the analyze/search path built here.)
bool is_valid_media_file(const std::string &path, bool full) {
  if (ends_with(path, FRAME_HASH_DATA_FILE_NAME)) return false;
  struct stat st;
  if (stat(path.c_str(), &st) != 0 || !S_ISREG(st.st_mode)) return false;
  std::ifstream f(path, std::ios::binary);
  char head[12] = {0};
  f.read(head, 12);
  if (f.gcount() != 12 || std::memcmp(head, "RIFF", 4) != 0 || std::memcmp(head + 8, "WAVE", 4) != 0) return false;
  if (!full) return true;
  WavInfo w;  // `full`: the header has to describe an encoding this build decodes (util.rs:33-49 opens the demuxer)
  return wav_probe(path, &w).ok();
}

void fill_result(const VideoResult &v, NeedleHipSearchResult *r) {
  std::memset(r, 0, sizeof(*r));
  r->has_result = v.has_result;
  r->has_opening = v.result.has_opening;
  r->has_ending = v.result.has_ending;
  r->opening_start_ns = v.result.opening_start;
  r->opening_end_ns = v.result.opening_end;
  r->ending_start_ns = v.result.ending_start;
  r->ending_end_ns = v.result.ending_end;
}

}  // namespace

extern "C" {

// ============================================================================================================
// needle.h
// ============================================================================================================
const char *needle_error_to_str(enum NeedleError error) {  // lib.rs:138-199
  switch (error) {
    case NeedleError_Ok: return "No error";
    case NeedleError_InvalidUtf8String: return "Invalid UTF-8 string";
    case NeedleError_NullArgument: return "Input argument is NULL";
    case NeedleError_InvalidArgument: return "One or more input arguments were invalid (usually zero)";
    case NeedleError_FrameHashDataNotFound: return "Frame hash data not found on disk";
    case NeedleError_FrameHashDataInvalidVersion: return "Frame hash data has an invalid version.";
    case NeedleError_InvalidFrameHashData: return "Invalid frame hash data read from disk";
    case NeedleError_ComparatorMinimumPaths: return "Comparator requires at least 2 video paths";
    case NeedleError_AnalyzerInvalidHashPeriod: return "Analyzer hash period must be greater than 0";
    case NeedleError_AnalyzerInvalidHashDuration: return "Analyzer hash duration must be greater than 3 seconds";
    case NeedleError_IOError: return "I/O error";
    case NeedleError_Unknown: break;
  }
  return "Unknown error occurred; please re-run with logging enabled";
}

enum NeedleError needle_util_find_video_files(const char *const *paths, size_t num_paths, bool full, bool audio,
                                              const char *const **videos, size_t *num_videos) {
  (void)audio;  // every decodable file here is audio
  if (!paths || !videos || !num_videos) return NeedleError_NullArgument;  // lib.rs:216-218
  if (num_paths == 0) return NeedleError_InvalidArgument;                // lib.rs:219-221
  return guarded([&]() -> NeedleError {
    std::vector<std::string> in;
    NeedleError e = paths_from_raw(paths, num_paths, &in);
    if (e != NeedleError_Ok) return e;
    struct stat st;
    for (const std::string &p : in)  // util.rs:66-71 -> Error::PathNotFound -> Unknown (lib.rs:131)
      if (stat(p.c_str(), &st) != 0)
        return report(Status::Make(NeedleError_Unknown, "path does not exist: \"" + p + "\""));
    std::vector<std::string> found;
    for (const std::string &p : in) {
      stat(p.c_str(), &st);
      if (S_ISDIR(st.st_mode)) {  // one level deep, util.rs:77-88
        std::vector<std::string> names;
        if (DIR *d = opendir(p.c_str())) {
          while (dirent *ent = readdir(d)) {
            const std::string name = ent->d_name;
            if (name != "." && name != "..") names.push_back(name);
          }
          closedir(d);
        }
        std::sort(names.begin(), names.end());
        for (const std::string &name : names) {
          const std::string child = (ends_with(p, "/") ? p : p + "/") + name;
          if (is_valid_media_file(child, full)) found.push_back(child);
        }
      } else if (is_valid_media_file(p, full)) {
        found.push_back(p);
      }
    }
    char **arr = static_cast<char **>(std::malloc(std::max<size_t>(found.size(), 1) * sizeof(char *)));
    if (!arr) return report(Status::Make(NeedleError_Unknown, "out of memory"));
    for (size_t i = 0; i < found.size(); i++) arr[i] = strdup(found[i].c_str());
    *videos = const_cast<const char *const *>(arr);
    *num_videos = found.size();
    return NeedleError_Ok;
  });
}

void needle_util_video_files_free(const char *const *videos, size_t num_videos) {  // lib.rs:259-281
  if (!videos || num_videos == 0) return;
  char **arr = const_cast<char **>(videos);
  for (size_t i = 0; i < num_videos; i++) std::free(arr[i]);
  std::free(arr);
}

enum NeedleError needle_audio_analyzer_new(const char *const *paths, size_t num_paths,
                                           float opening_search_percentage, float ending_search_percentage,
                                           bool include_endings, bool threaded_decoding, bool force,
                                           struct NeedleAudioAnalyzer **output) {
  if (!paths || !output) return NeedleError_NullArgument;  // lib.rs:383-385
  return guarded([&]() -> NeedleError {
    std::vector<std::string> p;
    NeedleError e = paths_from_raw(paths, num_paths, &p);
    if (e != NeedleError_Ok) return e;
    auto *a = new NeedleAudioAnalyzer();
    a->inner = Analyzer::from_files(std::move(p), threaded_decoding, force);
    a->inner.with_opening_search_percentage(opening_search_percentage)
        .with_ending_search_percentage(ending_search_percentage)
        .with_include_endings(include_endings);
    *output = a;
    return NeedleError_Ok;
  });
}

enum NeedleError needle_audio_analyzer_new_default(const char *const *paths, size_t num_paths,
                                                   struct NeedleAudioAnalyzer **output) {  // lib.rs:354-369
  return needle_audio_analyzer_new(paths, num_paths, DEFAULT_OPENING_SEARCH_PERCENTAGE,
                                   DEFAULT_ENDING_SEARCH_PERCENTAGE, false, false, false, output);
}

enum NeedleError needle_audio_analyzer_get_frame_hashes(const struct NeedleAudioAnalyzer *analyzer, size_t index,
                                                        const struct FrameHashes **output) {  // lib.rs:415-435
  if (!analyzer || !output) return NeedleError_NullArgument;
  if (index >= analyzer->frame_hashes.size()) return NeedleError_InvalidArgument;
  *output = &analyzer->frame_hashes[index];
  return NeedleError_Ok;
}

void needle_audio_analyzer_free(const struct NeedleAudioAnalyzer *analyzer) {  // lib.rs:439-446
  delete const_cast<NeedleAudioAnalyzer *>(analyzer);
}

void needle_audio_analyzer_print_paths(const struct NeedleAudioAnalyzer *analyzer) {  // lib.rs:450-461
  if (!analyzer) return;
  for (const std::string &p : analyzer->inner.videos()) std::printf("%s\n", p.c_str());
  std::fflush(stdout);
}

enum NeedleError needle_audio_analyzer_run(struct NeedleAudioAnalyzer *analyzer, float hash_duration, bool persist,
                                           bool threading) {  // lib.rs:465-491
  if (!analyzer) return NeedleError_NullArgument;
  if (!(hash_duration > 0.0f)) return NeedleError_AnalyzerInvalidHashDuration;  // lib.rs:474-476
  return guarded([&]() -> NeedleError {
    bool ok = true;
    const ns_t hd = duration_from_secs_f32(hash_duration, &ok);
    if (!ok) return NeedleError_AnalyzerInvalidHashDuration;
    std::vector<FrameHashesData> out;
    Status s = analyzer->inner.run(hd, persist, threading, &out);
    if (!s.ok()) return report(s);
    analyzer->frame_hashes.clear();
    for (FrameHashesData &d : out) analyzer->frame_hashes.push_back(FrameHashes{std::move(d)});
    return NeedleError_Ok;
  });
}

enum NeedleError needle_audio_comparator_new(const char *const *paths, size_t num_paths, bool include_endings,
                                             uint16_t hash_match_threshold, uint16_t min_opening_duration,
                                             uint16_t min_ending_duration, float time_padding,
                                             const struct NeedleAudioComparator **output) {  // lib.rs:556-597
  if (!paths || !output) return NeedleError_NullArgument;
  if (num_paths < 2) return NeedleError_ComparatorMinimumPaths;
  return guarded([&]() -> NeedleError {
    std::vector<std::string> p;
    NeedleError e = paths_from_raw(paths, num_paths, &p);
    if (e != NeedleError_Ok) return e;
    bool ok = true;
    const ns_t pad = duration_from_secs_f32(time_padding, &ok);  // Duration::from_secs_f32 panics on bad input
    if (!ok) return report(Status::Make(NeedleError_InvalidArgument, "time_padding must be a finite, non-negative number"));
    auto *c = new NeedleAudioComparator();
    c->inner = Comparator::from_files(std::move(p));
    c->inner.with_include_endings(include_endings)
        .with_hash_match_threshold(hash_match_threshold)
        .with_min_opening_duration((ns_t)min_opening_duration * kNanosPerSec)
        .with_min_ending_duration((ns_t)min_ending_duration * kNanosPerSec)
        .with_time_padding(pad);
    *output = c;
    return NeedleError_Ok;
  });
}

enum NeedleError needle_audio_comparator_new_default(const char *const *paths, size_t num_paths,
                                                     const struct NeedleAudioComparator **output) {  // lib.rs:537-552
  return needle_audio_comparator_new(paths, num_paths, false, DEFAULT_HASH_MATCH_THRESHOLD,
                                     DEFAULT_MIN_OPENING_DURATION, DEFAULT_MIN_ENDING_DURATION, 0.0f, output);
}

void needle_audio_comparator_free(const struct NeedleAudioComparator *comparator) {  // lib.rs:601-608
  delete const_cast<NeedleAudioComparator *>(comparator);
}

enum NeedleError needle_audio_comparator_run(const struct NeedleAudioComparator *comparator, bool analyze,
                                             bool display, bool use_skip_files, bool write_skip_files,
                                             bool threading) {  // lib.rs:612-637
  if (!comparator) return NeedleError_NullArgument;
  return guarded([&]() -> NeedleError {
    const bool trace = std::getenv("NEEDLE_HIP_TRACE") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<VideoResult> res;
    Status s = comparator->inner.run(analyze, display, use_skip_files, write_skip_files, threading, &res);
    if (trace)
      std::fprintf(stderr, "[needle_hip] comparator run, whole call: %.2f ms\n",
                   std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    return s.ok() ? NeedleError_Ok : report(s);
  });
}

// ============================================================================================================
// needle_hip.h — device / diagnostics
// ============================================================================================================
enum NeedleError needle_hip_device_count(int *count) {
  if (!count) return NeedleError_NullArgument;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    n = 0;
  }
  *count = n;
  return NeedleError_Ok;
}

enum NeedleError needle_hip_set_device(int ordinal) {
  return guarded([&]() -> NeedleError {
    Status s = ensure_device();
    if (!s.ok()) return report(s);
    if (hipSetDevice(ordinal) != hipSuccess) {
      (void)hipGetLastError();
      return report(Status::Make(NeedleError_InvalidArgument, "no such HIP device: " + std::to_string(ordinal)));
    }
    return NeedleError_Ok;
  });
}

enum NeedleError needle_hip_device_pci_bus_id(char out[32]) {
  if (!out) return NeedleError_NullArgument;
  return guarded([&]() -> NeedleError {
    Status s = ensure_device();
    if (!s.ok()) return report(s);
    int dev = 0;
    out[0] = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetPCIBusId(out, 32, dev) != hipSuccess) {
      (void)hipGetLastError();
      return report(Status::Make(NeedleError_Unknown, "the HIP runtime did not name the device's PCI address"));
    }
    return NeedleError_Ok;
  });
}

enum NeedleError needle_hip_synchronize(void) {
  return guarded([&]() -> NeedleError {
    Status s = ensure_device();
    if (!s.ok()) return report(s);
    if (hipStreamSynchronize(library_stream()) != hipSuccess || hipDeviceSynchronize() != hipSuccess)
      return report(Status::Make(NeedleError_Unknown, std::string("HIP error: ") + hipGetErrorString(hipGetLastError())));
    return NeedleError_Ok;
  });
}

void *needle_hip_stream(void) {
  void *out = nullptr;
  guarded([&]() -> NeedleError {
    Status s = ensure_device();
    if (!s.ok()) return report(s);
    out = reinterpret_cast<void *>(library_stream());
    return NeedleError_Ok;
  });
  return out;
}

const char *needle_hip_last_error_message(void) { return last_error(); }
const char *needle_hip_version(void) { return "needle-mi355x 0.1.0 (gfx950, f64 fingerprint, exact integer search)"; }

enum NeedleError needle_hip_malloc(void **device_ptr, size_t bytes) {
  if (!device_ptr) return NeedleError_NullArgument;
  return guarded([&]() -> NeedleError {
    Status s = ensure_device();
    if (!s.ok()) return report(s);
    if (hipMalloc(device_ptr, bytes ? bytes : 1) != hipSuccess)
      return report(Status::Make(NeedleError_Unknown, "hipMalloc failed for " + std::to_string(bytes) + " bytes"));
    return NeedleError_Ok;
  });
}

enum NeedleError needle_hip_free(void *device_ptr) {
  if (device_ptr && hipFree(device_ptr) != hipSuccess) return report(Status::Make(NeedleError_Unknown, "hipFree failed"));
  return NeedleError_Ok;
}

enum NeedleError needle_hip_memcpy_h2d(void *device_dst, const void *host_src, size_t bytes) {
  if (!device_dst || !host_src) return NeedleError_NullArgument;
  hipStream_t st = library_stream();
  if (hipMemcpyAsync(device_dst, host_src, bytes, hipMemcpyHostToDevice, st) != hipSuccess ||
      hipStreamSynchronize(st) != hipSuccess)
    return report(Status::Make(NeedleError_Unknown, "host-to-device copy failed"));
  return NeedleError_Ok;
}

enum NeedleError needle_hip_memcpy_d2h(void *host_dst, const void *device_src, size_t bytes) {
  if (!host_dst || !device_src) return NeedleError_NullArgument;
  hipStream_t st = library_stream();
  if (hipMemcpyAsync(host_dst, device_src, bytes, hipMemcpyDeviceToHost, st) != hipSuccess ||
      hipStreamSynchronize(st) != hipSuccess)
    return report(Status::Make(NeedleError_Unknown, "device-to-host copy failed"));
  return NeedleError_Ok;
}

void needle_hip_host_free(void *ptr) { std::free(ptr); }

enum NeedleError needle_hip_host_alloc(void **host_ptr, size_t bytes) {
  if (!host_ptr) return NeedleError_NullArgument;
  return guarded([&]() -> NeedleError {
    Status s = ensure_device();
    if (!s.ok()) return report(s);
    if (hipHostMalloc(host_ptr, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess)
      return report(Status::Make(NeedleError_Unknown, "pinned allocation of " + std::to_string(bytes) + " bytes failed"));
    return NeedleError_Ok;
  });
}

enum NeedleError needle_hip_host_alloc_free(void *host_ptr) {
  if (host_ptr && hipHostFree(host_ptr) != hipSuccess) return report(Status::Make(NeedleError_Unknown, "hipHostFree failed"));
  return NeedleError_Ok;
}

enum NeedleError needle_hip_fingerprint_cert_stats(uint64_t counts[4], bool reset) {
  if (!counts) return NeedleError_NullArgument;
  return guarded([&]() -> NeedleError {
    Status s = gpu_fingerprint_cert_stats(counts, reset);
    return s.ok() ? NeedleError_Ok : report(s);
  });
}

enum NeedleError needle_hip_scan_issued_evaluations(uint64_t *lane_evaluations, bool reset) {
  if (!lane_evaluations) return NeedleError_NullArgument;
  return guarded([&]() -> NeedleError {
    Status s = gpu_scan_issued_evaluations(lane_evaluations, reset);
    return s.ok() ? NeedleError_Ok : report(s);
  });
}

enum NeedleError needle_hip_scan_counts(uint64_t counts[2], bool reset) {
  if (!counts) return NeedleError_NullArgument;
  return guarded([&]() -> NeedleError {
    Status s = gpu_scan_issued_evaluations(&counts[0], reset, &counts[1]);
    return s.ok() ? NeedleError_Ok : report(s);
  });
}

enum NeedleError needle_hip_scan_last_launch(int32_t *form, uint64_t *matrix_products) {
  if (!form || !matrix_products) return NeedleError_NullArgument;
  gpu_scan_last_launch(form, matrix_products);
  return NeedleError_Ok;
}

enum NeedleError needle_hip_epilogue_host_fallbacks(uint64_t *jobs, bool reset) {
  if (!jobs) return NeedleError_NullArgument;
  *jobs = reset ? epilogue_host_fallbacks().exchange(0) : epilogue_host_fallbacks().load();
  return NeedleError_Ok;
}

enum NeedleError needle_hip_int_valu_ceiling(double *cells_per_second) {
  if (!cells_per_second) return NeedleError_NullArgument;
  return guarded([&]() -> NeedleError {
    Status s = gpu_int_valu_ceiling(cells_per_second);
    return s.ok() ? NeedleError_Ok : report(s);
  });
}

double needle_hip_last_kernel_ms(const char *kernel) { return kernel ? kernel_ms(kernel) : -1.0; }
void needle_hip_set_kernel_timing(const char *kernels) { set_kernel_timing(kernels); }

// ============================================================================================================
// fingerprint
// ============================================================================================================
int needle_hip_fingerprint_sample_rate(void) { return kSampleRate; }
int needle_hip_fingerprint_delay_ms(void) { return kDelayMs; }
int needle_hip_fingerprint_item_duration_ms(void) { return kItemDurationMs; }
size_t needle_hip_fingerprint_num_items(size_t samples_per_channel) { return num_items(samples_per_channel); }
size_t needle_hip_fingerprint_num_kept(size_t samples_per_channel, uint32_t step) {
  return num_kept(samples_per_channel, step);
}

enum NeedleError needle_hip_fingerprint_host(const int16_t *const *pcm, const size_t *num_values, size_t num_streams,
                                             int channels, uint32_t step, uint32_t *const *items) {
  if (!pcm || !num_values || !items) return NeedleError_NullArgument;
  return guarded([&]() -> NeedleError {
    std::vector<const int16_t *> p(pcm, pcm + num_streams);
    std::vector<size_t> n(num_values, num_values + num_streams);
    for (size_t i = 0; i < num_streams; i++)
      if ((!p[i] && n[i]) || !items[i]) return NeedleError_NullArgument;
    std::vector<std::vector<uint32_t>> out;
    Status s = gpu_fingerprint_host(p, n, channels, step, &out);
    if (!s.ok()) return report(s);
    for (size_t i = 0; i < num_streams; i++)
      if (!out[i].empty()) std::memcpy(items[i], out[i].data(), out[i].size() * sizeof(uint32_t));
    return NeedleError_Ok;
  });
}

enum NeedleError needle_hip_fingerprint_device(const int16_t *d_pcm, const uint64_t *pcm_offsets,
                                               const uint64_t *num_values, size_t num_streams, int channels,
                                               uint32_t step, uint32_t *d_items, const uint64_t *item_offsets,
                                               bool sync) {
  if (!d_pcm || !pcm_offsets || !num_values || !d_items || !item_offsets) return NeedleError_NullArgument;
  return guarded([&]() -> NeedleError {
    std::vector<StreamSpan> spans(num_streams);
    for (size_t i = 0; i < num_streams; i++) spans[i] = StreamSpan{pcm_offsets[i], num_values[i], item_offsets[i]};
    Status s = gpu_fingerprint_device(d_pcm, spans, channels, step, d_items, sync);
    return s.ok() ? NeedleError_Ok : report(s);
  });
}

enum NeedleError needle_hip_fingerprint_audit_device(const int16_t *d_pcm, const uint64_t *pcm_offsets, const uint64_t *num_values,
                                                     size_t num_streams, int channels, uint32_t step, const uint32_t *d_items,
                                                     const uint64_t *item_offsets, NeedleHipCertAudit *audit) {
  if (!d_pcm || !pcm_offsets || !num_values || !d_items || !item_offsets || !audit) return NeedleError_NullArgument;
  return guarded([&]() -> NeedleError {
    std::vector<StreamSpan> spans(num_streams);
    for (size_t i = 0; i < num_streams; i++) spans[i] = StreamSpan{pcm_offsets[i], num_values[i], item_offsets[i]};
    uint64_t c[4] = {0, 0, 0, 0};
    Status s = gpu_fingerprint_audit_device(d_pcm, spans, channels, step, d_items, c, &audit->max_error_over_s, &audit->max_s);
    if (!s.ok()) return report(s);
    audit->items = c[0];
    audit->accepted = c[1];
    audit->accepted_mismatches = c[2];
    audit->mismatches = c[3];
    return NeedleError_Ok;
  });
}

enum NeedleError needle_hip_fingerprint_debug(const int16_t *pcm, size_t num_values, int channels, double *chroma,
                                              double *features) {
  if (!pcm) return NeedleError_NullArgument;
  return guarded([&]() -> NeedleError {
    Status s = ensure_device();
    if (!s.ok()) return report(s);
    if (channels != 1 && channels != 2) return NeedleError_InvalidArgument;
    const size_t samples = num_values / (size_t)channels;
    const size_t frames = num_frames(samples), rows = frames >= 5 ? frames - 4 : 0, kept = num_items(samples);
    DeviceBuffer<int16_t> d_pcm;
    DeviceBuffer<uint32_t> d_items;
    DeviceBuffer<double> d_chroma, d_feat;
    if (!(s = d_pcm.reserve(std::max<size_t>(num_values, 1))).ok()) return report(s);
    if (!(s = d_items.reserve(std::max<size_t>(kept, 1))).ok()) return report(s);
    if (!(s = d_chroma.reserve(std::max<size_t>(frames, 1) * kBands)).ok()) return report(s);
    if (!(s = d_feat.reserve(std::max<size_t>(rows, 1) * kBands)).ok()) return report(s);
    if (hipMemcpy(d_pcm.ptr, pcm, num_values * sizeof(int16_t), hipMemcpyHostToDevice) != hipSuccess)
      return report(Status::Make(NeedleError_Unknown, "host-to-device copy failed"));
    std::vector<StreamSpan> spans{StreamSpan{0, num_values, 0}};
    s = gpu_fingerprint_device(d_pcm.ptr, spans, channels, 1, d_items.ptr, true, d_chroma.ptr, d_feat.ptr);
    if (!s.ok()) return report(s);
    if (chroma && frames && hipMemcpy(chroma, d_chroma.ptr, frames * kBands * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)
      return report(Status::Make(NeedleError_Unknown, "device-to-host copy failed"));
    if (features && rows && hipMemcpy(features, d_feat.ptr, rows * kBands * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)
      return report(Status::Make(NeedleError_Unknown, "device-to-host copy failed"));
    return NeedleError_Ok;
  });
}

// ============================================================================================================
// resampler front-end
// ============================================================================================================
size_t needle_hip_resample_out_len(size_t samples_per_channel, int sample_rate) {
  return sample_rate > 0 ? resample_out_len(samples_per_channel, sample_rate) : 0;
}

enum NeedleError needle_hip_resample_host(const int16_t *const *pcm, const size_t *num_values, size_t num_streams,
                                          int channels, int sample_rate, int16_t *const *out) {
  if (!pcm || !num_values || !out) return NeedleError_NullArgument;
  return guarded([&]() -> NeedleError {
    std::vector<const int16_t *> p(pcm, pcm + num_streams);
    std::vector<size_t> n(num_values, num_values + num_streams);
    for (size_t i = 0; i < num_streams; i++)
      if ((!p[i] && n[i]) || !out[i]) return NeedleError_NullArgument;
    std::vector<std::vector<int16_t>> res;
    Status s = gpu_resample_host(p, n, channels, sample_rate, &res);
    if (!s.ok()) return report(s);
    for (size_t i = 0; i < num_streams; i++)
      if (!res[i].empty()) std::memcpy(out[i], res[i].data(), res[i].size() * sizeof(int16_t));
    return NeedleError_Ok;
  });
}

// ============================================================================================================
// search
// ============================================================================================================
enum NeedleError needle_hip_hamming_runs_device(const uint32_t *d_hashes, const NeedleHipSeq *seqs, size_t num_seqs,
                                                const NeedleHipProblem *problems, size_t num_problems,
                                                uint32_t threshold, NeedleHipRun *d_runs, uint32_t capacity,
                                                uint32_t *d_count, bool sync) {
  if (!d_hashes || !seqs || (!problems && num_problems) || !d_runs || !d_count) return NeedleError_NullArgument;
  return guarded([&]() -> NeedleError {
    Status s = gpu_hamming_runs_device(d_hashes, seqs, num_seqs, problems, num_problems, threshold, d_runs, capacity,
                                       d_count, sync);
    return s.ok() ? NeedleError_Ok : report(s);
  });
}

enum NeedleError needle_hip_hamming_runs_host(const uint32_t *hashes, size_t num_hashes, const NeedleHipSeq *seqs,
                                              size_t num_seqs, const NeedleHipProblem *problems, size_t num_problems,
                                              uint32_t threshold, NeedleHipRun **runs, size_t *num_runs) {
  if ((!hashes && num_hashes) || !seqs || (!problems && num_problems) || !runs || !num_runs) return NeedleError_NullArgument;
  return guarded([&]() -> NeedleError {
    std::vector<NeedleHipRun> out;
    Status s = gpu_hamming_runs_host(hashes, num_hashes, seqs, num_seqs, problems, num_problems, threshold, &out);
    if (!s.ok()) return report(s);
    NeedleHipRun *arr = static_cast<NeedleHipRun *>(std::malloc(std::max<size_t>(out.size(), 1) * sizeof(NeedleHipRun)));
    if (!arr) return report(Status::Make(NeedleError_Unknown, "out of memory"));
    if (!out.empty()) std::memcpy(arr, out.data(), out.size() * sizeof(NeedleHipRun));
    *runs = arr;
    *num_runs = out.size();
    return NeedleError_Ok;
  });
}

// ============================================================================================================
// FrameHashes
// ============================================================================================================
enum NeedleError needle_hip_frame_hashes_new(const uint32_t *opening_hashes, const uint64_t *opening_ts_ns,
                                             size_t num_opening, const uint32_t *ending_hashes,
                                             const uint64_t *ending_ts_ns, size_t num_ending,
                                             uint64_t hash_duration_ns, const char *md5, FrameHashes **output) {
  if (!output || (num_opening && (!opening_hashes || !opening_ts_ns)) || (num_ending && (!ending_hashes || !ending_ts_ns)))
    return NeedleError_NullArgument;
  return guarded([&]() -> NeedleError {
    auto *fh = new FrameHashes();
    fh->d.opening.resize(num_opening);
    for (size_t i = 0; i < num_opening; i++) fh->d.opening[i] = HashTs{opening_hashes[i], opening_ts_ns[i]};
    fh->d.ending.resize(num_ending);
    for (size_t i = 0; i < num_ending; i++) fh->d.ending[i] = HashTs{ending_hashes[i], ending_ts_ns[i]};
    fh->d.hash_duration = hash_duration_ns;
    fh->d.md5 = md5 ? md5 : "";
    *output = fh;
    return NeedleError_Ok;
  });
}

void needle_hip_frame_hashes_free(FrameHashes *frame_hashes) { delete frame_hashes; }

size_t needle_hip_frame_hashes_len(const FrameHashes *fh, bool ending) {
  return !fh ? 0 : (ending ? fh->d.ending.size() : fh->d.opening.size());
}

enum NeedleError needle_hip_frame_hashes_copy(const FrameHashes *fh, bool ending, uint32_t *hashes, uint64_t *ts_ns,
                                              size_t capacity) {
  if (!fh) return NeedleError_NullArgument;
  const std::vector<HashTs> &v = ending ? fh->d.ending : fh->d.opening;
  if (capacity < v.size()) return NeedleError_InvalidArgument;
  for (size_t i = 0; i < v.size(); i++) {
    if (hashes) hashes[i] = v[i].hash;
    if (ts_ns) ts_ns[i] = v[i].ts;
  }
  return NeedleError_Ok;
}

uint64_t needle_hip_frame_hashes_hash_duration_ns(const FrameHashes *fh) { return fh ? fh->d.hash_duration : 0; }
const char *needle_hip_frame_hashes_md5(const FrameHashes *fh) { return fh ? fh->d.md5.c_str() : ""; }

enum NeedleError needle_hip_frame_hashes_read(const char *path, FrameHashes **output) {
  if (!path || !output) return NeedleError_NullArgument;
  return guarded([&]() -> NeedleError {
    auto fh = new FrameHashes();
    Status s = frame_hashes_read(path, &fh->d);
    if (!s.ok()) {
      delete fh;
      return report(s);
    }
    *output = fh;
    return NeedleError_Ok;
  });
}

enum NeedleError needle_hip_frame_hashes_write(const FrameHashes *fh, const char *path) {
  if (!fh || !path) return NeedleError_NullArgument;
  return guarded([&]() -> NeedleError {
    Status s = frame_hashes_write(path, fh->d);
    return s.ok() ? NeedleError_Ok : report(s);
  });
}

enum NeedleError needle_hip_header_md5(const char *path, char out[33]) {
  if (!path || !out) return NeedleError_NullArgument;
  return guarded([&]() -> NeedleError {
    std::string md5;
    Status s = header_md5(path, &md5);
    if (!s.ok()) return report(s);
    std::memcpy(out, md5.c_str(), 33);
    return NeedleError_Ok;
  });
}

// ============================================================================================================
// Analyzer / Comparator in memory
// ============================================================================================================
enum NeedleError needle_hip_analyzer_run_pcm(struct NeedleAudioAnalyzer *analyzer, const int16_t *const *pcm,
                                             const size_t *num_values, int channels, int sample_rate,
                                             float hash_duration, bool persist) {
  if (!analyzer || !pcm || !num_values) return NeedleError_NullArgument;
  if (!(hash_duration > 0.0f)) return NeedleError_AnalyzerInvalidHashDuration;
  return guarded([&]() -> NeedleError {
    bool ok = true;
    const ns_t hd = duration_from_secs_f32(hash_duration, &ok);
    if (!ok) return NeedleError_AnalyzerInvalidHashDuration;
    const size_t n = analyzer->inner.videos().size();
    std::vector<PcmView> views(n);
    for (size_t i = 0; i < n; i++) {
      if (!pcm[i] && num_values[i]) return NeedleError_NullArgument;
      views[i] = PcmView{pcm[i], num_values[i]};
    }
    std::vector<FrameHashesData> out;
    Status s = analyzer->inner.run_pcm(views, channels, sample_rate, hd, persist, &out);
    if (!s.ok()) return report(s);
    analyzer->frame_hashes.clear();
    for (FrameHashesData &d : out) analyzer->frame_hashes.push_back(FrameHashes{std::move(d)});
    return NeedleError_Ok;
  });
}

enum NeedleError needle_hip_comparator_run_with_frame_hashes(const struct NeedleAudioComparator *comparator,
                                                             const FrameHashes *const *frame_hashes,
                                                             size_t num_videos, bool display, bool use_skip_files,
                                                             bool write_skip_files, NeedleHipSearchResult *results) {
  if (!comparator || !frame_hashes || !results) return NeedleError_NullArgument;
  return guarded([&]() -> NeedleError {
    std::vector<const FrameHashesData *> fh(num_videos);
    for (size_t i = 0; i < num_videos; i++) {
      if (!frame_hashes[i]) return NeedleError_NullArgument;
      fh[i] = &frame_hashes[i]->d;
    }
    std::vector<VideoResult> res;
    Status s = comparator->inner.run_with_frame_hashes(fh, display, use_skip_files, write_skip_files, true, &res);
    if (!s.ok()) return report(s);
    for (size_t i = 0; i < num_videos; i++) fill_result(res[i], &results[i]);
    return NeedleError_Ok;
  });
}

enum NeedleError needle_hip_comparator_results_from_runs(const struct NeedleAudioComparator *comparator,
                                                         const FrameHashes *const *frame_hashes, size_t num_videos,
                                                         const NeedleHipRun *runs, size_t num_runs, size_t first_video,
                                                         size_t video_count, NeedleHipSearchResult *results) {
  if (!comparator || !frame_hashes || !results || (!runs && num_runs)) return NeedleError_NullArgument;
  if (first_video > num_videos || video_count > num_videos - first_video) return NeedleError_InvalidArgument;
  return guarded([&]() -> NeedleError {
    std::vector<const FrameHashesData *> fh(num_videos);
    for (size_t i = 0; i < num_videos; i++) {
      if (!frame_hashes[i]) return NeedleError_NullArgument;
      fh[i] = &frame_hashes[i]->d;
    }
    std::vector<VideoResult> res;
    Status s = comparator->inner.results_from_runs(fh, runs, num_runs, false, false, false, &res, first_video,
                                                   first_video + video_count);
    if (!s.ok()) return report(s);
    for (size_t i = 0; i < num_videos; i++) fill_result(res[i], &results[i]);
    return NeedleError_Ok;
  });
}

}  // extern "C"

// library.cpp needs these two without seeing the handle layouts twice
namespace needle {
const Comparator &comparator_of(const NeedleAudioComparator *c) { return c->inner; }
FrameHashes *make_frame_hashes(FrameHashesData &&d) { return new FrameHashes{std::move(d)}; }
void fill_c_result(const VideoResult &v, NeedleHipSearchResult *r) { fill_result(v, r); }
}  // namespace needle
