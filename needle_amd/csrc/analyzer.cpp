// needle::audio::Analyzer at and above the PCM boundary (needle/src/audio/analyzer.rs).
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>

#include "needle_core.h"

namespace needle {

Analyzer Analyzer::from_files(std::vector<std::string> videos, bool threaded_decoding, bool force) {
  Analyzer a;
  a.with_threaded_decoding(threaded_decoding).with_force(force);
  a.videos_ = std::move(videos);
  return a;
}

// PCM-boundary statement of analyzer.rs:362-402.  The reference derives the stream duration from the
// container (to_timestamp: raw * time_base in f64 -> Duration::from_secs_f64, audio/util.rs:7-14),
// scales it with Duration::mul_f32 and cuts on packet timestamps; with PCM in hand the same Durations
// are converted to sample counts with integer arithmetic: samples = floor(duration_ns * rate / 1e9).
Status Analyzer::windows(size_t total_samples, int sample_rate, float opening_pct, float ending_pct,
                         size_t *opening_samples, size_t *ending_first, ns_t *ending_seek) {
  bool ok = true;
  const double time_base = 1.0 / (double)sample_rate;
  const ns_t stream_duration = duration_from_secs_f64((double)total_samples * time_base, &ok);
  if (!ok) return Status::Make(NeedleError_Unknown, "invalid stream duration");
  const ns_t opening = duration_mul_f32(stream_duration, opening_pct, &ok);                  // :378
  if (!ok) return Status::Make(NeedleError_Unknown, "invalid opening search percentage");
  const ns_t seek = duration_mul_f32(stream_duration, 1.0f - ending_pct, &ok);               // :390
  if (!ok) return Status::Make(NeedleError_Unknown, "invalid ending search percentage");
  auto to_samples = [&](ns_t d) {
    const unsigned __int128 v = (unsigned __int128)d * (unsigned)sample_rate / kNanosPerSec;
    return v > total_samples ? total_samples : (size_t)v;
  };
  *opening_samples = to_samples(opening);
  *ending_first = to_samples(seek);
  *ending_seek = seek;
  return Status::Ok();
}

// One video's search windows, already cut: what process_frames (:180-284) feeds chromaprint for the opening
// and, with include_endings, for the ending.
struct Analyzer::WindowPcm {
  const int16_t *opening = nullptr, *ending = nullptr;  // interleaved s16
  size_t opening_values = 0, ending_values = 0;
  ns_t seek = 0;  // timestamp of the first ending sample (:390)
};

Status Analyzer::fingerprint_windows(const std::vector<WindowPcm> &win, int channels, int sample_rate, uint32_t step,
                                     ns_t hash_duration, std::vector<FrameHashesData> *out) const {
  std::vector<const int16_t *> ptrs;
  std::vector<size_t> lens;
  for (const WindowPcm &w : win) {
    ptrs.push_back(w.opening);
    lens.push_back(w.opening_values);
    if (include_endings_) {
      ptrs.push_back(w.ending);
      lens.push_back(w.ending_values);
    }
  }
  std::vector<std::vector<uint32_t>> kept;
  // PCM at another rate than chromaprint's 11025 Hz goes through the device resampler (the reference
  // resamples with swresample first, :180-187); the windows are cut at the stream's own rate
  Status s = gpu_fingerprint_host(ptrs, lens, channels, step, &kept, sample_rate);
  if (!s.ok()) return s;
  out->assign(win.size(), {});
  const size_t per = include_endings_ ? 2 : 1;
  for (size_t i = 0; i < win.size(); i++) {
    FrameHashesData &fh = (*out)[i];
    attach_timestamps(kept[i * per].data(), kept[i * per].size(), step, false, 0, &fh.opening);
    if (include_endings_)
      attach_timestamps(kept[i * per + 1].data(), kept[i * per + 1].size(), step, true, win[i].seek, &fh.ending);
    fh.hash_duration = hash_duration;  // :321,411
  }
  return Status::Ok();
}

Status Analyzer::run_pcm(const std::vector<PcmView> &pcm, int channels, int sample_rate, ns_t hash_duration,
                         bool persist, std::vector<FrameHashesData> *out) const {
  if (videos_.empty())  // :431-433
    return Status::Make(NeedleError_Unknown, "no paths provided to analyzer");
  if (pcm.size() != videos_.size())
    return Status::Make(NeedleError_InvalidArgument, "one PCM stream per video is required");
  if (sample_rate < 2000 || sample_rate > 768000)
    return Status::Make(NeedleError_InvalidArgument, "unsupported sample rate");
  if (channels != 1 && channels != 2) return Status::Make(NeedleError_InvalidArgument, "channels must be 1 or 2");
  uint32_t step = 0;
  if (!step_for_hash_duration(hash_duration, &step))  // the reference panics in step_by(0), :293-304
    return Status::Make(NeedleError_AnalyzerInvalidHashDuration,
                        "hash duration is shorter than one chromaprint item (123 ms)");

  const size_t n = videos_.size();
  std::vector<WindowPcm> win(n);
  for (size_t i = 0; i < n; i++) {
    const size_t total = pcm[i].num_values / (size_t)channels;
    size_t open_samples = 0, end_first = 0;
    Status s = windows(total, sample_rate, opening_search_percentage_, ending_search_percentage_, &open_samples,
                       &end_first, &win[i].seek);
    if (!s.ok()) return s;
    win[i].opening = pcm[i].data;
    win[i].opening_values = open_samples * (size_t)channels;
    win[i].ending = pcm[i].data + end_first * (size_t)channels;
    win[i].ending_values = (total - end_first) * (size_t)channels;
  }
  Status s = fingerprint_windows(win, channels, sample_rate, step, hash_duration, out);
  if (!s.ok()) return s;
  for (size_t i = 0; i < n; i++) {
    FrameHashesData &fh = (*out)[i];
    // md5 of the first 8 KiB of the video file when it exists (in-memory callers may pass paths that
    // do not): an unreadable header leaves the key empty instead of failing the analysis.
    std::string md5;
    if (header_md5(videos_[i], &md5).ok()) fh.md5 = md5;
    if (persist) {  // :414-417
      s = frame_hashes_write(with_extension(videos_[i], FRAME_HASH_DATA_FILE_NAME), fh);
      if (!s.ok()) return s;
    }
  }
  return Status::Ok();
}

namespace {

struct Trace {  // NEEDLE_HIP_TRACE=1: phase times of the file analyzer on stderr
  const bool on = std::getenv("NEEDLE_HIP_TRACE") != nullptr;
  std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
  void lap(const char *what, size_t items) {
    if (!on) return;
    const auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[needle_hip] analyzer %s (%zu): %.2f ms\n", what, items,
                 std::chrono::duration<double, std::milli>(now - t).count());
    t = now;
  }
};

// One video that needs analysis: where its windows lie in the file.
struct Pending {
  size_t index = 0;  // position in videos_
  WavInfo info;
  size_t opening_frames = 0, ending_first = 0;
  ns_t seek = 0;
};

}  // namespace

// Analyzer::run (:425-446).  The reference maps run_single over the videos (rayon when `threading`); here the
// per-video work that is not the fingerprint -- header MD5, cache check, cutting the search windows out of the
// file -- stays on the host, and the windows of all videos of one channel count and sample rate stream to the GPU
// together: reader threads (one when `threading` is off) pread them into a ring of pinned slabs while earlier
// slabs cross PCIe, so host memory does not grow with the library.
Status Analyzer::run(ns_t hash_duration, bool persist, bool threading, std::vector<FrameHashesData> *out) const {
  if (videos_.empty())  // :431-433
    return Status::Make(NeedleError_Unknown, "no paths provided to analyzer");
  uint32_t step = 0;
  if (!step_for_hash_duration(hash_duration, &step))
    return Status::Make(NeedleError_AnalyzerInvalidHashDuration,
                        "hash duration is shorter than one chromaprint item (123 ms)");
  const size_t n = videos_.size();
  Trace trace;
  out->assign(n, {});
  std::vector<std::string> md5s(n);
  std::vector<Pending> pending;
  // The reference's sequential map (:447-451) has analysed and persisted every video in front of the one that fails
  // (run_single persists as it goes, :414-417).  The GPU batch wants all windows up front, so a video that cannot
  // be probed ends the list there: the videos before it are analysed and persisted as usual, then its error is
  // returned.  A re-run of a large library therefore keeps its progress.
  Status deferred;
  auto probe_video = [&](size_t i) -> Status {
    // run_single :339-348
    Status s = header_md5(videos_[i], &md5s[i]);
    if (!s.ok()) return s;
    if (!force_) {
      FrameHashesData existing;
      const std::string dat = with_extension(videos_[i], FRAME_HASH_DATA_FILE_NAME);
      std::ifstream probe(dat, std::ios::binary);
      if (probe) {
        probe.close();
        s = frame_hashes_read(dat, &existing);
        if (!s.ok()) return s;  // the reference unwrap()s the deserialisation
        if (existing.md5 == md5s[i]) {
          std::printf("Skipping analysis for %s...\n", videos_[i].c_str());
          std::fflush(stdout);  // println! is line buffered
          (*out)[i] = std::move(existing);
          return Status::Ok();
        }
      }
    }
    Pending p;
    p.index = i;
    s = wav_probe(videos_[i], &p.info);
    if (!s.ok()) return s;
    if (p.info.sample_rate < 2000 || p.info.sample_rate > 768000)
      return Status::Make(NeedleError_InvalidArgument, "unsupported sample rate: " + videos_[i]);
    s = windows((size_t)p.info.frames, p.info.sample_rate, opening_search_percentage_, ending_search_percentage_,
                &p.opening_frames, &p.ending_first, &p.seek);
    if (!s.ok()) return s;
    pending.push_back(p);
    return Status::Ok();
  };
  for (size_t i = 0; i < n; i++) {
    deferred = probe_video(i);
    if (!deferred.ok()) break;
  }

  trace.lap("md5 + cache check + WAV headers", n);
  // one device pass per distinct (channels, rate), videos in input order
  std::vector<std::pair<int, int>> keys;
  for (const Pending &p : pending) {
    const std::pair<int, int> key{p.info.channels, p.info.sample_rate};
    if (std::find(keys.begin(), keys.end(), key) == keys.end()) keys.push_back(key);
  }
  const unsigned readers = threading ? std::min(host_threads(), 16u) : 1u;
  const size_t per = include_endings_ ? 2 : 1;
  for (const std::pair<int, int> &key : keys) {
    std::vector<const Pending *> group;
    for (const Pending &p : pending)
      if (p.info.channels == key.first && p.info.sample_rate == key.second) group.push_back(&p);
    const size_t c = (size_t)key.first;
    std::vector<size_t> lens;  // stream 2k (+1) = opening (ending) window of group[k]
    for (const Pending *p : group) {
      lens.push_back(p->opening_frames * c);
      if (include_endings_) lens.push_back(((size_t)p->info.frames - p->ending_first) * c);
    }
    const PcmReader read = [&](size_t stream, uint64_t first_value, uint64_t num_values, int16_t *dst) {
      const Pending &p = *group[stream / per];
      const uint64_t window_first = stream % per ? p.ending_first : 0;
      return wav_read_frames(videos_[p.index], p.info, window_first + first_value / c, num_values / c, dst);
    };
    std::vector<std::vector<uint32_t>> kept;
    Status s = gpu_fingerprint_streamed(lens, read, readers, key.first, step, &kept, key.second);
    if (!s.ok()) return s;
    trace.lap("read + upload + fingerprint", group.size());
    for (size_t k = 0; k < group.size(); k++) {
      const Pending &p = *group[k];
      FrameHashesData &fh = (*out)[p.index];
      attach_timestamps(kept[k * per].data(), kept[k * per].size(), step, false, 0, &fh.opening);
      if (include_endings_)
        attach_timestamps(kept[k * per + 1].data(), kept[k * per + 1].size(), step, true, p.seek, &fh.ending);
      fh.hash_duration = hash_duration;  // :321,411
      fh.md5 = md5s[p.index];
      if (persist) {  // :414-417
        s = frame_hashes_write(with_extension(videos_[p.index], FRAME_HASH_DATA_FILE_NAME), fh);
        if (!s.ok()) return s;
      }
    }
    trace.lap("timestamps + persist", group.size());
  }
  return deferred;
}

}  // namespace needle
