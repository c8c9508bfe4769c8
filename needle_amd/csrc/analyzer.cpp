// needle::audio::Analyzer at and above the PCM boundary (needle/src/audio/analyzer.rs).
#include <cstdio>
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
  std::vector<const int16_t *> ptrs;
  std::vector<size_t> lens;
  std::vector<ns_t> seeks(n, 0);
  for (size_t i = 0; i < n; i++) {
    const size_t total = pcm[i].num_values / (size_t)channels;
    size_t open_samples = 0, end_first = 0;
    Status s = windows(total, sample_rate, opening_search_percentage_, ending_search_percentage_, &open_samples,
                       &end_first, &seeks[i]);
    if (!s.ok()) return s;
    ptrs.push_back(pcm[i].data);
    lens.push_back(open_samples * (size_t)channels);
    if (include_endings_) {
      ptrs.push_back(pcm[i].data + end_first * (size_t)channels);
      lens.push_back((total - end_first) * (size_t)channels);
    }
  }
  std::vector<std::vector<uint32_t>> kept;
  // PCM at another rate than chromaprint's 11025 Hz goes through the device resampler (the reference
  // resamples with swresample first, :180-187); the windows above are cut at the stream's own rate
  Status s = gpu_fingerprint_host(ptrs, lens, channels, step, &kept, sample_rate);
  if (!s.ok()) return s;

  out->assign(n, {});
  const size_t per = include_endings_ ? 2 : 1;
  for (size_t i = 0; i < n; i++) {
    FrameHashesData &fh = (*out)[i];
    attach_timestamps(kept[i * per].data(), kept[i * per].size(), step, false, 0, &fh.opening);
    if (include_endings_)
      attach_timestamps(kept[i * per + 1].data(), kept[i * per + 1].size(), step, true, seeks[i], &fh.ending);
    fh.hash_duration = hash_duration;  // :321,411
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

Status Analyzer::run(ns_t hash_duration, bool persist, bool /*threading*/, std::vector<FrameHashesData> *out) const {
  if (videos_.empty())  // :431-433
    return Status::Make(NeedleError_Unknown, "no paths provided to analyzer");
  uint32_t step = 0;
  if (!step_for_hash_duration(hash_duration, &step))
    return Status::Make(NeedleError_AnalyzerInvalidHashDuration,
                        "hash duration is shorter than one chromaprint item (123 ms)");
  const size_t n = videos_.size();
  out->assign(n, {});
  std::vector<char> cached(n, 0);
  std::vector<std::string> md5s(n);
  std::vector<WavData> wavs(n);
  int channels = 0, rate = kSampleRate;
  for (size_t i = 0; i < n; i++) {
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
          cached[i] = 1;
          continue;
        }
      }
    }
    s = wav_read(videos_[i], &wavs[i]);
    if (!s.ok()) return s;
    if (channels == 0) {
      channels = wavs[i].channels;
      rate = wavs[i].sample_rate;
    }
    if (wavs[i].channels != channels || wavs[i].sample_rate != rate)
      return Status::Make(NeedleError_Unknown, "all files of one run must share channel count and sample rate: " + videos_[i]);
  }
  // analyse everything that was not cached, as one GPU batch
  Analyzer sub = *this;
  sub.videos_.clear();
  std::vector<PcmView> views;
  std::vector<size_t> index;
  for (size_t i = 0; i < n; i++) {
    if (cached[i]) continue;
    sub.videos_.push_back(videos_[i]);
    views.push_back(PcmView{wavs[i].pcm.data(), wavs[i].pcm.size()});
    index.push_back(i);
  }
  if (!views.empty()) {
    std::vector<FrameHashesData> fresh;
    Status s = sub.run_pcm(views, channels, rate, hash_duration, false, &fresh);
    if (!s.ok()) return s;
    for (size_t k = 0; k < index.size(); k++) {
      fresh[k].md5 = md5s[index[k]];
      if (persist) {
        s = frame_hashes_write(with_extension(videos_[index[k]], FRAME_HASH_DATA_FILE_NAME), fresh[k]);
        if (!s.ok()) return s;
      }
      (*out)[index[k]] = std::move(fresh[k]);
    }
  }
  return Status::Ok();
}

}  // namespace needle
