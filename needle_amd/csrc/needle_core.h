// C++ mirror of needle::audio::{Analyzer, Comparator} (needle/src/audio/analyzer.rs:86-151,425;
// comparator.rs:74-147,524,637): same fields, same builder setters, same run methods and error
// behaviour, with the inner loops on the GPU.  The C ABI in capi.cpp wraps these exactly like
// needle-capi/src/lib.rs wraps the Rust structs.
#pragma once

#include <string>
#include <vector>

#include "common.h"

namespace needle {

// audio/mod.rs:14-45
constexpr uint16_t DEFAULT_HASH_MATCH_THRESHOLD = 10;
constexpr float DEFAULT_OPENING_SEARCH_PERCENTAGE = 0.50f;
constexpr float DEFAULT_ENDING_SEARCH_PERCENTAGE = 0.25f;
constexpr uint16_t DEFAULT_MIN_OPENING_DURATION = 20;
constexpr uint16_t DEFAULT_MIN_ENDING_DURATION = 20;
constexpr float DEFAULT_HASH_DURATION = 0.3f;
constexpr float DEFAULT_OPENING_AND_ENDING_TIME_PADDING = 0.0f;

// lib.rs:154-155
constexpr const char *FRAME_HASH_DATA_FILE_NAME = "needle.dat";
constexpr const char *SKIP_FILE_NAME = "needle.skip.json";

// One decoded stream handed to the analyzer at the PCM boundary.
struct PcmView {
  const int16_t *data = nullptr;  // interleaved s16
  size_t num_values = 0;
};

class Analyzer {
 public:
  Analyzer() = default;                                                         // analyzer.rs:95-106
  static Analyzer from_files(std::vector<std::string> videos, bool threaded_decoding, bool force);  // :110
  const std::vector<std::string> &videos() const { return videos_; }            // :119
  Analyzer &with_opening_search_percentage(float v) { opening_search_percentage_ = v; return *this; }  // :124
  Analyzer &with_ending_search_percentage(float v) { ending_search_percentage_ = v; return *this; }    // :130
  Analyzer &with_include_endings(bool v) { include_endings_ = v; return *this; }                       // :136
  Analyzer &with_threaded_decoding(bool v) { threaded_decoding_ = v; return *this; }                   // :142
  Analyzer &with_force(bool v) { force_ = v; return *this; }                                           // :148

  // Analyzer::run (:425): every video is a RIFF/WAVE file here (decode is out of scope).  Only the search
  // windows are read from each file, streamed to the GPU through a fixed ring of pinned slabs, one device
  // pass per distinct (channel count, sample rate).  `threading` threads the file reads (the fingerprint
  // batch is data-parallel on the device either way).
  Status run(ns_t hash_duration, bool persist, bool threading, std::vector<FrameHashesData> *out) const;

  // Same, with FFmpeg's half of process_frames (:180-284) done by the caller.
  Status run_pcm(const std::vector<PcmView> &pcm, int channels, int sample_rate, ns_t hash_duration,
                 bool persist, std::vector<FrameHashesData> *out) const;

  // Window arithmetic at the PCM boundary (DESIGN.md "PCM boundary"): samples per channel of the
  // opening window and first sample / seek offset of the ending window (:378,390).
  static Status windows(size_t total_samples, int sample_rate, float opening_pct, float ending_pct,
                        size_t *opening_samples, size_t *ending_first, ns_t *ending_seek);

 private:
  friend class Comparator;
  struct WindowPcm;
  Status fingerprint_windows(const std::vector<WindowPcm> &win, int channels, int sample_rate, uint32_t step,
                             ns_t hash_duration, std::vector<FrameHashesData> *out) const;
  std::vector<std::string> videos_;
  float opening_search_percentage_ = DEFAULT_OPENING_SEARCH_PERCENTAGE;
  float ending_search_percentage_ = DEFAULT_ENDING_SEARCH_PERCENTAGE;
  bool include_endings_ = false;
  bool threaded_decoding_ = false;
  bool force_ = false;
};

struct SearchResult {  // comparator.rs:65-69
  bool has_opening = false, has_ending = false;
  ns_t opening_start = 0, opening_end = 0, ending_start = 0, ending_end = 0;
};

struct VideoResult {  // Option<SearchResult> per input video (the reference drops the Nones, :608-617)
  bool has_result = false;
  SearchResult result;
};

// comparator.rs:22-35 plus the table coordinates the run came from
struct HeapEntry {
  uint64_t score;
  ns_t src_start, src_end, dst_start, dst_end;
  uint32_t src_match_hash, dst_match_hash;
  bool is_opening;  // is_src_opening == is_dst_opening == !is_*_ending for every entry the reference builds
  ns_t src_hash_duration, dst_hash_duration;
};

class Comparator {
 public:
  Comparator() = default;                                                         // comparator.rs:83-94
  static Comparator from_files(std::vector<std::string> videos);                  // :108
  static Comparator from_analyzer(const Analyzer &a);                             // :96-104
  const std::vector<std::string> &videos() const { return videos_; }              // :115
  Comparator &with_include_endings(bool v) { include_endings_ = v; return *this; }            // :120
  Comparator &with_hash_match_threshold(uint32_t v) { hash_match_threshold_ = v; return *this; }  // :126
  Comparator &with_min_opening_duration(ns_t v) { min_opening_duration_ = v; return *this; }   // :132
  Comparator &with_min_ending_duration(ns_t v) { min_ending_duration_ = v; return *this; }     // :138
  Comparator &with_time_padding(ns_t v) { time_padding_ = v; return *this; }                   // :144

  bool include_endings() const { return include_endings_; }
  uint32_t hash_match_threshold() const { return hash_match_threshold_; }

  // comparator.rs:524.  per_video gets one slot per input video; results (optional) gets the
  // reference's compacted Vec<SearchResult>.
  Status run_with_frame_hashes(const std::vector<const FrameHashesData *> &frame_hashes, bool display,
                               bool use_skip_files, bool write_skip_files, bool threading,
                               std::vector<VideoResult> *per_video) const;
  // comparator.rs:637
  Status run(bool analyze, bool display, bool use_skip_files, bool write_skip_files, bool threading,
             std::vector<VideoResult> *per_video) const;

  // ---- pieces shared with the multi-GPU Library path ----
  // Smallest run length L for which some window ts[i] - ts[i-L] reaches min_duration (0 = never).
  static uint32_t min_run_length(const std::vector<HashTs> &seq, ns_t min_duration);
  uint32_t min_run_length_for(const std::vector<HashTs> &seq, bool is_opening) const {
    return min_run_length(seq, is_opening ? min_opening_duration_ : min_ending_duration_);
  }
  // Turns the GPU's raw runs of ONE sequence pair into the reference's Vec<ComparatorHeapEntry>
  // (:191-249): reverse-walk order, duration validity, simhash32, BinaryHeap array order.
  void entries_from_runs(const NeedleHipRun *runs, size_t num_runs, const std::vector<HashTs> &src,
                         const std::vector<HashTs> &dst, ns_t src_hash_duration, ns_t dst_hash_duration,
                         bool is_opening, std::vector<HeapEntry> *out) const;
  ns_t min_opening_duration() const { return min_opening_duration_; }
  ns_t min_ending_duration() const { return min_ending_duration_; }
  ns_t time_padding() const { return time_padding_; }
  // Run list of ALL pairs (NeedleHipRun.problem = pair_index * regions + region) -> per-video results.
  // [v0, v1): the videos whose results are wanted (a rank's block in a multi-GPU job; the other slots of
  // per_video stay empty and only the pairs that touch the block are worked on).
  Status results_from_runs(const std::vector<const FrameHashesData *> &frame_hashes, const NeedleHipRun *runs,
                           size_t num_runs, bool display, bool use_skip_files, bool write_skip_files,
                           std::vector<VideoResult> *per_video, size_t v0 = 0, size_t v1 = ~(size_t)0) const;
  // Heap entries of every pair, pairs in lexicographic order, in one allocation: those of pair p are
  // entries[first[p] .. first[p] + count[p]) (a library has ~n^2 / 2 pairs: no vector per pair).
  struct PairEntries {
    std::vector<HeapEntry> entries;
    std::vector<uint64_t> first;
    std::vector<uint32_t> count;
  };
  // :583-626 — per video best match from the per-pair entries.
  Status best_matches(size_t num_videos, const PairEntries &pair_entries, bool display, bool use_skip_files,
                      bool write_skip_files, std::vector<VideoResult> *per_video, size_t v0 = 0,
                      size_t v1 = ~(size_t)0) const;
  // :593-626, the walk with side effects over results that are already there (any[v]: video v has a candidate at all)
  Status walk_results(size_t num_videos, const std::vector<uint8_t> &any, const std::vector<Status> &status, bool display,
                      bool use_skip_files, bool write_skip_files, std::vector<VideoResult> *per_video, size_t v0,
                      size_t v1) const;

 private:
  std::vector<std::string> videos_;
  bool include_endings_ = false;
  uint32_t hash_match_threshold_ = DEFAULT_HASH_MATCH_THRESHOLD;
  ns_t min_opening_duration_ = (ns_t)DEFAULT_MIN_OPENING_DURATION * kNanosPerSec;
  ns_t min_ending_duration_ = (ns_t)DEFAULT_MIN_ENDING_DURATION * kNanosPerSec;
  ns_t time_padding_ = 0;
};

uint32_t simhash32(const uint32_t *data, size_t n);  // chromaprint-rust simhash::simhash32 (comparator.rs:152)
size_t pair_count(size_t num_videos);
void pair_at(size_t num_videos, size_t index, size_t *i, size_t *j);  // lexicographic (i<j), comparator.rs:534-545

}  // namespace needle
