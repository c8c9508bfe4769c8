// needle::audio::Comparator (needle/src/audio/comparator.rs) with the O(n*m) table sweeps replaced by
// the GPU diagonal scan (search.hip).  Everything that depends on order — the reverse table walk,
// BinaryHeap pushes, candidate numbering, tie-breaks — is reproduced on the host from the run list.
#include <chrono>
#include <cmath>
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <exception>
#include <functional>
#include <mutex>
#include <thread>
#include <tuple>
#include <unistd.h>

#include "needle_core.h"
#ifndef NEEDLE_EPILOGUE_DIRECT
#include "epilogue.h"
#endif

namespace needle {

namespace {
struct EpilogueTrace {  // NEEDLE_HIP_TRACE=1: phase times of the host epilogue on stderr
  const bool on = std::getenv("NEEDLE_HIP_TRACE") != nullptr;
  std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
  void lap(const char *what, size_t items) {
    if (!on) return;
    const auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[needle_hip] epilogue %s (%zu): %.2f ms\n", what, items,
                 std::chrono::duration<double, std::milli>(now - t).count());
    t = now;
  }
};

unsigned host_workers(uint64_t work) {  // NEEDLE_HOST_THREADS=1 forces the sequential path
  const unsigned hw = host_threads();  // this rank's share of the node's CPUs (common.h)
  return (work < (1u << 22) || hw <= 1) ? 1u : std::min(hw, 64u);
}

// Host threads that outlive a call.  A search-only call over a few hundred files is a handful of parallel phases of
// ~1 ms each (file reads, arena, heap entries, best matches): creating and joining 15 threads per phase cost as much
// as the phases themselves.  The pool grows to the largest worker count asked for and is never destroyed (its threads
// sleep on a condition variable; the process may exit underneath them).  One job at a time; a phase that is itself
// running on a pool thread (nested use) falls back to running inline.
class HostPool {
 public:
  static HostPool &instance() {
    static HostPool *pool = new HostPool();
    return *pool;
  }
  // runs body() on `workers` threads (the caller included) and returns when all of them have finished
  void run(unsigned workers, const std::function<void()> &body) {
    if (workers <= 1 || inside_) {
      body();
      return;
    }
    std::unique_lock<std::mutex> job_lock(job_mu_);  // one job at a time
    {
      std::lock_guard<std::mutex> lock(mu_);
      if (owner_ != getpid()) {  // after a fork the pool's threads exist in the parent only: start over
        new std::vector<std::thread>(std::move(threads_));  // (never destroyed: the thread objects are not joinable here)
        threads_.clear();
        owner_ = getpid();
      }
      while (threads_.size() + 1 < workers) threads_.emplace_back(&HostPool::worker, this, threads_.size());
      body_ = &body;
      wanted_ = workers - 1;  // pool threads 0 .. wanted_-1 take part
      running_ = wanted_;
      generation_++;
    }
    wake_.notify_all();
    // An exception (bad_alloc inside a phase) must not unwind past this frame while pool threads still run *body_: it
    // and its by-reference captures live on the caller's stack.  The caller's own exception, or the first one a pool
    // thread caught, is rethrown once every participant has finished.
    std::exception_ptr mine;
    try {
      body();
    } catch (...) {
      mine = std::current_exception();
    }
    std::unique_lock<std::mutex> lock(mu_);
    done_.wait(lock, [&] { return running_ == 0; });
    body_ = nullptr;
    std::exception_ptr theirs = error_;
    error_ = nullptr;
    lock.unlock();
    if (mine) std::rethrow_exception(mine);
    if (theirs) std::rethrow_exception(theirs);
  }

 private:
  void worker(size_t index) {
    inside_ = true;
    uint64_t seen = 0;
    for (;;) {
      const std::function<void()> *body = nullptr;
      {
        std::unique_lock<std::mutex> lock(mu_);
        wake_.wait(lock, [&] { return generation_ != seen; });
        seen = generation_;
        if (index < wanted_) body = body_;
      }
      if (body) {
        std::exception_ptr err;
        try {
          (*body)();
        } catch (...) {
          err = std::current_exception();
        }
        std::lock_guard<std::mutex> lock(mu_);
        if (err && !error_) error_ = err;
        if (--running_ == 0) done_.notify_all();
      }
    }
  }
  std::mutex job_mu_, mu_;
  std::condition_variable wake_, done_;
  std::vector<std::thread> threads_;
  const std::function<void()> *body_ = nullptr;
  std::exception_ptr error_;  // first exception thrown by a pool thread in the current job
  size_t wanted_ = 0, running_ = 0;
  uint64_t generation_ = 0;
  pid_t owner_ = getpid();
  static thread_local bool inside_;
};
thread_local bool HostPool::inside_ = false;

// f(begin, end) over [0, n) in chunks of `grain`, on `workers` threads (this one included)
template <class F>
void parallel_chunks(size_t n, size_t grain, unsigned workers, F f) {
  if (workers <= 1 || n <= grain) {
    if (n) f((size_t)0, n);
    return;
  }
  std::atomic<size_t> next{0};
  const std::function<void()> body = [&]() {
    for (size_t b = next.fetch_add(grain); b < n; b = next.fetch_add(grain)) f(b, std::min(n, b + grain));
  };
  HostPool::instance().run((unsigned)std::min<size_t>(workers, (n + grain - 1) / grain), body);
}

// distinct_matches (:434-454) as a count: links[a] = sum of mult[b] over {b : popcount(h[a] ^ h[b]) < bound}, a itself
// included (h holds distinct hashes, mult how often each occurs among the candidates).
// The relation is symmetric, so this equals the size of the set the reference builds for a.  Written as a dense
// c x c loop over a contiguous array so that the compiler vectorises it for whatever the host CPU offers.
#if defined(__SANITIZE_THREAD__)
// no clones: an ifunc resolver runs before the ThreadSanitizer runtime is up and crashes the process at load
#elif defined(__clang__)
__attribute__((target_clones("avx512vpopcntdq", "avx2", "default")))
#else  // g++ (the sanitizer builds) names the AVX-512 VPOPCNTDQ clone by architecture
__attribute__((target_clones("arch=icelake-server", "avx2", "default")))
#endif
void count_links_weighted(const uint32_t *h, const uint32_t *mult, size_t c, uint32_t bound, uint32_t *links) {
  for (size_t a = 0; a < c; a++) {
    const uint32_t ha = h[a];
    uint32_t count = 0;
    for (size_t b = 0; b < c; b++) count += (uint32_t)__builtin_popcount(ha ^ h[b]) < bound ? mult[b] : 0u;
    links[a] = count;
  }
}

}  // namespace


// chromaprint-rust simhash::simhash32: per bit, +1 for every set bit, -1 for every clear bit over the
// slice; output bit set iff the tally is > 0.  Counting set bits is enough: 2*ones > n.
uint32_t simhash32(const uint32_t *data, size_t n) {
  uint32_t ones[32] = {0};
  for (size_t i = 0; i < n; i++) {
    uint32_t h = data[i];
    while (h) {
      ones[__builtin_ctz(h)]++;
      h &= h - 1;
    }
  }
  uint32_t out = 0;
  for (int b = 0; b < 32; b++)
    if (2 * (uint64_t)ones[b] > n) out |= 1u << b;
  return out;
}

size_t pair_count(size_t n) { return n < 2 ? 0 : n * (n - 1) / 2; }

void pair_at(size_t n, size_t index, size_t *pi, size_t *pj) {
  // pairs (i, j), i < j, enumerated i-major (comparator.rs:537-545): row i starts at i (2n - i - 1) / 2.  The
  // root of that quadratic gives the row up to rounding; the two loops make it exact (a library has ~n^2 / 2
  // pairs and this is called for each, so it must not walk the rows).
  auto row_start = [n](size_t i) { return i * (2 * n - i - 1) / 2; };
  const double b = 2.0 * (double)n - 1.0;
  const double disc = b * b - 8.0 * (double)index;
  size_t i = disc > 0.0 ? (size_t)((b - std::sqrt(disc)) / 2.0) : 0;
  if (i + 2 > n) i = n >= 2 ? n - 2 : 0;
  while (i > 0 && row_start(i) > index) i--;
  while (i + 2 < n && row_start(i + 1) <= index) i++;
  *pi = i;
  *pj = i + 1 + (index - row_start(i));
}

Comparator Comparator::from_files(std::vector<std::string> videos) {
  Comparator c;
  c.videos_ = std::move(videos);
  return c;
}

Comparator Comparator::from_analyzer(const Analyzer &a) {
  Comparator c;
  c.videos_ = a.videos_;
  return c;
}

uint32_t Comparator::min_run_length(const std::vector<HashTs> &seq, ns_t min_duration) {
  // smallest L such that some end index i has ts[i] - ts[i-L] >= min_duration (the validity test of
  // comparator.rs:212-217 with start index i-L, :206); 0 if no window ever qualifies.
  if (seq.size() < 2) return 0;
  if (min_duration == 0) return 1;
  uint32_t best = 0;
  size_t s = 0;
  for (size_t i = 1; i < seq.size(); i++) {
    while (s + 1 < i && seq[i].ts >= seq[s + 1].ts && seq[i].ts - seq[s + 1].ts >= min_duration) s++;
    if (seq[i].ts >= seq[s].ts && seq[i].ts - seq[s].ts >= min_duration) {
      const uint32_t len = (uint32_t)(i - s);
      if (best == 0 || len < best) best = len;
    }
  }
  return best;
}

namespace {

// #[derive(Ord)] of ComparatorHeapEntry: lexicographic over the fields in declaration order
// (comparator.rs:22-35); false < true for the four flags.
auto entry_key(const HeapEntry &e) {
  return std::make_tuple(e.score, e.src_start, e.src_end, e.dst_start, e.dst_end, e.src_match_hash,
                         e.dst_match_hash, e.is_opening, !e.is_opening, e.is_opening, !e.is_opening,
                         e.src_hash_duration, e.dst_hash_duration);
}

// std::collections::BinaryHeap::push: append, then sift the new element up while it is greater than
// its parent.  `heap.into()` later returns this backing vector untouched (comparator.rs:249).
void binary_heap_push(std::vector<HeapEntry> *heap, const HeapEntry &e) {
  heap->push_back(e);
  size_t pos = heap->size() - 1;
  while (pos > 0) {
    const size_t parent = (pos - 1) / 2;
    if (entry_key((*heap)[pos]) <= entry_key((*heap)[parent])) break;
    std::swap((*heap)[pos], (*heap)[parent]);
    pos = parent;
  }
}

struct Candidate {  // comparator.rs:410-432
  ns_t start, end, hash_duration;
  uint32_t match_hash;
  bool is_opening;
};

Status read_skip_file_md5(const std::string &skip_path, std::string *md5) {
  std::ifstream f(skip_path);
  std::string text((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  const size_t key = text.find("\"md5\"");
  if (key == std::string::npos) return Status::Make(NeedleError_Unknown, "serde_json error: skip file has no md5: " + skip_path);
  const size_t q0 = text.find('"', text.find(':', key));
  const size_t q1 = q0 == std::string::npos ? q0 : text.find('"', q0 + 1);
  if (q1 == std::string::npos) return Status::Make(NeedleError_Unknown, "serde_json error: malformed skip file: " + skip_path);
  *md5 = text.substr(q0 + 1, q1 - q0 - 1);
  return Status::Ok();
}

}  // namespace

void Comparator::entries_from_runs(const NeedleHipRun *runs, size_t num_runs, const std::vector<HashTs> &src,
                                   const std::vector<HashTs> &dst, ns_t src_hash_duration, ns_t dst_hash_duration,
                                   bool is_opening, std::vector<HeapEntry> *out) const {
  out->clear();
  // `runs` is already in the order the reference pushes: i = n-1..1 and, inside, j = m-1..1 (:191-192)
  const ns_t min_duration = is_opening ? min_opening_duration_ : min_ending_duration_;
  for (size_t q = 0; q < num_runs; q++) {
    const NeedleHipRun &r = runs[q];
    const size_t i = r.src_end, j = r.dst_end, len = r.len;
    if (len == 0 || len > i || len > j || i >= src.size() || j >= dst.size()) continue;  // cannot happen
    const size_t si = i - len, sj = j - len;  // one BEFORE the first matched cell (:206-207)
    const ns_t src_start = src[si].ts, src_end = src[i].ts, dst_start = dst[sj].ts, dst_end = dst[j].ts;
    if (src_end < src_start || dst_end < dst_start) continue;  // the reference would panic on the subtraction
    if (src_end - src_start < min_duration || dst_end - dst_start < min_duration) continue;  // :212-223
    HeapEntry e;
    e.score = len;
    e.src_start = src_start;
    e.src_end = src_end;
    e.dst_start = dst_start;
    e.dst_end = dst_end;
    // simhash32 over [start_idx ..= end_idx] (L+1 hashes, :149-153,226-229) arrives with the run
    e.src_match_hash = r.src_match_hash;
    e.dst_match_hash = r.dst_match_hash;
    e.is_opening = is_opening;
    e.src_hash_duration = src_hash_duration;
    e.dst_hash_duration = dst_hash_duration;
    binary_heap_push(out, e);
  }
}

Status Comparator::best_matches(size_t num_videos, const PairEntries &pair_entries, bool display,
                                bool use_skip_files, bool write_skip_files,
                                std::vector<VideoResult> *per_video, size_t v0, size_t v1) const {
  per_video->assign(num_videos, {});
  v1 = std::min(v1, num_videos);
  v0 = std::min(v0, v1);
  auto mine = [&](size_t v) { return v >= v0 && v < v1; };
  // info_map (:583-588): for every non-empty pair, (pair, as source) for i and (pair, as dest) for j -- here as
  // one array per kind with a start offset per video, filled in pair order
  const size_t np = pair_entries.count.size();
  static thread_local std::vector<uint64_t> tl_info_first;  // scratch kept across calls, as in results_from_runs
  std::vector<uint64_t> &info_first = tl_info_first;
  info_first.assign(num_videos + 1, 0);
  {
    size_t i = 0, j = 1;
    for (size_t p = 0; p < np; p++) {
      if (pair_entries.count[p]) {
        if (mine(i)) info_first[i + 1]++;
        if (mine(j)) info_first[j + 1]++;
      }
      if (++j == num_videos) j = ++i + 1;
    }
  }
  for (size_t v = 0; v < num_videos; v++) info_first[v + 1] += info_first[v];
  struct Info {
    uint64_t pair;
    bool is_source;
  };
  static thread_local std::vector<Info> tl_info;
  std::vector<Info> &info = tl_info;
  info.resize(info_first[num_videos]);  // every element is written below
  {
    std::vector<uint64_t> fill(info_first.begin(), info_first.end() - 1);
    size_t i = 0, j = 1;
    for (size_t p = 0; p < np; p++) {
      if (pair_entries.count[p]) {
        if (mine(i)) info[fill[i]++] = Info{p, true};
        if (mine(j)) info[fill[j]++] = Info{p, false};
      }
      if (++j == num_videos) j = ++i + 1;
    }
  }
  auto info_count = [&](size_t v) { return info_first[v + 1] - info_first[v]; };
  const uint32_t bound = hash_match_threshold_ + hash_match_threshold_ / 2;  // :441
  // The reference walks the videos in order: skip-file check, find_best_match, display, skip-file write
  // (:593-626).  find_best_match -- quadratic in a video's candidate count, i.e. the dominant host cost at
  // library scale -- has no side effects, so it runs first for all videos on host threads; everything with side
  // effects (the skip-file check included: a skip file written for video k is seen by a later video with the same
  // path, and an unreadable one must not hide the output of the videos before it) then happens video by video in
  // the reference's order.
  // find_best_match (:405-515) of one video
  std::vector<Status> status(num_videos);
  auto find_best = [&](size_t v) {
    std::vector<Candidate> cand;
    for (uint64_t q = info_first[v]; q < info_first[v + 1]; q++) {
      const Info &in = info[q];
      const HeapEntry *entries = pair_entries.entries.data() + pair_entries.first[in.pair];
      for (int pass = 0; pass < 2; pass++) {  // openings first, then endings (:414-431)
        for (uint32_t k = 0; k < pair_entries.count[in.pair]; k++) {
          const HeapEntry &e = entries[k];
          if (e.is_opening != (pass == 0)) continue;
          if (in.is_source)
            cand.push_back({e.src_start, e.src_end, e.src_hash_duration, e.src_match_hash, e.is_opening});
          else
            cand.push_back({e.dst_start, e.dst_end, e.dst_hash_duration, e.dst_match_hash, e.is_opening});
        }
      }
    }
    // links[k] = #{b : popcount(h_k ^ h_b) < bound} (:434-454).  A library's candidates of one video are mostly runs
    // over the same shared segment and their simhashes repeat: the c x c comparison is done over the DISTINCT hashes,
    // weighted by how often each occurs -- the same counts, at u x u + c log c instead of c x c.
    std::vector<uint32_t> links(cand.size(), 0);
    {
      std::vector<std::pair<uint32_t, uint32_t>> order(cand.size());  // (hash, candidate)
      for (size_t k = 0; k < cand.size(); k++) order[k] = {cand[k].match_hash, (uint32_t)k};
      std::sort(order.begin(), order.end());
      std::vector<uint32_t> distinct, mult, weighted;
      for (size_t k = 0; k < order.size(); k++) {
        if (k == 0 || order[k].first != order[k - 1].first) {
          distinct.push_back(order[k].first);
          mult.push_back(0);
        }
        mult.back()++;
      }
      weighted.assign(distinct.size(), 0);
      count_links_weighted(distinct.data(), mult.data(), distinct.size(), bound, weighted.data());
      size_t u = 0;
      for (size_t k = 0; k < order.size(); k++) {
        if (k && order[k].first != order[k - 1].first) u++;
        links[order[k].second] = weighted[u];
      }
    }

    VideoResult &vr = (*per_video)[v];
    vr.has_result = true;  // Some(best) even if neither side is found (:514)
    for (int pass = 0; pass < 2; pass++) {
      const bool want_opening = pass == 0;
      if (!want_opening && !include_endings_) break;  // :486
      bool have = false;
      float best_score = 0.f;
      size_t best_idx = 0;
      for (size_t k = 0; k < cand.size(); k++) {
        if (links[k] == 0 || cand[k].is_opening != want_opening) continue;
        const float count = (float)(int64_t)links[k];
        const float duration_secs = duration_as_secs_f32(cand[k].end - cand[k].start);
        const float weighted = count * 0.3f + duration_secs * 0.7f;  // :469 (f32, no fused multiply-add)
        const float score = -weighted;
        // ascending sort of (score, index) and take the first (:473-475)
        if (!have || score < best_score) {
          have = true;
          best_score = score;
          best_idx = k;
        }
      }
      if (!have) continue;
      const Candidate &w = cand[best_idx];
      if (w.end < time_padding_ || w.end - time_padding_ < w.hash_duration) {  // Duration underflow panics upstream
        status[v] = Status::Make(NeedleError_Unknown, "overflow when subtracting durations (time_padding / hash_duration exceed the match end)");
        return;
      }
      const ns_t start = w.start + time_padding_;                   // :479
      const ns_t end = w.end - time_padding_ - w.hash_duration;     // :481
      if (want_opening) {
        vr.result.has_opening = true;
        vr.result.opening_start = start;
        vr.result.opening_end = end;
      } else {
        vr.result.has_ending = true;
        vr.result.ending_start = start;
        vr.result.ending_end = end;
      }
    }
  };
  std::vector<size_t> todo;
  uint64_t work = 0;  // candidate pairs to compare
  for (size_t v = v0; v < v1; v++) {
    if (info_count(v) == 0) continue;
    todo.push_back(v);
    uint64_t c = 0;
    for (uint64_t q = info_first[v]; q < info_first[v + 1]; q++) c += pair_entries.count[info[q].pair];
    work += c * c;
  }
  parallel_chunks(todo.size(), 1, (unsigned)std::min<size_t>(host_workers(work), std::max<size_t>(todo.size(), 1)),
                  [&](size_t b, size_t e) {
                    for (size_t k = b; k < e; k++) find_best(todo[k]);
                  });
  std::vector<uint8_t> any(num_videos, 0);
  for (size_t v = v0; v < v1; v++) any[v] = info_count(v) != 0;
  return walk_results(num_videos, any, status, display, use_skip_files, write_skip_files, per_video, v0, v1);
}

// The part of comparator.rs:593-626 with side effects, video by video in the reference's order: skip-file check, what
// find_best_match found (computed before, on host threads or on the device), display, skip-file write.
Status Comparator::walk_results(size_t num_videos, const std::vector<uint8_t> &any, const std::vector<Status> &status, bool display,
                                bool use_skip_files, bool write_skip_files, std::vector<VideoResult> *per_video, size_t v0,
                                size_t v1) const {
  v1 = std::min(v1, num_videos);
  v0 = std::min(v0, v1);
  for (size_t v = v0; v < v1; v++) {
    const std::string &path = v < videos_.size() ? videos_[v] : std::string();
    if (display) std::printf("\n%s\n\n", path.c_str());  // :595-597
    if (use_skip_files) {  // :600-605, check_skip_file :310-327
      const std::string skip = with_extension(path, SKIP_FILE_NAME);
      std::ifstream probe(skip);
      if (probe) {
        std::string md5, stored;
        Status s = header_md5(path, &md5);
        if (!s.ok()) return s;
        s = read_skip_file_md5(skip, &stored);
        if (!s.ok()) return s;
        if (stored == md5) {
          (*per_video)[v] = VideoResult{};
          if (display) std::printf("Skipping due to existing skip file...\n");
          continue;
        }
      }
    }
    if (!any[v]) {
      if (display) std::printf(include_endings_ ? "No opening or ending found.\n" : "No opening found.\n");
      continue;
    }
    if (!status[v].ok()) return status[v];
    const VideoResult &vr = (*per_video)[v];
    if (display) {  // display_opening_ending_info (:356-381)
      if (vr.result.has_opening)
        std::printf("* Opening - \"%s\"-\"%s\"\n", format_time(vr.result.opening_start).c_str(),
                    format_time(vr.result.opening_end).c_str());
      else
        std::printf("* Opening - N/A\n");
      if (include_endings_) {
        if (vr.result.has_ending)
          std::printf("* Ending - \"%s\"-\"%s\"\n", format_time(vr.result.ending_start).c_str(),
                      format_time(vr.result.ending_end).c_str());
        else
          std::printf("* Ending - N/A\n");
      }
    }
    if (write_skip_files && (vr.result.has_opening || vr.result.has_ending)) {  // create_skip_file (:329-354)
      std::string md5;
      Status s = header_md5(path, &md5);
      if (!s.ok()) return s;
      std::string json = "{\"opening\":";
      auto pair_text = [](ns_t a, ns_t b) {
        return "[" + format_f32_json(duration_as_secs_f32(a)) + "," + format_f32_json(duration_as_secs_f32(b)) + "]";
      };
      json += vr.result.has_opening ? pair_text(vr.result.opening_start, vr.result.opening_end) : "null";
      json += ",\"ending\":";
      json += vr.result.has_ending ? pair_text(vr.result.ending_start, vr.result.ending_end) : "null";
      json += ",\"md5\":\"" + md5 + "\"}";
      std::ofstream f(with_extension(path, SKIP_FILE_NAME), std::ios::trunc);
      if (!f) return Status::Make(NeedleError_IOError, "IO error: cannot create skip file for " + path);
      f << json;
    }
  }
  if (display) std::fflush(stdout);
  return Status::Ok();
}

Status Comparator::run_with_frame_hashes(const std::vector<const FrameHashesData *> &fh, bool display,
                                         bool use_skip_files, bool write_skip_files, bool /*threading*/,
                                         std::vector<VideoResult> *per_video) const {
  const size_t n = fh.size();
  const size_t regions = include_endings_ ? 2 : 1;
  EpilogueTrace trace;
  // hash arena: [video][region] sequences back to back; filled, and the minimum run lengths derived, on host threads
  std::vector<NeedleHipSeq> seqs(n * regions);
  std::vector<uint32_t> min_len(n * regions, 0);
  uint64_t total = 0;
  for (size_t v = 0; v < n; v++)
    for (size_t r = 0; r < regions; r++) {
      const std::vector<HashTs> &seq = r == 0 ? fh[v]->opening : fh[v]->ending;
      if (total + seq.size() > UINT32_MAX)  // offsets into the arena are 32-bit on the device (NeedleHipSeq.offset)
        return Status::Make(NeedleError_InvalidArgument, "library too large for one search call: more than 2^32 hashes or sequence pairs");
      seqs[v * regions + r] = NeedleHipSeq{(uint32_t)total, (uint32_t)seq.size()};
      total += seq.size();
    }
  // (in pinned host memory when there is some to be had: search.hip gpu_pinned_arena_acquire)
  struct Arena {
    uint32_t *pinned = nullptr;
    std::vector<uint32_t> pageable;
    uint32_t *data() { return pinned ? pinned : pageable.data(); }
    ~Arena() { gpu_pinned_arena_release(pinned); }
  } arena;
  arena.pinned = total ? gpu_pinned_arena_acquire(total) : nullptr;
  if (!arena.pinned) arena.pageable.resize(total);
  std::vector<uint8_t> ts_like_first(n * regions, 0);  // the row's timestamps equal video 0's of the same region (device epilogue)
  const unsigned workers = host_workers(total * 16);
  parallel_chunks(n, 8, workers, [&](size_t v0, size_t v1) {
    for (size_t v = v0; v < v1; v++)
      for (size_t r = 0; r < regions; r++) {
        const std::vector<HashTs> &seq = r == 0 ? fh[v]->opening : fh[v]->ending;
        const std::vector<HashTs> &first = r == 0 ? fh[0]->opening : fh[0]->ending;
        uint32_t *dst = arena.data() + seqs[v * regions + r].offset;
        bool same = seq.size() == first.size();
        for (size_t k = 0; k < seq.size(); k++) {
          dst[k] = seq[k].hash;
          same = same && seq[k].ts == first[k].ts;
        }
        ts_like_first[v * regions + r] = same;
        min_len[v * regions + r] = min_run_length(seq, r == 0 ? min_opening_duration_ : min_ending_duration_);
      }
  });
  if (arena.pinned) gpu_prefetch_hashes(arena.pinned, total);  // goes up while the pair table is built
  const size_t np = pair_count(n);
  if (np * regions > UINT32_MAX)  // problem tags are 32-bit on the device (NeedleHipProblem.tag)
    return Status::Make(NeedleError_InvalidArgument, "library too large for one search call: more than 2^32 hashes or sequence pairs");
  if (include_endings_)
    for (size_t v = 0; v < n && n > 1; v++)
      if (fh[v]->ending.empty())  // :271-273; the caller unwrap()s (every video is in some pair)
        return Status::Make(NeedleError_Unknown, "no ending hash data present");
  // (i, j) in the reference's i-major order (:537-545).  A pair one of whose sequences can hold no run long enough is
  // left out; when there is none such (the usual case) slot p * regions + r of the table is known up front and the
  // table is filled on host threads.
  std::vector<NeedleHipProblem> problems;
  const bool dense = std::find(min_len.begin(), min_len.end(), 0u) == min_len.end();
  if (dense) {
    problems.resize(np * regions);
    parallel_chunks(np, 8192, host_workers((uint64_t)np * 256), [&](size_t p0, size_t p1) {
      size_t i, j;
      pair_at(n, p0, &i, &j);
      for (size_t p = p0; p < p1; p++) {
        for (size_t r = 0; r < regions; r++)
          problems[p * regions + r] = NeedleHipProblem{(uint32_t)(i * regions + r), (uint32_t)(j * regions + r),
                                                       std::max(min_len[i * regions + r], min_len[j * regions + r]),
                                                       (uint32_t)(p * regions + r)};
        if (++j == n) j = ++i + 1;
      }
    });
  } else {
    problems.reserve(np * regions);
    for (size_t p = 0, i = 0, j = 1; p < np; p++) {
      for (size_t r = 0; r < regions; r++) {
        const uint32_t a = min_len[i * regions + r], b = min_len[j * regions + r];
        if (a == 0 || b == 0) continue;  // no run of this pair can satisfy the duration test
        problems.push_back(NeedleHipProblem{(uint32_t)(i * regions + r), (uint32_t)(j * regions + r), std::max(a, b),
                                            (uint32_t)(p * regions + r)});
      }
      if (++j == n) j = ++i + 1;
    }
  }
  trace.lap("arena + pair table", problems.size());
  std::vector<NeedleHipRun> runs;
#ifndef NEEDLE_EPILOGUE_DIRECT
  // Library scale: scan AND per-video epilogue on the device (epilogue.hip), only the n results come back.  The arena's
  // rows are the videos' windows with their own timestamps; identical timestamp tables (videos of equal length) go up once.
  bool on_device = (uint64_t)np * regions >= kDeviceEpiloguePairs && n >= 2;
  if (const char *e = std::getenv("NEEDLE_HIP_DEVICE_EPILOGUE")) on_device = std::atoi(e) != 0 && n >= 2;
  if (on_device) {
    std::vector<uint32_t> row_len(n * regions), row_ts(n * regions);
    std::vector<uint64_t> row_seek(n * regions, 0), ts;
    for (size_t row = 0; row < n * regions; row++) {  // a library's videos mostly share their timestamps: one table each
      const std::vector<HashTs> &seq = row % regions == 0 ? fh[row / regions]->opening : fh[row / regions]->ending;
      row_len[row] = (uint32_t)seq.size();
      if (row >= regions && ts_like_first[row]) {
        row_ts[row] = row_ts[row % regions];
        continue;
      }
      row_ts[row] = (uint32_t)ts.size();
      for (const HashTs &h : seq) ts.push_back(h.ts);
    }
    EpilogueJob job;
    job.slot = 2;  // a workspace of its own, beside the two job slots of a library
    job.n = (uint32_t)n;
    job.regions = job.rows_per_video = (uint32_t)regions;
    job.v0 = 0;
    job.v1 = (uint32_t)n;
    job.threshold = hash_match_threshold_;
    job.include_endings = include_endings_;
    job.min_opening_duration = min_opening_duration_;
    job.min_ending_duration = min_ending_duration_;
    job.time_padding = time_padding_;
    job.hash_duration = n ? fh[0]->hash_duration : 0;
    bool uniform = true;  // Candidate.hash_duration is per video upstream (:410-432): one value on the device
    for (size_t v = 0; v < n; v++) uniform = uniform && fh[v]->hash_duration == job.hash_duration;
    job.row_len = &row_len;
    job.row_ts = &row_ts;
    job.row_seek = &row_seek;
    job.ts = &ts;
    std::vector<NeedleHipSearchResult> results;
    uint32_t failed = 0;
    size_t found = 0;
    if (uniform) {
      Status s = gpu_search_results_host(arena.data(), total, seqs.data(), seqs.size(), problems.data(), problems.size(),
                                         hash_match_threshold_, job, &results, &failed, &runs, &found);
      if (!s.ok()) return s;
      trace.lap("upload + scan + simhash + device epilogue", found);
      if (failed == 0) {
        per_video->assign(n, {});
        std::vector<uint8_t> any(n, 0);
        for (size_t v = 0; v < n; v++) {
          const NeedleHipSearchResult &r = results[v];
          any[v] = r.has_result;
          VideoResult &vr = (*per_video)[v];
          vr.has_result = r.has_result;
          vr.result.has_opening = r.has_opening;
          vr.result.has_ending = r.has_ending;
          vr.result.opening_start = r.opening_start_ns;
          vr.result.opening_end = r.opening_end_ns;
          vr.result.ending_start = r.ending_start_ns;
          vr.result.ending_end = r.ending_end_ns;
        }
        const std::vector<Status> status(n);
        Status w = walk_results(n, any, status, display, use_skip_files, write_skip_files, per_video, 0, n);
        trace.lap("display / skip files per video", n);
        return w;
      }
      if (failed & kEpilogueBucketTooLarge) note_epilogue_host_fallback("Comparator::run_with_frame_hashes", runs.size(), n);
      return results_from_runs(fh, runs.data(), runs.size(), display, use_skip_files, write_skip_files, per_video);
    }
  }
#endif
  // an empty problem list still goes through the device entry point: there is no CPU path to fall to
  Status s = gpu_hamming_runs_host(arena.data(), total, seqs.data(), seqs.size(), problems.data(),
                                   problems.size(), hash_match_threshold_, &runs);
  if (!s.ok()) return s;
  trace.lap("upload + scan + simhash + download", runs.size());
  return results_from_runs(fh, runs.data(), runs.size(), display, use_skip_files, write_skip_files, per_video);
}

Status Comparator::results_from_runs(const std::vector<const FrameHashesData *> &fh,
                                     const NeedleHipRun *runs, size_t num_runs, bool display, bool use_skip_files,
                                     bool write_skip_files, std::vector<VideoResult> *per_video, size_t v0,
                                     size_t v1) const {
  const size_t n = fh.size();
  const size_t regions = include_endings_ ? 2 : 1;
  const size_t np = pair_count(n);
  v1 = std::min(v1, n);
  v0 = std::min(v0, v1);
  // A rank that owns videos [v0, v1) needs the pairs with at least one end in that block, nothing else.
  const bool partial = v0 > 0 || v1 < n;
  std::vector<uint8_t> relevant;
  if (partial) {
    relevant.assign(np, 0);
    size_t i = 0, j = 1;
    for (size_t p = 0; p < np; p++) {
      relevant[p] = (i >= v0 && i < v1) || (j >= v0 && j < v1);
      if (++j == n) j = ++i + 1;
    }
  }
  auto wanted = [&](uint32_t problem) { return problem < np * regions && (!partial || relevant[problem / regions]); };
  // Bucket the runs by problem (NeedleHipRun.problem = pair * regions + region) with a counting sort, then order
  // each bucket the way the reference walks its table backwards: src_end descending, then dst_end descending.
  EpilogueTrace trace;
  const size_t buckets = np * regions;
  // Scratch that outlives the call (per calling thread): at library scale these are 30 + 95 + 285 MB, and a fresh
  // std::vector value-initialises every byte on one thread -- that zeroing was most of the "heap entries" phase.
  // (bound to local references here: a lambda that names a thread_local would get the EXECUTING pool thread's copy)
  static thread_local std::vector<uint64_t> tl_start;
  static thread_local std::vector<NeedleHipRun> tl_sorted;
  static thread_local PairEntries tl_pair_entries;
  std::vector<uint64_t> &start = tl_start;
  std::vector<NeedleHipRun> &sorted = tl_sorted;
  PairEntries &pair_entries = tl_pair_entries;
  start.resize(buckets + 1);
  std::fill(start.begin(), start.end(), 0);
  // Counting sort, one slice of the BUCKET range per host thread: a thread reads all runs twice (count, then place)
  // and touches only its own buckets and its own stretch of the output, so there is nothing to synchronise; a
  // library's few million runs are ~100 MB, read at memory speed, while the serial form spent its time on 2 x that
  // many random accesses into a 16 MB offset table.
  const unsigned slices = num_runs >= (1u << 16) ? std::max(1u, host_workers((uint64_t)num_runs * 64)) : 1u;
  if (slices <= 1) {
    for (size_t q = 0; q < num_runs; q++)
      if (wanted(runs[q].problem)) start[runs[q].problem + 1]++;
    for (size_t b = 0; b < buckets; b++) start[b + 1] += start[b];
    sorted.resize(start[buckets]);
    std::vector<uint64_t> fill(start.begin(), start.end() - 1);
    for (size_t q = 0; q < num_runs; q++)
      if (wanted(runs[q].problem)) sorted[fill[runs[q].problem]++] = runs[q];
  } else {
    std::vector<uint64_t> slice_total(slices + 1, 0);
    auto slice_lo = [&](unsigned t) { return (size_t)((uint64_t)buckets * t / slices); };
    parallel_chunks(slices, 1, slices, [&](size_t t0, size_t t1) {
      for (size_t t = t0; t < t1; t++) {
        const size_t lo = slice_lo((unsigned)t), hi = slice_lo((unsigned)t + 1);
        uint64_t total = 0;
        for (size_t q = 0; q < num_runs; q++) {
          const uint32_t b = runs[q].problem;
          if (b >= lo && b < hi && wanted(b)) {
            start[b + 1]++;  // bucket b's count lives at b + 1; slot hi belongs to this slice (bucket hi - 1), lo to the previous
            total++;
          }
        }
        slice_total[t + 1] = total;
      }
    });
    for (unsigned t = 0; t < slices; t++) slice_total[t + 1] += slice_total[t];
    sorted.resize(slice_total[slices]);
    parallel_chunks(slices, 1, slices, [&](size_t t0, size_t t1) {
      for (size_t t = t0; t < t1; t++) {
        const size_t lo = slice_lo((unsigned)t), hi = slice_lo((unsigned)t + 1);
        if (lo == hi) continue;
        // counts at [lo + 1, hi] -> end offsets, in place; then walk the runs again and fill from a private cursor
        uint64_t run = slice_total[t];
        std::vector<uint64_t> cursor(hi - lo);
        for (size_t b = lo; b < hi; b++) {
          cursor[b - lo] = run;
          run += start[b + 1];
          start[b + 1] = run;
        }
        for (size_t q = 0; q < num_runs; q++) {
          const uint32_t b = runs[q].problem;
          if (b >= lo && b < hi && wanted(b)) sorted[cursor[b - lo]++] = runs[q];
        }
      }
    });
  }
  trace.lap("bucket runs", sorted.size());
  // Heap entries pair by pair (both regions of a pair by the same thread: entries.extend(opening);
  // entries.extend(ending), :262-281), written where the pair's runs start: a run yields at most one entry.
  if (pair_entries.entries.size() < sorted.size()) pair_entries.entries.resize(sorted.size());  // every slot read is written first
  pair_entries.first.resize(np);
  pair_entries.count.resize(np);  // every element is written below
  parallel_chunks(np, 4096, host_workers((uint64_t)sorted.size() * 64), [&](size_t p0, size_t p1) {
    std::vector<HeapEntry> tmp;
    size_t i, j;
    pair_at(n, p0, &i, &j);
    for (size_t p = p0; p < p1; p++) {
      uint64_t out = start[p * regions];
      pair_entries.first[p] = out;
      for (size_t r = 0; r < regions; r++) {
        const uint64_t lo = start[p * regions + r], hi = start[p * regions + r + 1];
        if (hi == lo) continue;
        if (hi - lo > 1)
          std::sort(sorted.begin() + lo, sorted.begin() + hi, [](const NeedleHipRun &a, const NeedleHipRun &b) {
            return a.src_end != b.src_end ? a.src_end > b.src_end : a.dst_end > b.dst_end;
          });
        entries_from_runs(&sorted[lo], hi - lo, r == 0 ? fh[i]->opening : fh[i]->ending,
                          r == 0 ? fh[j]->opening : fh[j]->ending, fh[i]->hash_duration, fh[j]->hash_duration,
                          r == 0, &tmp);
        for (const HeapEntry &e : tmp) pair_entries.entries[out++] = e;
      }
      pair_entries.count[p] = (uint32_t)(out - pair_entries.first[p]);
      if (++j == n) j = ++i + 1;
    }
  });
  trace.lap("heap entries per pair", np);
  Status s = best_matches(n, pair_entries, display, use_skip_files, write_skip_files, per_video, v0, v1);
  trace.lap("best match per video", v1 - v0);
  return s;
}

Status Comparator::run(bool analyze, bool display, bool use_skip_files, bool write_skip_files, bool threading,
                       std::vector<VideoResult> *per_video) const {
  // The videos' hashes of a search-only call live in objects kept between calls (up to kPoolBytes of them): allocating and
  // releasing 280 vectors of 46 KB each call was 0.3 of a 2 ms call.  One call at a time owns the pool; another one running
  // beside it finds it empty and allocates as before.
  struct Pool {
    std::mutex mu;
    std::vector<FrameHashesData> data;
  };
  static Pool *pool = new Pool();  // (never destroyed: see HostPool)
  constexpr size_t kPoolBytes = (size_t)64 << 20;
  std::vector<FrameHashesData> data;
  if (!analyze) {
    std::lock_guard<std::mutex> lock(pool->mu);
    data.swap(pool->data);
  }
  data.resize(videos_.size());
  EpilogueTrace trace;
  if (!analyze) {
    // FrameHashes::from_video(video, false), data.rs:124-128, for every video.  The reads are independent; with
    // `threading` they go to host threads (a library is thousands of small files), and the error reported is the
    // one the sequential walk would have hit first.
    std::vector<Status> status(videos_.size());
    const unsigned workers = threading ? std::min(host_threads(), 16u) : 1u;
    parallel_chunks(videos_.size(), 16, videos_.size() >= 64 ? workers : 1u, [&](size_t b, size_t e) {
      for (size_t v = b; v < e; v++)
        status[v] = frame_hashes_read(with_extension(videos_[v], FRAME_HASH_DATA_FILE_NAME), &data[v]);
    });
    for (const Status &s : status)
      if (!s.ok()) return s;
  } else {
    // data.rs:134-136: default Analyzer (no endings), force, 0.3 s, not persisted — here as one GPU batch
    Analyzer a = Analyzer::from_files(videos_, false, true);
    bool ok = true;
    const ns_t hd = duration_from_secs_f32(DEFAULT_HASH_DURATION, &ok);
    Status s = a.run(hd, false, threading, &data);
    if (!s.ok()) return s;
  }
  trace.lap(analyze ? "analyze" : "read .needle.dat files", videos_.size());
  std::vector<const FrameHashesData *> ptrs;
  for (const FrameHashesData &d : data) ptrs.push_back(&d);
  Status s = run_with_frame_hashes(ptrs, display, use_skip_files, write_skip_files, threading, per_video);
  trace.lap("search (the phases above) + release of its tables", ptrs.size());
  // The videos' hash vectors were allocated on the pool's threads; released one after the other by this thread they cost
  // 0.25 ms of a 2.1 ms call over 280 files (a lock and a consolidation per 46 KB chunk): release them where they came from.
  size_t held = 0;
  for (const FrameHashesData &d : data) held += (d.opening.capacity() + d.ending.capacity()) * sizeof(HashTs);
  if (!analyze && held <= kPoolBytes) {  // kept for the next call
    std::lock_guard<std::mutex> lock(pool->mu);
    if (pool->data.empty()) pool->data.swap(data);
  }
  if (threading && data.size() >= 64)
    parallel_chunks(data.size(), 16, std::min(host_threads(), 16u), [&](size_t b, size_t e) {
      for (size_t v = b; v < e; v++) {
        std::vector<HashTs>().swap(data[v].opening);
        std::vector<HashTs>().swap(data[v].ending);
      }
    });
  trace.lap("release of the videos' hashes", data.size());
  return s;
}

}  // namespace needle
