// The host epilogue (needle_hip_comparator_results_from_runs: counting sort of the runs, heap entries per pair on host
// threads, find_best_match per video on host threads; comparator.rs:191-249,405-515,583-626) on a synthetic run list
// large enough to take the threaded paths: NEEDLE_HOST_THREADS=1 against NEEDLE_HOST_THREADS=8, the full range against
// the union of per-block ranges (what the ranks of a multi-GPU job compute).  Host code only, no GPU: this is the
// binary the ASan/UBSan and TSan builds run (make -C needle_amd/csrc asan tsan; tests/test_sanitizers.py).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "needle_hip.h"

#ifdef NEEDLE_EPILOGUE_DIRECT
// ThreadSanitizer build: libamdhip64 cannot be loaded into a TSan process (it crashes in its own initialisers), so
// this variant links the two host translation units of the epilogue (comparator.cpp, hostutil.cpp) directly and calls
// the C++ class the C ABI wraps.  The three device-side entry points those files reference are never reached from
// results_from_runs; they are defined here to fail loudly.
#include "../../needle_amd/csrc/needle_core.h"
namespace needle {
Analyzer Analyzer::from_files(std::vector<std::string>, bool, bool) { std::abort(); }
Status Analyzer::run(ns_t, bool, bool, std::vector<FrameHashesData> *) const { std::abort(); }
Status gpu_hamming_runs_host(const uint32_t *, size_t, const NeedleHipSeq *, size_t, const NeedleHipProblem *, size_t, uint32_t,
                             std::vector<NeedleHipRun> *) { std::abort(); }
uint32_t *gpu_pinned_arena_acquire(size_t) { return nullptr; }
void gpu_pinned_arena_release(uint32_t *) {}
void gpu_prefetch_hashes(const uint32_t *, size_t) {}
}  // namespace needle
struct FrameHashes { needle::FrameHashesData d; };
struct NeedleAudioComparator { needle::Comparator inner; };
static NeedleError make_frame_hashes(const uint32_t *h, const uint64_t *ts, size_t n, FrameHashes **out) {
  *out = new FrameHashes();
  for (size_t i = 0; i < n; i++) (*out)->d.opening.push_back(needle::HashTs{h[i], ts[i]});
  (*out)->d.hash_duration = 300000012ull;
  return NeedleError_Ok;
}
static void free_frame_hashes(FrameHashes *f) { delete f; }
static NeedleError make_comparator(const char *const *paths, size_t n, const NeedleAudioComparator **out) {
  auto *c = new NeedleAudioComparator();
  c->inner = needle::Comparator::from_files(std::vector<std::string>(paths, paths + n));
  *out = c;
  return NeedleError_Ok;
}
static void free_comparator(const NeedleAudioComparator *c) { delete c; }
static NeedleError results_from_runs(const NeedleAudioComparator *cmp, FrameHashes *const *fh, size_t n, const NeedleHipRun *runs,
                                     size_t num_runs, size_t first, size_t count, NeedleHipSearchResult *results) {
  std::vector<const needle::FrameHashesData *> data(n);
  for (size_t i = 0; i < n; i++) data[i] = &fh[i]->d;
  std::vector<needle::VideoResult> res;
  needle::Status s = cmp->inner.results_from_runs(data, runs, num_runs, false, false, false, &res, first, first + count);
  if (!s.ok()) return s.code;
  for (size_t v = 0; v < n; v++) {
    results[v] = NeedleHipSearchResult{};
    results[v].has_result = res[v].has_result;
    results[v].has_opening = res[v].result.has_opening;
    results[v].opening_start_ns = res[v].result.opening_start;
    results[v].opening_end_ns = res[v].result.opening_end;
  }
  return NeedleError_Ok;
}
#else
static NeedleError make_frame_hashes(const uint32_t *h, const uint64_t *ts, size_t n, FrameHashes **out) {
  return needle_hip_frame_hashes_new(h, ts, n, nullptr, nullptr, 0, 300000012ull, "", out);
}
static void free_frame_hashes(FrameHashes *f) { needle_hip_frame_hashes_free(f); }
static NeedleError make_comparator(const char *const *paths, size_t n, const NeedleAudioComparator **out) {
  return needle_audio_comparator_new_default(paths, n, out);
}
static void free_comparator(const NeedleAudioComparator *c) { needle_audio_comparator_free(c); }
static NeedleError results_from_runs(const NeedleAudioComparator *cmp, FrameHashes *const *fh, size_t n, const NeedleHipRun *runs,
                                     size_t num_runs, size_t first, size_t count, NeedleHipSearchResult *results) {
  return needle_hip_comparator_results_from_runs(cmp, fh, n, runs, num_runs, first, count, results);
}
#endif

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd() {
  rng_state ^= rng_state << 7;
  rng_state ^= rng_state >> 9;
  return (uint32_t)(rng_state >> 16);
}

int main() {
  const size_t n = 240, len = 400;
  std::vector<FrameHashes *> fh(n);
  std::vector<uint64_t> ts(len);
  for (size_t i = 0; i < len; i++) ts[i] = 2600000000ull + 246000000ull * i;
  std::vector<uint32_t> h(len);
  for (size_t v = 0; v < n; v++) {
    for (uint32_t &x : h) x = rnd();
    if (make_frame_hashes(h.data(), ts.data(), len, &fh[v]) != NeedleError_Ok) return 2;
  }
  std::vector<std::string> names(n);
  std::vector<const char *> paths(n);
  for (size_t v = 0; v < n; v++) {
    names[v] = "video" + std::to_string(v) + ".wav";
    paths[v] = names[v].c_str();
  }
  const NeedleAudioComparator *cmp = nullptr;
  if (make_comparator(paths.data(), n, &cmp) != NeedleError_Ok) return 2;
  // one to three runs per pair, a few thousand pairs without any; simhashes drawn from a handful of clusters so that
  // find_best_match has links to count
  std::vector<NeedleHipRun> runs;
  uint32_t p = 0;
  for (size_t i = 0; i < n; i++)
    for (size_t j = i + 1; j < n; j++, p++) {
      if (rnd() % 9 == 0) continue;
      const int k = 2 + rnd() % 4;
      for (int q = 0; q < k; q++) {
        const uint32_t L = 90 + rnd() % 60, a = L + 1 + rnd() % (len - L - 2), b = L + 1 + rnd() % (len - L - 2);
        const uint32_t base = 0x0F0F0F0Fu * (1 + rnd() % 3);
        runs.push_back(NeedleHipRun{p, a, b, L, base ^ (1u << (rnd() % 32)), base ^ (1u << (rnd() % 32))});
      }
    }
  auto run = [&](const char *threads, size_t first, size_t count, std::vector<NeedleHipSearchResult> *out) {
    setenv("NEEDLE_HOST_THREADS", threads, 1);
    out->assign(n, NeedleHipSearchResult{});
    return results_from_runs(cmp, fh.data(), n, runs.data(), runs.size(), first, count, out->data());
  };
  auto same = [](const NeedleHipSearchResult &a, const NeedleHipSearchResult &b) {
    return a.has_result == b.has_result && a.has_opening == b.has_opening && a.has_ending == b.has_ending &&
           (!a.has_opening || (a.opening_start_ns == b.opening_start_ns && a.opening_end_ns == b.opening_end_ns));
  };
  std::vector<NeedleHipSearchResult> seq, par, part;
  if (run("1", 0, n, &seq) != NeedleError_Ok || run("8", 0, n, &par) != NeedleError_Ok) return 3;
  size_t with_opening = 0;
  for (size_t v = 0; v < n; v++) {
    if (!same(seq[v], par[v])) {
      std::printf("video %zu: threaded epilogue differs from the sequential one\n", v);
      return 1;
    }
    with_opening += seq[v].has_opening;
  }
  for (int world : {2, 3, 8}) {
    const size_t b = (n + world - 1) / world;
    for (int r = 0; r < world; r++) {
      const size_t first = std::min(n, (size_t)r * b), count = std::min(b, n - first);
      if (run("8", first, count, &part) != NeedleError_Ok) return 3;
      for (size_t v = 0; v < n; v++) {
        const bool mine = v >= first && v < first + count;
        if (mine ? !same(part[v], seq[v]) : part[v].has_result) {
          std::printf("world %d rank %d video %zu: block epilogue differs\n", world, r, v);
          return 1;
        }
      }
    }
  }
  for (FrameHashes *f : fh) free_frame_hashes(f);
  free_comparator(cmp);
  std::printf("epilogue ok: %zu runs, %zu videos with an opening\n", runs.size(), with_opening);
  return with_opening > n / 2 ? 0 : 1;
}
