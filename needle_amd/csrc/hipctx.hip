#include "hipctx.h"

#include <algorithm>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace needle {

namespace {
thread_local std::string g_last_error;
std::mutex g_mu;
std::map<int, hipStream_t> g_streams;

struct TimerEvents {  // two alternating event pairs, so the newest COMPLETED launch can be read without blocking
  hipEvent_t start[2] = {nullptr, nullptr}, stop[2] = {nullptr, nullptr};   // what kernel_ms reads: own[] or a caller's events
  hipEvent_t own_start[2] = {nullptr, nullptr}, own_stop[2] = {nullptr, nullptr};  // created by KernelTimer, recorded only by it
  bool recorded[2] = {false, false};
  int cur = 1;
  // "sum" mode (set_kernel_timing("...,sum")): every launch since the selection was set keeps an event pair of its own and
  // kernel_ms returns their SUM -- a kernel that a job launches several times (the first pass of a library-scale job runs
  // in three workspace-sized launches) then reads as the job's, not as its last launch's
  std::vector<hipEvent_t> hist_start, hist_stop;
  size_t hist_n = 0;
};
std::map<std::string, TimerEvents> g_timers;  // keyed by "<device>:<kernel>"

std::string timer_key(const char *name) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  return std::to_string(dev) + ":" + name;
}
}  // namespace

std::recursive_mutex &gpu_mutex() {
  static std::recursive_mutex mu;
  return mu;
}

void set_last_error(const std::string &m) { g_last_error = m; }
const char *last_error() { return g_last_error.c_str(); }

NeedleError report(const Status &s) {
  if (!s.ok()) {
    g_last_error = s.message;
    std::fprintf(stderr, "needle error: %s\n", s.message.c_str());  // needle-capi/src/lib.rs:124
  }
  return s.code;
}

Status ensure_device() {
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0) {
    (void)hipGetLastError();
    return Status::Make(NeedleError_Unknown,
                        "no HIP device: the needle analyze/search path has no CPU fallback");
  }
  return Status::Ok();
}

hipStream_t library_stream() {
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lock(g_mu);
  auto it = g_streams.find(dev);
  if (it != g_streams.end()) return it->second;
  hipStream_t s = nullptr;
  // NEEDLE_HIP_LIBRARY_PRIORITY=1 (experiment, with NEEDLE_HIP_STFT_SHARE): the library stream at the highest priority, so
  // that the tail kernels of job k are dispatched into the slots the next job's retiring STFT workgroups free
  // ... the default whenever that second stream exists (NEEDLE_HIP_STFT_SHARE not 0); NEEDLE_HIP_LIBRARY_PRIORITY=0 / 1 forces
  const char *prio = getenv("NEEDLE_HIP_LIBRARY_PRIORITY");
  const char *share = getenv("NEEDLE_HIP_STFT_SHARE");
  if (prio ? atoi(prio) != 0 : (share ? atoi(share) != 0 : true)) {
    int least = 0, greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
    if (hipStreamCreateWithPriority(&s, hipStreamNonBlocking, greatest) != hipSuccess) {
      (void)hipGetLastError();
      s = nullptr;
    }
  }
  if (!s && hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) s = nullptr;
  g_streams[dev] = s;
  return s;
}

namespace {
// Which kernels get start/stop events.  Off unless asked for: every event record is one more packet between two
// dependent dispatches, and timing all five kernels of a 28 x 24 min job costs 3 % of its step time.
struct TimingSelection {
  bool all = false, sum = false;
  std::vector<std::string> names;
  TimingSelection() {
    if (const char *e = getenv("NEEDLE_HIP_KERNEL_TIMING")) set(e);
  }
  void set(const char *list) {
    all = sum = false;
    names.clear();
    if (!list) return;
    std::string item;
    for (const char *p = list;; p++) {
      if (*p == ',' || *p == '\0') {
        if (item == "all") all = true;
        else if (item == "sum") sum = true;
        else if (!item.empty() && item != "none") names.push_back(item);
        item.clear();
        if (*p == '\0') break;
      } else if (*p != ' ') {
        item.push_back(*p);
      }
    }
  }
  bool on(const char *name) const {
    return all || std::find(names.begin(), names.end(), name) != names.end();
  }
};
TimingSelection &timing_selection() {  // callers hold g_mu
  static TimingSelection sel;
  return sel;
}
}  // namespace

void set_kernel_timing(const char *kernels) {
  std::lock_guard<std::mutex> lock(g_mu);
  timing_selection().set(kernels);
  for (auto &kv : g_timers) kv.second.hist_n = 0;  // a new selection starts a new sum (the events are kept for reuse)
}

// Downloads of results go here, behind an event of the library stream: a copy enqueued on the library stream itself
// was observed (ROCm 7.2) to complete only after the kernels enqueued BEHIND it, i.e. after the next job, which
// serialised a job's host epilogue with the following job's device work.
hipStream_t download_stream() {
  static std::map<int, hipStream_t> streams;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lock(g_mu);
  auto it = streams.find(dev);
  if (it != streams.end()) return it->second;
  hipStream_t s = nullptr;
  int least = 0, greatest = 0;
  (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
  if (hipStreamCreateWithPriority(&s, hipStreamNonBlocking, greatest) != hipSuccess) {
    (void)hipGetLastError();
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) s = nullptr;
  }
  streams[dev] = s;
  return s;
}

// The f32 STFT of a job that is pipelined behind another one goes here: a stream confined (CU mask) to all but a few
// of the device's CUs.  The kernels that follow the STFT in a job -- certification, f64 recomputation, fix-up, scan,
// simhash -- are short and latency-bound: alone they leave the chip mostly idle (0.13 of a 0.62 ms job at 28 x 24 min).
// With the NEXT job's STFT confined to its share of the CUs and running on its own queue, those kernels find the
// reserved CUs free and run underneath it.  (Stream priorities alone do not do this: a retiring STFT workgroup frees
// 35 KB of LDS and the dispatcher refills the slot before a 61 KB kernel ever fits -- NOTES §4.3, round 2.)
// nullptr: no such stream (NEEDLE_HIP_STFT_RESERVE_CUS=0, or the runtime refuses the mask).
hipStream_t stft_stream() {
  static std::map<int, hipStream_t> streams;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lock(g_mu);
  auto it = streams.find(dev);
  if (it != streams.end()) return it->second;
  hipStream_t s = nullptr;
  // OFF unless NEEDLE_HIP_STFT_RESERVE_CUS names a number of CUs to keep out of the STFT's reach.  Measured at 28 x 24
  // min (profiles/NOTES.md, round 3): with 32 reserved the next job's STFT does run beside the previous job's tail, but
  // it takes 0.578 instead of 0.477 ms (224 CUs, and the tail kernels -- unconfined -- still take slots on them), the tail
  // 0.40 instead of 0.13 ms, and a job 0.595 instead of 0.606 ms: 2 %, for a reported STFT time 20 % worse.  Not worth
  // being the default; kept for measurements.
  int cus = 0, reserve = 0;
  if (const char *e = getenv("NEEDLE_HIP_STFT_RESERVE_CUS")) reserve = atoi(e);
  // Default (round 4): a plain second stream, no CU mask, lowest priority -- the next job's first pass shares every CU
  // with the previous job's tail kernels, which are dispatched into the holes its retiring workgroups leave (they are all
  // built to fit one: the f64 recomputation in its 168-VGPR form).  A job then takes its first pass (stretched by what the
  // tail takes beside it: 0.46 -> 0.52 ms at 28 x 24 min) instead of first pass + tail (0.57): 0.54 ms.
  // NEEDLE_HIP_STFT_SHARE=0: one stream, as before; =1: the first pass waits for the other pipe's recomputation (the
  // ordering needed while that kernel took two holes).
  const char *share_env = getenv("NEEDLE_HIP_STFT_SHARE");
  if ((share_env ? atoi(share_env) != 0 : true) && reserve == 0) {
    int least = 0, greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
    if (hipStreamCreateWithPriority(&s, hipStreamNonBlocking, least) != hipSuccess) {
      (void)hipGetLastError();
      s = nullptr;
    }
    streams[dev] = s;
    return s;
  }
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  if (reserve > 0 && cus >= 4 * reserve) {
    // Bit i of the mask is CU i / 8 of XCD i % 8 on this part (tools/cu_mask_probe.hip: a one-bit mask confines one
    // XCD to one CU -- and an XCD whose bits are ALL clear is not confined at all, so "every 8th bit" reserves
    // nothing).  The low cus - reserve bits therefore keep reserve / 8 CUs of every XCD out of the STFT's reach.
    std::vector<uint32_t> mask((size_t)(cus + 31) / 32, 0);
    for (int c = 0; c < cus - reserve; c++) mask[(size_t)c / 32] |= 1u << (c % 32);
    if (hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()) != hipSuccess) {
      (void)hipGetLastError();
      s = nullptr;
    }
  }
  streams[dev] = s;
  return s;
}

// Host -> device PCM copies of the streaming analyzer go here, so that the fingerprint kernels of the streams that
// have landed (library stream, behind an event) run underneath the copies of the streams that follow.
hipStream_t upload_stream() {
  static std::map<int, hipStream_t> streams;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lock(g_mu);
  auto it = streams.find(dev);
  if (it != streams.end()) return it->second;
  hipStream_t s = nullptr;
  if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) s = nullptr;
  streams[dev] = s;
  return s;
}

KernelTimer::KernelTimer(const char *n, hipStream_t on) : name(n), stream(on ? on : library_stream()) {
  hipStream_t s = stream;
  std::lock_guard<std::mutex> lock(g_mu);
  active = timing_selection().on(name);
  if (!active) return;
  TimerEvents &t = g_timers[timer_key(name)];
  if (timing_selection().sum && t.hist_n < 4096) {
    if (t.hist_n == t.hist_start.size()) {
      hipEvent_t a = nullptr, b = nullptr;
      (void)hipEventCreate(&a);
      (void)hipEventCreate(&b);
      t.hist_start.push_back(a);
      t.hist_stop.push_back(b);
    }
    hist = (int)t.hist_n++;
    (void)hipEventRecord(t.hist_start[(size_t)hist], s);
    return;
  }
  t.cur ^= 1;
  if (!t.own_start[t.cur]) {
    (void)hipEventCreate(&t.own_start[t.cur]);
    (void)hipEventCreate(&t.own_stop[t.cur]);
  }
  // (a slot may hold events bound by bind_kernel_events: a pipelined job's synchronisation points, never recorded from here)
  t.start[t.cur] = t.own_start[t.cur];
  t.stop[t.cur] = t.own_stop[t.cur];
  t.recorded[t.cur] = false;
  (void)hipEventRecord(t.start[t.cur], s);
}

KernelTimer::~KernelTimer() {
  if (!active) return;
  hipStream_t s = stream;
  std::lock_guard<std::mutex> lock(g_mu);
  TimerEvents &t = g_timers[timer_key(name)];
  if (hist >= 0) {
    (void)hipEventRecord(t.hist_stop[(size_t)hist], s);
    return;
  }
  (void)hipEventRecord(t.stop[t.cur], s);
  t.recorded[t.cur] = true;
}

bool kernel_timing_on(const char *name) {
  std::lock_guard<std::mutex> lock(g_mu);
  return timing_selection().on(name);
}

void bind_kernel_events(const char *name, hipEvent_t start, hipEvent_t stop) {
  std::lock_guard<std::mutex> lock(g_mu);
  TimerEvents &t = g_timers[timer_key(name)];
  t.cur ^= 1;
  t.start[t.cur] = start;
  t.stop[t.cur] = stop;
  t.recorded[t.cur] = true;
}

double kernel_ms(const std::string &name) {
  std::lock_guard<std::mutex> lock(g_mu);
  auto it = g_timers.find(timer_key(name.c_str()));
  if (it == g_timers.end()) return -1.0;
  TimerEvents &t = it->second;
  if (timing_selection().sum) {  // every launch since the selection was set (waits for them)
    if (t.hist_n == 0) return -1.0;
    double total = 0.0;
    for (size_t i = 0; i < t.hist_n; i++) {
      float ms = 0.f;
      if (hipEventSynchronize(t.hist_stop[i]) != hipSuccess || hipEventElapsedTime(&ms, t.hist_start[i], t.hist_stop[i]) != hipSuccess) {
        (void)hipGetLastError();
        return -1.0;
      }
      total += (double)ms;
    }
    return total;
  }
  // newest launch if it has finished, otherwise the one before it (never blocks behind queued work unless
  // nothing has completed yet)
  int idx = t.cur;
  if (!t.recorded[idx] || hipEventQuery(t.stop[idx]) != hipSuccess) {
    (void)hipGetLastError();
    if (t.recorded[idx ^ 1]) idx ^= 1;
  }
  if (!t.recorded[idx]) return -1.0;
  if (hipEventSynchronize(t.stop[idx]) != hipSuccess) return -1.0;
  float ms = 0.f;
  if (hipEventElapsedTime(&ms, t.start[idx], t.stop[idx]) != hipSuccess) return -1.0;
  return (double)ms;
}

namespace {

// ---- PCM upload from a reader through a ring of pinned slabs ------------------------------------------------------
// What the file analyzer uses: reader threads fill slabs (pread straight into pinned memory), this thread issues
// one H2D copy per slab in segment order and hands slabs back when their copy has executed.  Host memory is the ring
// (kSlabs x slab bytes) whatever the size of the library, nothing is allocated or freed per run, and reading,
// sample-format conversion and the PCIe copy overlap.
struct Segment {
  size_t stream;
  uint64_t first, count;  // values of the stream
  uint64_t dev_off;       // values into the device arena
};

struct SlabRing {
  static constexpr size_t kSlabs = 16;
  char *base = nullptr;
  size_t slab_bytes = 0;
  hipEvent_t done[kSlabs] = {};
  Status ensure(size_t bytes) {
    if (bytes > slab_bytes) {
      if (base) (void)hipHostFree(base);
      base = nullptr;
      slab_bytes = 0;
      NEEDLE_HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&base), bytes * kSlabs, hipHostMallocDefault));
      slab_bytes = bytes;
    }
    for (hipEvent_t &e : done)
      if (!e) NEEDLE_HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    return Status::Ok();
  }
  int16_t *slab(size_t seg) const { return reinterpret_cast<int16_t *>(base + (seg % kSlabs) * slab_bytes); }
};

SlabRing *slab_ring() {  // per device, never destroyed (HIP may be gone when static destructors run)
  static std::mutex mu;
  static std::map<int, SlabRing *> all;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lock(mu);
  SlabRing *&r = all[dev];
  if (!r) r = new SlabRing();
  return r;
}

size_t upload_slab_bytes() {
  size_t v = 8u << 20;
  if (const char *e = getenv("NEEDLE_HIP_UPLOAD_SLAB_BYTES")) v = (size_t)std::max(16ll, atoll(e));  // tests: tiny slabs
  return (v + 15) & ~(size_t)15;
}

Status upload_through_ring(const std::vector<Segment> &segs, const PcmReader &read, unsigned readers, int16_t *d_pcm,
                           hipStream_t stream, const StreamIssued &issued_cb) {
  SlabRing &ring = *slab_ring();  // guarded by gpu_mutex(), which the callers hold
  constexpr size_t K = SlabRing::kSlabs;
  const size_t N = segs.size();
  if (N == 0) return Status::Ok();
  Status s = ring.ensure(upload_slab_bytes());
  if (!s.ok()) return s;

  long long fail_at = -1;
  if (const char *e = getenv("NEEDLE_HIP_TEST_FAIL_READ_AT")) fail_at = atoll(e);
  std::mutex mu;
  std::condition_variable cv;
  std::vector<uint8_t> filled(N, 0);  // 1 = slab holds the segment, 2 = the reader failed
  size_t released = K;                // segments below this index may write their slab
  size_t next = 0;
  bool abort = false;
  Status failure;
  auto reader = [&]() {
    for (;;) {
      size_t seg;
      {
        std::unique_lock<std::mutex> lock(mu);
        seg = next++;
        if (seg >= N) return;
        cv.wait(lock, [&] { return abort || seg < released; });
        if (abort) return;
      }
      Status rs;
      try {  // an exception must not leave a reader thread: the C ABI reports errors as codes
        if ((long long)seg == fail_at)  // tests: a read error in the middle of a transfer
          rs = Status::Make(NeedleError_IOError, "IO error: injected read failure");
        else
          rs = read(segs[seg].stream, segs[seg].first, segs[seg].count, ring.slab(seg));
      } catch (const std::exception &e) {
        rs = Status::Make(NeedleError_Unknown, std::string("PCM reader failed: ") + e.what());
      } catch (...) {
        rs = Status::Make(NeedleError_Unknown, "PCM reader failed");
      }
      {
        std::lock_guard<std::mutex> lock(mu);
        filled[seg] = rs.ok() ? 1 : 2;
        if (!rs.ok() && failure.ok()) failure = rs;
      }
      cv.notify_all();
    }
  };
  readers = (unsigned)std::max<size_t>(1, std::min<size_t>({readers, N, K}));
  std::vector<std::thread> pool;
  for (unsigned t = 0; t < readers; t++) pool.emplace_back(reader);

  size_t completed = 0;  // copies known to have executed, in segment order
  auto retire = [&](size_t upto, bool wait) {  // hands slabs of finished copies back to the readers
    bool moved = false;
    while (completed < upto) {
      hipEvent_t e = ring.done[completed % K];
      if (wait) {
        if (hipEventSynchronize(e) != hipSuccess) break;
      } else if (hipEventQuery(e) != hipSuccess) {
        break;
      }
      completed++;
      moved = true;
    }
    if (moved) {
      {
        std::lock_guard<std::mutex> lock(mu);
        released = completed + K;
      }
      cv.notify_all();
    }
  };
  Status result;
  size_t issued = 0;
  for (; issued < N; issued++) {
    retire(issued, false);
    if (issued - completed >= K / 2) retire(issued - K / 2 + 1, true);  // keep half the ring for the readers
    uint8_t state;
    {
      std::unique_lock<std::mutex> lock(mu);
      cv.wait(lock, [&] { return filled[issued] != 0; });
      state = filled[issued];
    }
    if (state != 1) {
      result = failure;
      break;
    }
    const Segment &g = segs[issued];
    hipError_t e = hipMemcpyAsync(d_pcm + g.dev_off, ring.slab(issued), g.count * sizeof(int16_t),
                                  hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) e = hipEventRecord(ring.done[issued % K], stream);
    if (e != hipSuccess) {
      result = Status::Make(NeedleError_Unknown, std::string("PCM upload failed: ") + hipGetErrorString(e));
      break;
    }
    // the last segment of a stream is on its way: the caller may queue work behind it (an event on `stream`)
    if (issued_cb && (issued + 1 == N || segs[issued + 1].stream != g.stream)) {
      result = issued_cb(g.stream);
      if (!result.ok()) {
        issued++;
        break;
      }
    }
  }
  {
    std::lock_guard<std::mutex> lock(mu);
    abort = !result.ok();
    if (abort) next = N;
  }
  cv.notify_all();
  // the ring is reused by the next call: every copy out of it has to be over before this one returns
  const hipError_t e = hipStreamSynchronize(stream);
  {
    std::lock_guard<std::mutex> lock(mu);
    released = N + K;
  }
  cv.notify_all();
  for (std::thread &t : pool) t.join();
  if (result.ok() && e != hipSuccess)
    result = Status::Make(NeedleError_Unknown, std::string("PCM upload failed: ") + hipGetErrorString(e));
  return result;
}


}  // namespace

Status gpu_upload_pcm_streamed(const std::vector<size_t> &num_values, const std::vector<uint64_t> &dev_off,
                               const PcmReader &read, unsigned readers, int16_t *d_pcm, hipStream_t stream,
                               const StreamIssued &issued_cb) {
  std::lock_guard<std::recursive_mutex> gpu_lock(gpu_mutex());
  Status s = ensure_device();
  if (!s.ok()) return s;
  const uint64_t slab_values = upload_slab_bytes() / sizeof(int16_t);
  std::vector<Segment> segs;
  for (size_t i = 0; i < num_values.size(); i++)
    for (uint64_t at = 0; at < num_values[i]; at += slab_values)
      segs.push_back(Segment{i, at, std::min<uint64_t>(slab_values, num_values[i] - at), dev_off[i] + at});
  return upload_through_ring(segs, read, readers, d_pcm, stream ? stream : library_stream(), issued_cb);
}

// PCM that already sits in (pageable) host memory: small uploads go as plain asynchronous copies, large ones
// through the ring with a few threads doing the copy into pinned memory -- the runtime's own staging of a
// pageable source reaches about half the PCIe rate.
Status gpu_upload_pcm(const std::vector<const int16_t *> &pcm, const std::vector<size_t> &num_values,
                      const std::vector<uint64_t> &dev_off, int16_t *d_pcm, hipStream_t stream,
                      const StreamIssued &issued_cb) {
  std::lock_guard<std::recursive_mutex> gpu_lock(gpu_mutex());
  Status s = ensure_device();
  if (!s.ok()) return s;
  if (!stream) stream = library_stream();
  uint64_t total = 0;
  for (size_t v : num_values) total += v;
  size_t threshold = 16u << 20;
  if (const char *e = getenv("NEEDLE_HIP_RING_UPLOAD_MIN_BYTES")) threshold = (size_t)std::max(0ll, atoll(e));
  // Caller-owned PINNED memory (hipHostMalloc / hipHostRegister, e.g. needle_hip_host_alloc): the copy engine reads
  // it in place, no staging.
  bool pinned = total > 0 && getenv("NEEDLE_HIP_NO_DIRECT_UPLOAD") == nullptr;
  for (size_t i = 0; pinned && i < pcm.size(); i++) {
    if (!num_values[i]) continue;
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, pcm[i]) != hipSuccess) {
      (void)hipGetLastError();
      pinned = false;
    } else if (attr.type != hipMemoryTypeHost) {
      pinned = false;
    }
  }
  if (pinned || total * sizeof(int16_t) < threshold) {
    for (size_t i = 0; i < pcm.size(); i++) {
      if (num_values[i])
        NEEDLE_HIP_TRY(hipMemcpyAsync(d_pcm + dev_off[i], pcm[i], num_values[i] * sizeof(int16_t),
                                      hipMemcpyHostToDevice, stream));
      if (issued_cb && !(s = issued_cb(i)).ok()) return s;
    }
    return Status::Ok();
  }
  const PcmReader read = [&](size_t stream_index, uint64_t first, uint64_t count, int16_t *dst) {
    std::memcpy(dst, pcm[stream_index] + first, count * sizeof(int16_t));
    return Status::Ok();
  };
  unsigned threads = std::min(host_threads(), 16u);
  if (const char *e = getenv("NEEDLE_HIP_UPLOAD_THREADS")) threads = (unsigned)std::max(1, atoi(e));
  return gpu_upload_pcm_streamed(num_values, dev_off, read, threads, d_pcm, stream, issued_cb);
}

}  // namespace needle
