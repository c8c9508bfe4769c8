#include "hipctx.h"

#include <cstdio>
#include <mutex>

namespace needle {

namespace {
thread_local std::string g_last_error;
std::mutex g_mu;
std::map<int, hipStream_t> g_streams;

struct TimerEvents {  // two alternating event pairs, so the newest COMPLETED launch can be read without blocking
  hipEvent_t start[2] = {nullptr, nullptr}, stop[2] = {nullptr, nullptr};
  bool recorded[2] = {false, false};
  int cur = 1;
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
  if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) s = nullptr;
  g_streams[dev] = s;
  return s;
}

KernelTimer::KernelTimer(const char *n) : name(n) {
  hipStream_t s = library_stream();
  std::lock_guard<std::mutex> lock(g_mu);
  TimerEvents &t = g_timers[timer_key(name)];
  t.cur ^= 1;
  if (!t.start[t.cur]) {
    (void)hipEventCreate(&t.start[t.cur]);
    (void)hipEventCreate(&t.stop[t.cur]);
  }
  t.recorded[t.cur] = false;
  (void)hipEventRecord(t.start[t.cur], s);
}

KernelTimer::~KernelTimer() {
  hipStream_t s = library_stream();
  std::lock_guard<std::mutex> lock(g_mu);
  TimerEvents &t = g_timers[timer_key(name)];
  (void)hipEventRecord(t.stop[t.cur], s);
  t.recorded[t.cur] = true;
}

double kernel_ms(const std::string &name) {
  std::lock_guard<std::mutex> lock(g_mu);
  auto it = g_timers.find(timer_key(name.c_str()));
  if (it == g_timers.end()) return -1.0;
  TimerEvents &t = it->second;
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

}  // namespace needle
