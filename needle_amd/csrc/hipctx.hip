#include "hipctx.h"

#include <cstdio>
#include <mutex>

namespace needle {

namespace {
thread_local std::string g_last_error;
std::mutex g_mu;
std::map<int, hipStream_t> g_streams;

struct TimerEvents {
  hipEvent_t start = nullptr, stop = nullptr;
  bool recorded = false;
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
  if (!t.start) {
    (void)hipEventCreate(&t.start);
    (void)hipEventCreate(&t.stop);
  }
  (void)hipEventRecord(t.start, s);
}

KernelTimer::~KernelTimer() {
  hipStream_t s = library_stream();
  std::lock_guard<std::mutex> lock(g_mu);
  TimerEvents &t = g_timers[timer_key(name)];
  (void)hipEventRecord(t.stop, s);
  t.recorded = true;
}

double kernel_ms(const std::string &name) {
  std::lock_guard<std::mutex> lock(g_mu);
  auto it = g_timers.find(timer_key(name.c_str()));
  if (it == g_timers.end() || !it->second.recorded) return -1.0;
  if (hipEventSynchronize(it->second.stop) != hipSuccess) return -1.0;
  float ms = 0.f;
  if (hipEventElapsedTime(&ms, it->second.start, it->second.stop) != hipSuccess) return -1.0;
  return (double)ms;
}

}  // namespace needle
