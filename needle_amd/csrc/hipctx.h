// HIP plumbing shared by the kernel translation units: error mapping, the library stream, a tiny
// device-buffer RAII type, and per-kernel HIP-event timers (what needle_hip_last_kernel_ms reports).
#pragma once

#include <hip/hip_runtime.h>

#include <cstring>
#include <vector>

#include <map>
#include <mutex>
#include <string>

#include "common.h"

namespace needle {

#define NEEDLE_HIP_TRY(expr)                                                                         \
  do {                                                                                               \
    hipError_t _e = (expr);                                                                          \
    if (_e != hipSuccess)                                                                            \
      return Status::Make(NeedleError_Unknown, std::string("HIP error: ") + hipGetErrorString(_e) +  \
                                                   " at " #expr);                                    \
  } while (0)

// The device workspaces (chroma/feature buffers, descriptor staging) are shared per device, so GPU entry points
// are serialised; host threads (e.g. a rayon pool calling chromaprint_finish) may call in concurrently.
std::recursive_mutex &gpu_mutex();

// Fails loudly (NeedleError_Unknown "no HIP device") when no GPU is usable: there is no CPU path.
Status ensure_device();
hipStream_t library_stream();
hipStream_t download_stream();  // result downloads, ordered behind the library stream with events
hipStream_t upload_stream();    // PCM uploads of the streaming analyzer; kernels follow on the library stream behind events
hipStream_t stft_stream();    // CU-masked stream for a pipelined job's f32 STFT, or nullptr (hipctx.hip)

template <typename T>
struct DeviceBuffer {
  T *ptr = nullptr;
  size_t count = 0;
  DeviceBuffer() = default;
  DeviceBuffer(const DeviceBuffer &) = delete;
  DeviceBuffer &operator=(const DeviceBuffer &) = delete;
  ~DeviceBuffer() { release(); }
  void release() {
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    count = 0;
  }
  // Grows (never shrinks) to hold n elements; contents are not preserved.
  Status reserve(size_t n) {
    if (n <= count) return Status::Ok();
    release();
    if (n == 0) return Status::Ok();
    NEEDLE_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&ptr), n * sizeof(T)));
    count = n;
    return Status::Ok();
  }
};

// Pinned host staging for small descriptor uploads: acquire() waits until the previous async copy out
// of the buffer has executed, mark() records that point on the stream.
struct PinnedStage {
  void *ptr = nullptr;
  size_t cap = 0;
  hipEvent_t done = nullptr;
  bool pending = false;
  Status acquire(size_t bytes) {
    if (pending) {
      NEEDLE_HIP_TRY(hipEventSynchronize(done));
      pending = false;
    }
    if (bytes > cap) {
      if (ptr) (void)hipHostFree(ptr);
      ptr = nullptr;
      cap = 0;
      size_t want = bytes < 4096 ? 4096 : bytes * 2;
      NEEDLE_HIP_TRY(hipHostMalloc(&ptr, want, hipHostMallocDefault));
      cap = want;
    }
    if (!done) NEEDLE_HIP_TRY(hipEventCreateWithFlags(&done, hipEventDisableTiming));
    return Status::Ok();
  }
  void mark(hipStream_t stream) {
    if (hipEventRecord(done, stream) == hipSuccess) pending = true;
  }
};


// Descriptor arrays (a few KB of offsets and lengths per launch) rarely change between launches of a job that
// is run again and again: the upload through pinned staging -- one more dispatch on the stream and a host wait
// for the staging buffer -- is skipped when the device copy already holds exactly these bytes.
template <class T>
struct DescriptorUpload {
  std::vector<T> resident;     // what dst holds (or will hold, by stream order)
  const T *resident_at = nullptr;
  Status put(DeviceBuffer<T> *dst, PinnedStage *stage, const std::vector<T> &items, hipStream_t stream,
             bool *uploaded = nullptr) {
    if (uploaded) *uploaded = false;
    Status s = dst->reserve(items.size());
    if (!s.ok()) return s;
    if (dst->ptr == resident_at && resident.size() == items.size() &&
        std::memcmp(resident.data(), items.data(), items.size() * sizeof(T)) == 0)
      return Status::Ok();
    if (uploaded) *uploaded = true;
    if (!(s = stage->acquire(items.size() * sizeof(T))).ok()) return s;
    std::memcpy(stage->ptr, items.data(), items.size() * sizeof(T));
    NEEDLE_HIP_TRY(hipMemcpyAsync(dst->ptr, stage->ptr, items.size() * sizeof(T), hipMemcpyHostToDevice, stream));
    stage->mark(stream);
    resident = items;
    resident_at = dst->ptr;
    return Status::Ok();
  }
};

// Records start/stop events around a kernel launch on the library stream, for the kernels selected with
// set_kernel_timing; elapsed time is read
// lazily (after a sync) by needle_hip_last_kernel_ms.
struct KernelTimer {
  explicit KernelTimer(const char *name, hipStream_t stream = nullptr);  // nullptr: the library stream
  ~KernelTimer();
  const char *name;
  hipStream_t stream;
  bool active = false;
  int hist = -1;  // "sum" mode: this launch's own event pair
};
double kernel_ms(const std::string &name);
// A launch that carries its own events (hipExtLaunchKernelGGL: the dispatch packet's completion signal IS the stop
// event -- no marker packets around the kernel): is `name` being timed, and if so, these are its newest start / stop events
// (not owned; they must outlive the next two launches of that name).
bool kernel_timing_on(const char *name);
void bind_kernel_events(const char *name, hipEvent_t start, hipEvent_t stop);
// "all", a comma-separated list of kernel names, or NULL / "" / "none" (the default, unless the environment
// variable NEEDLE_HIP_KERNEL_TIMING says otherwise)
void set_kernel_timing(const char *kernels);

}  // namespace needle
