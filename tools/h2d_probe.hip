// Host -> device copy rate from pinned memory as a function of the pinned footprint (why does streaming 59.5 GB of PCM
// reach 31.6 GB/s when a 0.44 GB job reaches 54?).  For each footprint: N pinned buffers of 29.8 MB (a 45-minute
// episode's opening window), touched by the CPU, copied one after another on one stream into a 2 GiB device arena
// (wrapping), timed with events; the same with hipHostMallocNumaUser / default flags, and with the buffers touched by a
// device -> host copy first (what tools/library_stream_device.py does).
//   hipcc -O2 --offload-arch=gfx950 tools/h2d_probe.hip -o tools/h2d_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

__global__ void spin(float *x, int iters) {
  float v = x[threadIdx.x];
  for (int i = 0; i < iters; i++) v = v * 1.0001f + 0.5f;
  x[blockIdx.x * blockDim.x + threadIdx.x] = v;
}

int main() {
  const size_t piece = 29767500ull;  // bytes
  char *d = nullptr;
  const size_t arena = 2ull << 30;
  CK(hipMalloc(&d, arena));
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  for (int mode = 0; mode < 3; mode++) {
    for (size_t count : {16ul, 64ul, 250ul, 500ul}) {
      std::vector<char *> h(count, nullptr);
      for (size_t i = 0; i < count; i++) {
        CK(hipHostMalloc((void **)&h[i], piece, mode == 1 ? hipHostMallocNumaUser : hipHostMallocDefault));
        if (mode == 2) CK(hipMemcpy(h[i], d, piece, hipMemcpyDeviceToHost));  // first touch by the device
        else std::memset(h[i], 1, piece);
      }
      for (int rep = 0; rep < 2; rep++) {
        CK(hipEventRecord(a, s));
        size_t off = 0;
        for (size_t i = 0; i < count; i++) {
          if (off + piece > arena) off = 0;
          CK(hipMemcpyAsync(d + off, h[i], piece, hipMemcpyHostToDevice, s));
          off += (piece + 255) & ~(size_t)255;
        }
        CK(hipEventRecord(b, s));
        CK(hipEventSynchronize(b));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, a, b));
        if (rep == 1)
          std::printf("%s, %3zu buffers (%.1f GB pinned): %.1f GB/s\n",
                      mode == 0 ? "default flags, CPU-touched" : mode == 1 ? "NumaUser, CPU-touched" : "default flags, device-touched",
                      count, count * piece / 1e9, count * piece / (ms * 1e6));
      }
      for (size_t i = 0; i < count; i++) CK(hipHostFree(h[i]));
    }
  }
  // what tools/library_stream_device.py does between its calls: synchronous device -> host copies into the pinned buffers
  // (hipMemcpy), then the timed host -> device burst on the stream -- four rounds over two sets of buffers
  {
    const size_t count = 250;
    std::vector<char *> h[2];
    for (int k = 0; k < 2; k++) {
      h[k].assign(count, nullptr);
      for (size_t i = 0; i < count; i++) CK(hipHostMalloc((void **)&h[k][i], piece, hipHostMallocDefault));
    }
    char *gen = nullptr;
    CK(hipMalloc(&gen, count * piece));
    for (int round = 0; round < 4; round++) {
      std::vector<char *> &set = h[round & 1];
      for (size_t i = 0; i < count; i++) CK(hipMemcpy(set[i], gen + i * piece, piece, hipMemcpyDeviceToHost));
      CK(hipEventRecord(a, s));
      size_t off = 0;
      for (size_t i = 0; i < count; i++) {
        if (off + piece > arena) off = 0;
        CK(hipMemcpyAsync(d + off, set[i], piece, hipMemcpyHostToDevice, s));
        off += (piece + 255) & ~(size_t)255;
      }
      CK(hipEventRecord(b, s));
      CK(hipEventSynchronize(b));
      float ms = 0;
      CK(hipEventElapsedTime(&ms, a, b));
      std::printf("round %d (D2H hipMemcpy into set %d, then H2D burst): %.1f GB/s\n", round, round & 1, count * piece / (ms * 1e6));
    }
  }
  // the same burst while short kernels run on another stream (what the fingerprinter does under the copies): does compute
  // activity slow the copy engine down?
  {
    const size_t count = 250;
    std::vector<char *> h(count, nullptr);
    for (size_t i = 0; i < count; i++) CK(hipHostMalloc((void **)&h[i], piece, hipHostMallocDefault));
    float *x = nullptr;
    CK(hipMalloc(&x, 4096 * 256 * sizeof(float)));
    hipStream_t k;
    CK(hipStreamCreateWithFlags(&k, hipStreamNonBlocking));
    for (int pass = 0; pass < 4; pass++) {
    const bool heavy = pass >= 1;  // pass 1..: the kernels keep the whole chip busy for the whole burst
    const int mode = pass == 0 ? 0 : pass == 3 ? 0 : pass;  // last pass: copies alone again (is the slowdown persistent?)
    {  // 0: copies alone, 1: + a 60 us kernel every copy, 2: + a kernel behind an event of every copy
      CK(hipEventRecord(a, s));
      size_t off = 0;
      hipEvent_t landed;
      CK(hipEventCreateWithFlags(&landed, hipEventDisableTiming));
      std::vector<float> each(count);
      std::vector<hipEvent_t> ev(count + 1);
      for (auto &e : ev) CK(hipEventCreate(&e));
      CK(hipEventRecord(ev[0], s));
      for (size_t i = 0; i < count; i++) {
        if (off + piece > arena) off = 0;
        CK(hipMemcpyAsync(d + off, h[i], piece, hipMemcpyHostToDevice, s));
        CK(hipEventRecord(ev[i + 1], s));
        off += (piece + 255) & ~(size_t)255;
        if (mode == 2) {
          CK(hipEventRecord(landed, s));
          CK(hipStreamWaitEvent(k, landed, 0));
        }
        if (mode >= 1) hipLaunchKernelGGL(spin, dim3(4096), dim3(256), 0, k, x, heavy ? 200000 : 2000);
      }
      CK(hipEventRecord(b, s));
      CK(hipEventSynchronize(b));
      CK(hipStreamSynchronize(k));
      float ms = 0;
      CK(hipEventElapsedTime(&ms, a, b));
      float first = 0, last = 0;
      for (size_t i = 0; i < 50; i++) { float t; CK(hipEventElapsedTime(&t, ev[i], ev[i + 1])); first += t; }
      for (size_t i = count - 50; i < count; i++) { float t; CK(hipEventElapsedTime(&t, ev[i], ev[i + 1])); last += t; }
      std::printf("%s: %.1f GB/s (first 50 copies %.0f us each, last 50 %.0f us each)\n",
                  mode == 0 ? "copies alone" : mode == 1 ? "copies + kernels on another stream" : "copies + kernels behind an event of each copy",
                  count * piece / (ms * 1e6), first / 50 * 1e3, last / 50 * 1e3);
    }
    }
  }
  // does allocation churn on the device change the copy rate?  burst, [hipMalloc 3 GB, touch, hipFree], burst, ...
  {
    const size_t count = 100;
    std::vector<char *> h(count, nullptr);
    for (size_t i = 0; i < count; i++) CK(hipHostMalloc((void **)&h[i], piece, hipHostMallocDefault));
    for (int round = 0; round < 4; round++) {
      if (round > 0) {
        std::vector<float *> g(round == 2 ? 1 : 100, nullptr);   // round 2: one 3 GB block; others: 100 blocks of 30 MB
        for (auto &p : g) CK(hipMalloc(&p, g.size() == 1 ? 100 * piece : piece));
        for (auto &p : g) hipLaunchKernelGGL(spin, dim3(4096), dim3(256), 0, 0, p, 10);
        CK(hipDeviceSynchronize());
        for (auto &p : g) CK(hipFree(p));
      }
      CK(hipEventRecord(a, s));
      size_t off = 0;
      for (size_t i = 0; i < count; i++) {
        if (off + piece > arena) off = 0;
        CK(hipMemcpyAsync(d + off, h[i], piece, hipMemcpyHostToDevice, s));
        off += (piece + 255) & ~(size_t)255;
      }
      CK(hipEventRecord(b, s));
      CK(hipEventSynchronize(b));
      float ms = 0;
      CK(hipEventElapsedTime(&ms, a, b));
      std::printf("churn round %d: H2D burst %.1f GB/s\n", round, count * piece / (ms * 1e6));
    }
  }
  return 0;
}