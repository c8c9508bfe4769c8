// Probe (not part of the product): does gfx950 convert s16 -> f32 inside a typed buffer load?  tbuffer_load_format_x
// with [BUF_DATA_FORMAT_16, BUF_NUM_FORMAT_SSCALED] should return (float)s16 exactly, which would take the 32
// v_cvt_f32_i32 per thread and frame pair out of stft_chroma32_kernel.  Checks every s16 value, then times 32 loads per
// thread against global_load_sshort + v_cvt_f32_i32 at 3 workgroups per CU.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/format_load_probe.hip -o tools/format_load_probe
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));

__device__ __forceinline__ v4i make_rsrc(const void *base, uint32_t bytes) {
  const uint64_t a = (uint64_t)(uintptr_t)base;
  v4i r;
  r.x = (int)(uint32_t)a;
  r.y = (int)(uint32_t)(a >> 32) & 0xffff;  // stride 0
  r.z = (int)bytes;                         // num_records in bytes (stride 0)
  r.w = 4 | (5 << 3) | (6 << 6) | (7 << 9) | (3 << 12) | (2 << 15);  // dst_sel xyzw, SSCALED, 16
  return r;
}

__global__ void check_kernel(const int16_t *in, float *out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const v4i r = make_rsrc(in, (uint32_t)n * 2u);
  float v;
  const int off = i * 2;
  asm volatile("tbuffer_load_format_x %0, %1, %2, 0 format:[BUF_DATA_FORMAT_16,BUF_NUM_FORMAT_SSCALED] offen\n s_waitcnt vmcnt(0)"
               : "=v"(v) : "v"(off), "s"(r) : "memory");
  out[i] = v;
}

template <int MODE>
__global__ __launch_bounds__(256) void time_kernel(const int16_t *in, float *out, unsigned long long *cycles, int iters, uint32_t bytes) {
  extern __shared__ float lds[];
  const int t = threadIdx.x;
  const v4i r = make_rsrc(in, bytes);
  float acc = 0.0f;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
    const uint32_t frame = ((uint32_t)(blockIdx.x * 37 + it) * 2730u) % (bytes - 16384u);  // an even byte offset inside the buffer
    float f[16];
    if (MODE == 0) {
      const int off = (int)frame + 2 * t;
      const int hi = 4096;
#pragma unroll
      for (int k = 0; k < 8; k++)
        asm volatile("tbuffer_load_format_x %0, %1, %2, 0 format:[BUF_DATA_FORMAT_16,BUF_NUM_FORMAT_SSCALED] offen offset:%3"
                     : "=v"(f[k]) : "v"(off), "s"(r), "n"(512 * k) : "memory");
#pragma unroll
      for (int k = 0; k < 8; k++)
        asm volatile("tbuffer_load_format_x %0, %1, %2, %4 format:[BUF_DATA_FORMAT_16,BUF_NUM_FORMAT_SSCALED] offen offset:%3"
                     : "=v"(f[8 + k]) : "v"(off), "s"(r), "n"(512 * k), "s"(hi) : "memory");
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]),
                   "+v"(f[8]), "+v"(f[9]), "+v"(f[10]), "+v"(f[11]), "+v"(f[12]), "+v"(f[13]), "+v"(f[14]), "+v"(f[15]));
    } else {
      const int16_t *q = reinterpret_cast<const int16_t *>(reinterpret_cast<const char *>(in) + frame) + t;
#pragma unroll
      for (int k = 0; k < 16; k++) f[k] = (float)(int)q[256 * k];
    }
#pragma unroll
    for (int k = 0; k < 16; k++) acc += f[k];
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  if (t == 0) cycles[blockIdx.x] = c1 - c0;
  out[blockIdx.x * 256 + t] = acc;
}

int main() {
  const int n = 65536 + 8192;
  std::vector<int16_t> h(n);
  for (int i = 0; i < n; i++) h[i] = (int16_t)(i < 65536 ? i - 32768 : (i * 2654435761u) >> 16);
  int16_t *d_in;
  float *d_out;
  unsigned long long *d_cycles;
  if (hipMalloc(&d_in, n * 2) != hipSuccess || hipMalloc(&d_out, 1024 * 256 * 4) != hipSuccess || hipMalloc(&d_cycles, 8192) != hipSuccess) return 2;
  (void)hipMemcpy(d_in, h.data(), n * 2, hipMemcpyHostToDevice);
  check_kernel<<<(n + 255) / 256, 256>>>(d_in, d_out, n);
  std::vector<float> got(n);
  if (hipMemcpy(got.data(), d_out, n * 4, hipMemcpyDeviceToHost) != hipSuccess) {
    std::printf("format load FAILED to run: %s\n", hipGetErrorString(hipGetLastError()));
    return 1;
  }
  long bad = 0;
  for (int i = 0; i < n; i++) bad += got[i] != (float)h[i];
  std::printf("tbuffer_load_format_x [16, SSCALED]: %ld of %d values differ from (float)s16 (first: in %d -> %g)\n", bad, n, (int)h[0], got[0]);
  for (int mode = 0; mode < 2; mode++) {
    for (int wgs : {1, 3}) {
      std::vector<unsigned long long> c(256 * wgs);
      double best = 1e30;
      for (int rep = 0; rep < 3; rep++) {
        if (mode == 0) time_kernel<0><<<256 * wgs, 256, 48 * 1024>>>(d_in, d_out, d_cycles, 2000, n * 2);
        else time_kernel<1><<<256 * wgs, 256, 48 * 1024>>>(d_in, d_out, d_cycles, 2000, n * 2);
        (void)hipMemcpy(c.data(), d_cycles, c.size() * 8, hipMemcpyDeviceToHost);
        double sum = 0;
        for (auto v : c) sum += (double)v;
        best = std::min(best, sum / c.size());
      }
      std::printf("%-46s %d workgroup(s)/CU: %.1f cycles per 16 loads (+ use) of one wave\n",
                  mode == 0 ? "tbuffer_load_format_x (converted in the load)" : "global_load_sshort + v_cvt_f32_i32", wgs, best / 2000.0);
    }
  }
  return bad ? 1 : 0;
}
