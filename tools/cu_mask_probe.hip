// Probe (not product): does hipExtStreamCreateWithCUMask partition the CUs on this stack?  Two streams with
// complementary masks (A: all but the last `small` CUs, B: the last `small` CUs) each run a compute-bound kernel sized
// for the whole chip; times alone and together.  If masks work: together ~= max(alone_A_on_its_share, alone_B_on_its).
//   hipcc -O3 --offload-arch=gfx950 tools/cu_mask_probe.hip -o tools/cu_mask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void spin(float *out, int iters) {
  float a = threadIdx.x * 1e-3f, b = 1.0001f;
  for (int i = 0; i < iters; i++) a = a * b + 1e-6f;
  if (a == 123.f) out[0] = a;
}
__global__ void where(unsigned *cu_hist) {  // which CU did this workgroup land on (XCC_ID + CU id from HW_ID)
  if (threadIdx.x == 0) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const unsigned cu = (hw >> 8) & 0xf, se = (hw >> 13) & 0x7, sh = (hw >> 12) & 1;
    atomicAdd(&cu_hist[((xcc & 0xf) * 8 + se) * 32 + sh * 16 + cu], 1u);
  }
}

int main() {
  int cus = 0;
  CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  const int small = 32;
  const int words = (cus + 31) / 32;
  std::vector<uint32_t> ma(words, 0), mb(words, 0);
  for (int c = 0; c < cus; c++) (c < cus - small ? ma : mb)[c / 32] |= 1u << (c % 32);
  hipStream_t sa, sb, s0;
  CK(hipStreamCreate(&s0));
  hipError_t e = hipExtStreamCreateWithCUMask(&sa, words, ma.data());
  if (e != hipSuccess) { std::printf("hipExtStreamCreateWithCUMask failed: %s\n", hipGetErrorString(e)); return 0; }
  CK(hipExtStreamCreateWithCUMask(&sb, words, mb.data()));
  float *out;
  unsigned *hist;
  CK(hipMalloc(&out, 4));
  CK(hipMalloc(&hist, 4096 * 4));
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  auto timed = [&](hipStream_t s, int grid, int iters, const char *what) {
    hipLaunchKernelGGL(spin, dim3(grid), dim3(256), 0, s, out, 1000);
    hipStreamSynchronize(s);
    hipEventRecord(a, s);
    hipLaunchKernelGGL(spin, dim3(grid), dim3(256), 0, s, out, iters);
    hipEventRecord(b, s);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    std::printf("%-44s %.3f ms\n", what, ms);
    return ms;
  };
  const int grid = cus * 8, iters = 200000;
  timed(s0, grid, iters, "unmasked stream, whole chip");
  timed(sa, grid, iters, "stream A (all but 32 CUs)");
  timed(sb, grid, iters, "stream B (32 CUs)");
  for (int which = 0; which < 2; which++) {
    CK(hipMemset(hist, 0, 4096 * 4));
    hipLaunchKernelGGL(where, dim3(cus * 16), dim3(64), 0, which ? sb : sa, hist);
    CK(hipDeviceSynchronize());
    std::vector<unsigned> h(4096);
    CK(hipMemcpy(h.data(), hist, 4096 * 4, hipMemcpyDeviceToHost));
    int used = 0;
    for (unsigned v : h) used += v != 0;
    std::printf("stream %c: workgroups landed on %d distinct CUs\n", which ? 'B' : 'A', used);
  }
  // the mask's bit order: where do the workgroups of a stream with ONE bit set land?
  for (int bit : {0, 1, 2, 3, 7, 8, 9, 15, 16, 31, 32, 33, 63, 64, 128, 255}) {
    if (bit >= cus) continue;
    std::vector<uint32_t> one(words, 0);
    one[bit / 32] = 1u << (bit % 32);
    hipStream_t so;
    if (hipExtStreamCreateWithCUMask(&so, words, one.data()) != hipSuccess) { std::printf("bit %d: refused\n", bit); continue; }
    CK(hipMemset(hist, 0, 4096 * 4));
    hipLaunchKernelGGL(where, dim3(64), dim3(64), 0, so, hist);
    CK(hipDeviceSynchronize());
    std::vector<unsigned> h(4096);
    CK(hipMemcpy(h.data(), hist, 4096 * 4, hipMemcpyDeviceToHost));
    std::printf("mask bit %3d ->", bit);
    for (int i = 0; i < 4096; i++)
      if (h[i]) std::printf(" xcc %d se %d sh %d cu %d (%u wgs)", i / 256, (i / 32) % 8, (i / 16) % 2, i % 16, h[i]);
    std::printf("\n");
    CK(hipStreamDestroy(so));
  }
  // together: a small kernel chain on B while A runs a chip-filling kernel
  hipEventRecord(a, s0);
  hipStreamWaitEvent(sa, a, 0);
  hipStreamWaitEvent(sb, a, 0);
  hipLaunchKernelGGL(spin, dim3(grid), dim3(256), 0, sa, out, iters);
  for (int k = 0; k < 5; k++) hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, sb, out, iters / 10);
  hipEvent_t ea, eb;
  CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
  hipEventRecord(ea, sa);
  hipEventRecord(eb, sb);
  hipEventSynchronize(ea);
  hipEventSynchronize(eb);
  float ta = 0, tb = 0;
  hipEventElapsedTime(&ta, a, ea);
  hipEventElapsedTime(&tb, a, eb);
  std::printf("together: A's big kernel done after %.3f ms, B's chain of 5 small kernels after %.3f ms\n", ta, tb);
  return 0;
}
