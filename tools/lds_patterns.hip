// Micro-benchmark: what the 128-bit LDS access patterns of stft_chroma_kernel cost on this GPU, for candidate
// slot swizzles S(i) (fp_core.h pidx).  Each thread gets its 16 slot numbers of a pattern from a table built on the
// host, then issues them as ds_read_b128 / ds_write_b128 in a loop (256 threads, 2 workgroups per CU as in the
// kernel); the figure printed is LDS-pipeline cycles per wave-instruction at the measured time, i.e. how far a
// pattern is from the conflict-free cost.  Also probes which lanes of a wave are served together (pairs of lanes made
// to collide while all others are spread).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>

typedef double v2d __attribute__((ext_vector_type(2), aligned(16)));

template <bool WRITE, int OPS>
__global__ __launch_bounds__(256, 2) void lds_kernel(const uint16_t *__restrict__ slots, int reps, double *out) {
  extern __shared__ v2d lds[];
  uint32_t s[OPS];
#pragma unroll
  for (int k = 0; k < OPS; k++) s[k] = slots[threadIdx.x * 16 + k];
  for (int i = threadIdx.x; i < 4352; i += 256) lds[i] = v2d{(double)i, 1.0};
  __syncthreads();
  v2d acc = v2d{0.0, 0.0};
  for (int r = 0; r < reps; r++) {
#pragma unroll
    for (int k = 0; k < OPS; k++) {
      if (WRITE) {
        lds[s[k]] = acc;
      } else {
        acc += lds[s[k]];
      }
    }
    if (WRITE) acc.x += 1.0;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  if (acc.x == 12345.678) out[blockIdx.x * 256 + threadIdx.x] = acc.x + acc.y;
}

static double run(bool write, int ops, const std::vector<uint16_t> &slots, int reps) {
  static uint16_t *d_slots = nullptr;
  static double *d_out = nullptr;
  if (!d_slots) {
    hipMalloc(&d_slots, 256 * 16 * sizeof(uint16_t));
    hipMalloc(&d_out, 512 * 256 * sizeof(double));
    hipFuncSetAttribute(reinterpret_cast<const void *>(lds_kernel<false, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, 70000);
    hipFuncSetAttribute(reinterpret_cast<const void *>(lds_kernel<true, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, 70000);
    hipFuncSetAttribute(reinterpret_cast<const void *>(lds_kernel<false, 6>), hipFuncAttributeMaxDynamicSharedMemorySize, 70000);
    hipFuncSetAttribute(reinterpret_cast<const void *>(lds_kernel<true, 6>), hipFuncAttributeMaxDynamicSharedMemorySize, 70000);
  }
  hipMemcpy(d_slots, slots.data(), 256 * 16 * sizeof(uint16_t), hipMemcpyHostToDevice);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  float best = 1e30f;
  for (int it = 0; it < 3; it++) {
    hipEventRecord(a);
    if (ops == 16) {
      if (write) lds_kernel<true, 16><<<512, 256, 69632>>>(d_slots, reps, d_out);
      else lds_kernel<false, 16><<<512, 256, 69632>>>(d_slots, reps, d_out);
    } else {
      if (write) lds_kernel<true, 6><<<512, 256, 69632>>>(d_slots, reps, d_out);
      else lds_kernel<false, 6><<<512, 256, 69632>>>(d_slots, reps, d_out);
    }
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    if (ms < best) best = ms;
  }
  // per CU: 2 workgroups x 4 waves x reps x ops wave-instructions share one LDS pipeline
  const double instr_per_cu = 2.0 * 4.0 * reps * ops;
  return best * 1e-3 * 2.4e9 / instr_per_cu;  // cycles (at 2.4 GHz) per wave-instruction on the CU's LDS pipeline
}

int main() {
  const int reps = 4000;
  // ---- which lanes are served together.  Y1: slot residue (mod 16) = lane mod 16: conflict-free whether a 128-bit
  // access is served by 16 CONSECUTIVE lanes at a time or by the interleaved groups {0-3,12-15,20-27} / {4-11,16-19,
  // 28-31} (+32) that resample.hip was tuned for.  Y2: lanes 16-31 (48-63) shifted by 8 residues: still conflict-free
  // for consecutive groups, every access 2-way conflicted for the interleaved ones.  Y3: residue = lane / 4 (4-way). ----
  {
    std::vector<uint16_t> p(256 * 16);
    auto fill = [&](std::function<int(int)> residue) {
      for (int t = 0; t < 256; t++)
        for (int k = 0; k < 16; k++) p[t * 16 + k] = (uint16_t)(residue(t & 63) + 16 * (((t & 63) >> 4) + 4 * k + 64 * (t >> 6)));
    };
    fill([](int l) { return l & 15; });
    std::printf("lane-group probe, cycles per wave-instruction: Y1 read %.2f write %.2f", run(false, 16, p, reps), run(true, 16, p, reps));
    fill([](int l) { return (l & 16) ? ((l + 8) & 15) : (l & 15); });
    std::printf(" | Y2 read %.2f write %.2f", run(false, 16, p, reps), run(true, 16, p, reps));
    fill([](int l) { return l >> 2; });
    std::printf(" | Y3 read %.2f write %.2f\n", run(false, 16, p, reps), run(true, 16, p, reps));
    // residue period: is a slot's bank group slot mod 16 (64 banks) or slot mod 8 (32 banks)?  Z: lanes l and l + 8 of
    // every 16 consecutive lanes 8 slots apart (conflict only if the period is 8)
    fill([](int l) { return l & 15; });
    std::printf("(if Y1 ~ Y2: consecutive 16-lane groups; if Y2 ~ 2 x Y1: interleaved groups)\n");
  }
  // ---- the kernel's patterns under candidate swizzles -------------------------------------------------------------
  struct Swz {
    const char *name;
    std::function<int(int)> f;
  };
  std::vector<Swz> swz = {
      {"i + (i>>4)            [current]", [](int i) { return i + (i >> 4); }},
      {"i                     [none]", [](int i) { return i; }},
      {"i + (i>>5)", [](int i) { return i + (i >> 5); }},
      {"i + (i>>6)", [](int i) { return i + (i >> 6); }},
      {"i ^ ((i>>4)&15)", [](int i) { return i ^ ((i >> 4) & 15); }},
      {"i ^ ((i>>8)&15)", [](int i) { return i ^ ((i >> 8) & 15); }},
      {"i ^ (((i>>4)^(i>>8))&15)", [](int i) { return i ^ (((i >> 4) ^ (i >> 8)) & 15); }},
      {"i ^ ((i>>4)&7)", [](int i) { return i ^ ((i >> 4) & 7); }},
      {"i ^ (((i>>4)&3)<<2)", [](int i) { return i ^ (((i >> 4) & 3) << 2); }},
      {"i + (i>>4) + (i>>8)", [](int i) { return i + (i >> 4) + (i >> 8); }},
      {"i + 2(i>>4)", [](int i) { return i + 2 * (i >> 4); }},
      {"i + (i>>4)*4 mod", [](int i) { return (i & 15) + 20 * (i >> 4); }},
  };
  auto slot_of_bin = [](int kf) { return 256 * (kf & 15) + 16 * ((kf >> 4) & 15) + (kf >> 8); };
  std::printf("\npattern cost, cycles per wave-instruction (read / write as the kernel uses them):\n");
  std::printf("%-34s %8s %8s %8s %8s %8s %8s\n", "swizzle", "s0 wr", "s1 rd", "s1 wr", "s2 rd", "publ wr", "partn rd");
  for (const Swz &z : swz) {
    std::vector<uint16_t> p(256 * 16);
    double c[6];
    for (int t = 0; t < 256; t++)
      for (int k = 0; k < 16; k++) p[t * 16 + k] = (uint16_t)z.f(t + 256 * k);
    c[0] = run(true, 16, p, reps);
    for (int t = 0; t < 256; t++)
      for (int k = 0; k < 16; k++) p[t * 16 + k] = (uint16_t)z.f(256 * (t >> 4) + (t & 15) + 16 * k);
    c[1] = run(false, 16, p, reps);
    c[2] = run(true, 16, p, reps);
    for (int t = 0; t < 256; t++)
      for (int k = 0; k < 16; k++) p[t * 16 + k] = (uint16_t)z.f(16 * t + k);
    c[3] = run(false, 16, p, reps);
    for (int t = 0; t < 256; t++)
      for (int k = 0; k < 6; k++) p[t * 16 + k] = (uint16_t)z.f(16 * t + 10 + k);
    c[4] = run(true, 6, p, reps);
    for (int t = 0; t < 256; t++)
      for (int k = 0; k < 6; k++) {
        const int kf = (t >> 4) + 16 * (t & 15) + 256 * k;
        p[t * 16 + k] = (uint16_t)z.f(slot_of_bin((4096 - kf) & 4095));
      }
    c[5] = run(false, 6, p, reps);
    std::printf("%-34s %8.2f %8.2f %8.2f %8.2f %8.2f %8.2f\n", z.name, c[0], c[1], c[2], c[3], c[4], c[5]);
  }
  return 0;
}
