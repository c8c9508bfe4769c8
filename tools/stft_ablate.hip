// Diagnostic (not part of the product): the loop of stft_chroma_kernel (fingerprint.hip) with pieces switched off,
// to see what each piece costs at the product's launch shape.  Outputs are wrong by construction; only times count.
//   bit 0: window from a constant instead of the table      bit 1: no PCM loads
//   bit 2: no pitch-class fold (power store, fold, 2 barriers)  bit 3: no stream lookup (pointer arithmetic only)
//   bit 4: every second workgroup (by arrival parity on its CU, approximated by blockIdx) starts late
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../needle_amd/csrc/fp_core.h"

using needle::core::cd;
namespace core = needle::core;
constexpr int kHop = 1365, kBands = 12;

struct Stream { uint64_t pcm_off; uint32_t frames, frame_base, pair_base, pad; };

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ int find_stream(const Stream *s, int n, uint32_t g) {
  int lo = 0, hi = n - 1;
  while (lo < hi) { int mid = (lo + hi + 1) >> 1; if (s[mid].pair_base <= g) lo = mid; else hi = mid - 1; }
  return lo;
}
struct PairSrc { const int16_t *a, *b; double keep_b; uint64_t row; bool has_b; };

template <int F>
__global__ __launch_bounds__(256, 2) void kernel(const int16_t *__restrict__ pcm, const Stream *__restrict__ streams,
                                                 int num_streams, const cd *__restrict__ tw,
                                                 const double *__restrict__ window,
                                                 const uint16_t *__restrict__ bin_slot,
                                                 const uint32_t *__restrict__ class_start, double *__restrict__ chroma,
                                                 uint32_t total_pairs, uint32_t pairs_per_block, int skew) {
  extern __shared__ cd lds[];
  const int t = threadIdx.x;
  const uint32_t first = blockIdx.x * pairs_per_block;
  const uint32_t last = min(total_pairs, first + pairs_per_block);
  if (first >= last) return;
  const cd base0 = tw[t], base1 = tw[16 * (t & 15)];
  if (F & 16) {  // skew by the wave slot this workgroup's waves were given on their SIMD (HW_ID[3:0])
    const int slot = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | ((4 - 1) << 11));
    if (slot & 1) for (int i = 0; i < skew; i++) __builtin_amdgcn_s_sleep(16);
  }
  int si = find_stream(streams, num_streams, first);
  Stream st = streams[si];
  uint32_t st_end = st.pair_base + (st.frames + 1) / 2;
  uint32_t slot_pk[3];
#pragma unroll
  for (int j = 0; j < 6; j++) {
    const int kf = core::dif_bin_of(t, j);
    const uint32_t idx = (kf >= core::kMinBin && kf < core::kMaxBin) ? bin_slot[kf - core::kMinBin] : core::kPowerTrashSlot;
    slot_pk[j >> 1] = (j & 1) ? (slot_pk[j >> 1] | (idx << 16)) : idx;
  }
  if (t == 0) lds[core::kPowerZeroSlot] = cd{0.0, 0.0};
  const bool folds = t < kBands * core::kClassLanes;
  const int fold_c = t >> 4, fold_l = t & 15;
  const uint32_t fold_bounds = folds ? (class_start[fold_c] | (class_start[fold_c + 1] << 16)) : 0;
  auto locate = [&](uint32_t g) {
    PairSrc p;
    if (F & 8) {
      p.has_b = true; p.a = pcm + (uint64_t)g * 2 * kHop; p.b = p.a + kHop; p.keep_b = 1.0; p.row = 2 * (uint64_t)g;
      return p;
    }
    while (g >= st_end) { st = streams[++si]; st_end = st.pair_base + (st.frames + 1) / 2; }
    const uint32_t fa = 2 * (g - st.pair_base);
    p.has_b = fa + 1 < st.frames;
    p.a = pcm + st.pcm_off + (uint64_t)fa * kHop;
    p.b = p.has_b ? p.a + kHop : p.a;
    p.keep_b = p.has_b ? 1.0 : 0.0;
    p.row = (uint64_t)st.frame_base + fa;
    return p;
  };
  int16_t ra[16], rb[16];
  double wv[16];
  auto issue_loads = [&](const PairSrc &p) {
    int tt = t;
    asm volatile("" : "+v"(tt));
#pragma unroll
    for (int k = 0; k < 16; k++) {
      if (F & 2) { ra[k] = (int16_t)(tt + k); rb[k] = (int16_t)(tt - k); }
      else { ra[k] = p.a[tt + 256 * k]; rb[k] = p.b[tt + 256 * k]; }
    }
#pragma unroll
    for (int k = 0; k < 16; k++) {
      if (F & 1) { wv[k] = 3.0e-5; asm volatile("" : "+v"(wv[k])); } else wv[k] = window[tt + 256 * k];
    }
  };
  cd fv[core::kClassLaneMax];
  auto fold_issue = [&]() {
    if (folds) {
      uint32_t fb = fold_bounds;
      asm volatile("" : "+v"(fb));
      core::class_lane_load(lds, (int)(fb & 0xffffu), (int)(fb >> 16), fold_l, fv);
    }
  };
  auto fold_finish = [&](const PairSrc &p) {
    if (folds) {
      cd acc = core::class_lane_add(fv);
#pragma unroll
      for (int off = 1; off < 16; off <<= 1) acc = cd{acc.x + __shfl_xor(acc.x, off, 16), acc.y + __shfl_xor(acc.y, off, 16)};
      if (fold_l == 0) {
        chroma[p.row * kBands + fold_c] = acc.x;
        if (p.has_b) chroma[(p.row + 1) * kBands + fold_c] = acc.y;
      }
    }
  };
  PairSrc cur = locate(first), prev = cur;
  issue_loads(cur);
  for (uint32_t g = first; g < last; g++) {
    int tt = t;
    asm volatile("" : "+v"(tt));
    if (!(F & 4) && g != first) fold_issue();
    cd r[16];
#pragma unroll
    for (int k = 0; k < 16; k++) r[k] = cd{(double)ra[k] * wv[k], (double)rb[k] * wv[k]};
    if (!(F & 4) && g != first) fold_finish(prev);
    core::fft16(r);
    lds_barrier();
    core::dif0_store(tt, base0, lds, r);
    lds_barrier();
    core::dif1(tt, base1, lds, r);
    wave_lds_fence();
    core::dif2(tt, lds, r);
    core::dif2_publish(tt, lds, r);
    lds_barrier();
    double keep = 0;
#pragma unroll
    for (int j = 0; j < core::kBinsPerThread; j++) {
      const uint32_t idx = (slot_pk[j >> 1] >> (16 * (j & 1))) & 0xffffu;
      double pa, pb;
      core::dif_bin_power_any(tt, j, lds, r, &pa, &pb);
      if (F & 4) keep += pa + pb;
      else lds[idx] = cd{pa, pb};
    }
    const PairSrc nxt = locate(min(g + 1, last - 1));
    issue_loads(nxt);
    if ((F & 4) && keep == 1.2345) chroma[cur.row] = keep;
    lds_barrier();
    prev = cur;
    cur = nxt;
  }
  if (!(F & 4)) {
    fold_issue();
    fold_finish(prev);
  }
}

int main(int argc, char **argv) {
  const int eps = 28, frames = 5813, pairs_per_ep = (frames + 1) / 2;
  const size_t samples_per_ep = 7938000;
  const uint32_t total_pairs = eps * pairs_per_ep;
  std::vector<int16_t> pcm(samples_per_ep * eps + 8192);
  for (size_t i = 0; i < pcm.size(); i++) pcm[i] = (int16_t)((i * 2654435761u) >> 17);
  std::vector<Stream> st(eps);
  for (int e = 0; e < eps; e++) st[e] = Stream{samples_per_ep * e, (uint32_t)frames, (uint32_t)(frames * e), (uint32_t)(pairs_per_ep * e), 0};
  std::vector<cd> tw(4096); std::vector<double> win(4096);
  for (int k = 0; k < 4096; k++) { tw[k] = cd{std::cos(-2 * M_PI * k / 4096), std::sin(-2 * M_PI * k / 4096)}; win[k] = (0.54 - 0.46 * std::cos(2 * M_PI * k / 4095)) / 32767; }
  std::vector<uint16_t> slot(core::kNumBins); std::vector<uint32_t> cs(13);
  for (int i = 0; i < core::kNumBins; i++) slot[i] = (uint16_t)core::dif_power_slot(i);
  for (int c = 0; c <= 12; c++) cs[c] = c * 108;
  cs[12] = core::kNumBins;
  int16_t *d_pcm; Stream *d_st; cd *d_tw; double *d_win, *d_chroma; uint16_t *d_slot; uint32_t *d_cs;
  (void)hipMalloc(&d_pcm, pcm.size() * 2); (void)hipMalloc(&d_st, st.size() * sizeof(Stream)); (void)hipMalloc(&d_tw, 4096 * sizeof(cd));
  (void)hipMalloc(&d_win, 4096 * 8); (void)hipMalloc(&d_chroma, (size_t)eps * (frames + 1) * 12 * 8); (void)hipMalloc(&d_slot, slot.size() * 2); (void)hipMalloc(&d_cs, 13 * 4);
  (void)hipMemcpy(d_pcm, pcm.data(), pcm.size() * 2, hipMemcpyHostToDevice); (void)hipMemcpy(d_st, st.data(), st.size() * sizeof(Stream), hipMemcpyHostToDevice);
  (void)hipMemcpy(d_tw, tw.data(), 4096 * sizeof(cd), hipMemcpyHostToDevice); (void)hipMemcpy(d_win, win.data(), 4096 * 8, hipMemcpyHostToDevice);
  (void)hipMemcpy(d_slot, slot.data(), slot.size() * 2, hipMemcpyHostToDevice); (void)hipMemcpy(d_cs, cs.data(), 13 * 4, hipMemcpyHostToDevice);
  const size_t lds = core::kLds2Slots * sizeof(cd);
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  size_t lds_req = lds;
  auto run = [&](auto kern, const char *name, uint32_t ppb, int skew) {
    const size_t lds = lds_req;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const uint32_t grid = (total_pairs + ppb - 1) / ppb;
    float best = 1e9;
    for (int rep = 0; rep < 6; rep++) {
      (void)hipEventRecord(a);
      kern<<<grid, 256, lds>>>(d_pcm, d_st, eps, d_tw, d_win, d_slot, d_cs, d_chroma, total_pairs, ppb, skew);
      (void)hipEventRecord(b); (void)hipEventSynchronize(b);
      float ms; (void)hipEventElapsedTime(&ms, a, b);
      if (rep && ms < best) best = ms;
    }
    printf("%-44s ppb=%3u skew=%3d  %.3f ms\n", name, ppb, skew, best);
  };
  run(kernel<0>, "full", 16, 0);
  run(kernel<1>, "no window loads", 16, 0);
  run(kernel<4>, "no fold", 16, 0);
  run(kernel<15>, "FFT core only", 16, 0);
  return 0;
}
