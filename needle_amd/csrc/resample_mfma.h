// Resampler on the matrix cores: the kernel for decimation steps M >= 64 with M % 4 == 0 (48, 32, 16, 8, 96 kHz ...).
// Included by resample.hip only (it uses that file's RsStream and the down-mix helper).
//
// Specification and oracle are unchanged (oracle/ora_resample.h): every output is a chain of f32 fused multiply-adds in
// tap order.  v_mfma_f32_16x16x4_f32 IS such a chain -- D = fma(a3, b3, fma(a2, b2, fma(a1, b1, fma(a0, b0, C)))), one
// rounding per product, bit for bit (MI355X_MICROARCH.md, matrix cores) -- at 32 FMAs per clock per SIMD, where the
// DPP-operand v_fmac_f32 of resample_quad_kernel reaches 15.
//
// The product: a tile is sixteen rows of L outputs; rows are L outputs = M input samples apart, so output p of every
// row uses the same coefficient row and a window at the same offset from its row's first sample.  For the sixteen
// consecutive outputs 16 b .. 16 b + 15 of the rows ("block" b):
//     D[i][j] = sum_k A[i][k] B[k][j],  A[i][k] = sample k0(b) + k of row i,  B[k][j] = coefficient of output 16 b + j
// at tap k - (window start of output 16 b + j - k0(b)), ZERO outside its T taps.  k runs over the union of the
// sixteen windows, 4 STEPS >= T + the ~15 M / L samples between the first and the last window start (48 kHz: 205 of
// which 140 are taps: two thirds of the FMAs are useful; a zero coefficient leaves an accumulator unchanged, so the
// result is the oracle's chain).  A wave owns one block: its B operands (STEPS registers) never change and stay in
// registers while the persistent workgroup walks over its tiles; the A operand of step s is ONE ds_read_b32 at a
// compile-time offset, because the samples lie in LDS transposed, [sample][row]: lane l = 16 k + i of step s reads word
// 64 s + l of the block's window -- consecutive lanes, consecutive words, no bank conflict.
//
// Staging: thread t takes row t % 16 and every (blockDim / 16)-th group of four samples of it: one 8- or 16-byte load,
// down-mix, four ds_write_b32 sixteen words apart (the 32 lanes of an LDS lane group then differ in the row -- 16
// banks -- and in two groups: 2-way, which costs a store nothing).  The loads of the NEXT tile are issued before the
// MFMA loop of this one and written to LDS after it, so a workgroup's HBM round trip lies under its own arithmetic.
#pragma once

namespace mfma_rs {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kRows = 16;        // rows of a tile = the M dimension of the MFMA
constexpr int kProducers = 3;    // waves of a workgroup that stage samples; the others multiply
constexpr int kConsumers = 10;   // multiplying waves of a workgroup at most (see the role layout in resample.hip)
constexpr int kPrefetch = 19;    // groups a staging thread holds in registers for the tile after the next
constexpr int kStagedGroups = kPrefetch * 4 * kProducers;  // groups of a row a staging pass moves (>= the groups read)

struct Geom {
  int L, M, half, delta;
  int nblocks;         // blocks of sixteen outputs per row: ceil(L / 16)
  int blocks_per_wg;   // = waves per workgroup
  int splits;          // workgroups per tile
  // what wave w of a workgroup does: 0..15 multiplies block jb0 + role[w], 0x80 + k is staging wave k, 0xFF leaves at once
  // (waves w, w + 4, w + 8 ... share a SIMD: the layout decides which waves compete for one)
  unsigned char role[16];
};

// k0[b]: first sample of block b's window union, counted from the row's origin (input sample row * M - half + 1 -
// delta, a multiple of 4)

template <int CH>
__device__ __forceinline__ int downmixed(const int16_t *src, uint64_t n_in, long long idx) {
  const bool ok = idx >= 0 && (uint64_t)idx < n_in;
  const long long at = idx < 0 ? 0 : ((uint64_t)idx < n_in ? idx : (long long)n_in - 1);
  int sv;
  if (CH == 1) {
    sv = src[at];
  } else {
    const int v = reinterpret_cast<const int *>(src)[at];
    sv = ((int)(int16_t)v + (v >> 16)) / 2;  // integer down-mix, C truncation
  }
  return ok ? sv : 0;
}

template <int CH>
struct Raw { typedef typename std::conditional<CH == 1, int2, int4>::type type; };

// four down-mixed samples of one aligned group as f32.  The down-mix (L + R) / 2 with C truncation is done in f32 -- the sum
// is exact there, so are the halving and the truncation --: four VALU instructions per sample instead of five (these run on
// a SIMD that also multiplies, and on it they add to the MFMA time)
template <int CH>
__device__ __forceinline__ float4 group_f32(typename Raw<CH>::type v) {
  if constexpr (CH == 1) {
    return float4{(float)(int16_t)v.x, (float)(v.x >> 16), (float)(int16_t)v.y, (float)(v.y >> 16)};
  } else {
    auto mix = [](int w) { return __builtin_truncf((float)((int)(int16_t)w + (w >> 16)) * 0.5f); };
    return float4{mix(v.x), mix(v.y), mix(v.z), mix(v.w)};
  }
}

// where a tile's samples come from
struct TileSrc {
  uint64_t in_off;     // the stream's first value, in s16 values from the arena's start
  uint64_t n_in;
  long long first0;    // input sample at [0][row 0] of the region
  bool fast;           // the tile lies inside its stream and the stream is 16-byte aligned: staged by aligned groups
};

__device__ __forceinline__ bool tile_is_fast(const int16_t *src, uint64_t n_in, long long first0, int M) {
  const long long tile_last = first0 + (long long)(kRows - 1) * M + 4ll * kStagedGroups;
  return (reinterpret_cast<uintptr_t>(src) & 15) == 0 && first0 >= 0 && tile_last <= (long long)n_in;
}

// One pass of a staging thread over its groups: group u of tile `cur` goes from its register to LDS (the wait is for that
// load only), and the same register receives group u of tile `next` at once -- a thread always has kPrefetch loads in
// flight, and the memory pipeline never drains while a tile is written.  (Separate passes -- write everything, then
// issue everything -- left HBM idle for the writes and the barrier: 40 % of the time.)  LAB: 1 no loads, 2 no writes.
template <int CH, int LAB>
__device__ __forceinline__ void stage_pass(typename Raw<CH>::type (&v)[kPrefetch], float *lds,
                                           const int16_t *__restrict__ in, const TileSrc *cur, const TileSrc *next, int M,
                                           int groups, int pt) {
  using raw_t = typename Raw<CH>::type;
  const int row = pt & 15, g0 = pt >> 4, gstep = 4 * kProducers;
  const uint32_t row_groups = (uint32_t)(row * M) >> 2;  // M is a multiple of 4
  // Every thread moves exactly kPrefetch groups, g0 + u gstep: no clamps and no conditions, so the loads are one
  // address register plus immediate offsets (no VALU work that would have to squeeze in between the other waves' MFMAs)
  // and the buffer has kStagedGroups groups per row whether or not the blocks read the last ones.
  auto write = [&](int u) {
    const int g = g0 + u * gstep;
    const float4 f = group_f32<CH>(v[u]);
    float *dst = lds + (size_t)(4 * g) * kRows + row;
    dst[0] = f.x; dst[kRows] = f.y; dst[2 * kRows] = f.z; dst[3 * kRows] = f.w;
  };
  const bool cur_fast = cur && cur->fast, next_fast = next && next->fast;
  // (derived from the kernel argument in every path, so that the loads are global_load, not flat_load)
  const raw_t *base = reinterpret_cast<const raw_t *>(in + (next_fast ? next->in_off + (uint64_t)CH * next->first0 : 0)) + row_groups;
  if (cur_fast && next_fast) {  // the steady state: straight-line code, so that the wait before write u is vmcnt(kPrefetch - 1)
#pragma unroll
    for (int u = 0; u < kPrefetch; u++) {
      if (!(LAB & 2)) write(u);
      else asm volatile("" ::"v"(v[u].x), "v"(v[u].y));  // (lab: the loads stay although nothing reads them)
      if (!(LAB & 1)) v[u] = base[g0 + u * gstep];
      else if (LAB & 32) asm volatile("v_mov_b32 %0, %1" : "=v"(v[u].x) : "v"(pt + u));  // (lab: values the compiler cannot fold)
    }
    return;
  }
  if (cur_fast && !(LAB & 2)) {
#pragma unroll
    for (int u = 0; u < kPrefetch; u++) write(u);
  } else if (cur && !cur_fast && !(LAB & 2)) {  // first / last tiles of a stream, unaligned streams: sample by sample
    const long long from = cur->first0 + (long long)row * M;
    for (int m = g0; m < 4 * groups; m += gstep)
      lds[(size_t)m * kRows + row] = (float)downmixed<CH>(in + cur->in_off, cur->n_in, from + m);
  }
  if (next_fast && !(LAB & 1)) {
#pragma unroll
    for (int u = 0; u < kPrefetch; u++) v[u] = base[g0 + u * gstep];
  }
}

// The workgroup barrier of this kernel: LDS traffic of the wave complete, then s_barrier -- and NOT __syncthreads(),
// whose fence also waits for the wave's global loads (vmcnt(0)): the staging waves arrive with the next tile's loads in
// flight, that is the point of them.
__device__ __forceinline__ void wg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// One MFMA step per sample group with the A operands read `kAhead` steps in front of their use: the chain of dependent
// MFMAs (40 cycles each) then never waits for the LDS.
constexpr int kAhead = 8;

// LAB & 16: s_memtime ticks block 0 spends per phase: [0] multiply, [1] multiplying waves at the barrier, [2] a staging
// pass, [4] staging waves at the barrier, [5] tiles
__device__ unsigned long long g_rs_clock[8];

// gridDim.x is a multiple of geo.splits; block_k0 has geo.nblocks entries; coef_b is [nblocks][STEPS][64].
// blockDim.x = 1024: what each of the sixteen waves does is geo.role.
// LAB (timing experiments, wrong results; NEEDLE_HIP_LAB_BUILD only): 1 no global loads, 2 no LDS writes, 4 no MFMA loop,
// 8 no output stores, 16 clocks, 32 (with 1) opaque values in place of the loads: the conversions stay
template <int CH, int STEPS, int LAB = 0>
__global__ __launch_bounds__(1024) void resample_mfma_kernel(
    const int16_t *__restrict__ in, const RsStream *__restrict__ streams, int num_streams,
    const float *__restrict__ coef_b, const int *__restrict__ block_k0, Geom geo, uint32_t total_blocks, int buffer_floats,
    int16_t *__restrict__ out) {
  extern __shared__ float lds[];  // two buffers of [sample][row]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int my_role = geo.role[wave];
  if (my_role == 0xFF) return;
  const bool producer = my_role >= 0x80;  // wave-uniform
  const int pt = 64 * (my_role & 0x7F) + lane;  // a staging thread's index
  const int split = (int)(blockIdx.x % (uint32_t)geo.splits);
  const int jb0 = split * geo.blocks_per_wg, jb_end = min(jb0 + geo.blocks_per_wg, geo.nblocks);
  const int jb = jb0 + (producer ? 0 : my_role);
  const bool live = jb < jb_end;  // a multiplying wave with a block
  const int region_start = block_k0[jb0] & ~3;
  const int groups = (block_k0[jb_end - 1] + 4 * STEPS - region_start + 3) >> 2;

  // a workgroup's tiles have ascending block numbers: the stream of the next one is the current stream or a later one,
  // found by a walk that loads nothing while the tile stays in its stream (a binary search per tile is four dependent
  // scalar loads, ~1 us, on the critical path of a 3 us tile)
  struct Cursor {
    int at = -1;
    uint32_t next_base = 0;  // block_base of stream at + 1, or 2^32 - 1
    RsStream st;
  };
  auto locate = [&](Cursor &c, uint32_t vb, uint64_t &tile) {
    while (c.at < 0 || vb >= c.next_base) {
      c.at++;
      c.st = streams[c.at];
      c.next_base = c.at + 1 < num_streams ? streams[c.at + 1].block_base : 0xFFFFFFFFu;
    }
    tile = (vb - c.st.block_base) / (uint32_t)geo.splits;
  };
  auto first_of = [&](uint64_t tile) {
    return (long long)(tile * (uint64_t)kRows * (uint64_t)geo.M) - geo.half + 1 - geo.delta + region_start;
  };
  if (blockIdx.x >= total_blocks) return;
  const uint32_t my_tiles = (total_blocks - blockIdx.x + gridDim.x - 1) / gridDim.x;

  if (producer) {
    // The staging waves' few instructions go first: with equal priority the arbiter keeps issuing the other waves'
    // MFMAs (one is always ready), the pass starts when they are done, and the two phases ADD (measured: 0.31 ms of
    // MFMA + 0.35 ms of staging = 0.58 ms).
    __builtin_amdgcn_s_setprio(3);
    // period t: tile t + 1 goes from registers to buffer (t + 1) & 1 while the loads of tile t + 2 take its place
    typename Raw<CH>::type v[kPrefetch];
    if (LAB & 1)
      for (int u = 0; u < kPrefetch; u++) v[u] = typename Raw<CH>::type{};
    Cursor cur;
    auto source = [&](uint32_t t) {
      uint64_t tile;
      locate(cur, blockIdx.x + t * gridDim.x, tile);
      TileSrc ts;
      ts.in_off = cur.st.in_off;
      ts.n_in = cur.st.n_in;
      ts.first0 = first_of((LAB & 64) ? (tile & 31) + 1 : tile);  // (lab 64: every load hits the L2)
      ts.fast = (LAB & 1) || tile_is_fast(in + ts.in_off, ts.n_in, ts.first0, geo.M);
      return ts;
    };
    TileSrc a = source(0), nx = a;
    stage_pass<CH, LAB>(v, lds, in, nullptr, &a, geo.M, groups, pt);
    if (my_tiles > 1) nx = source(1);
    stage_pass<CH, LAB>(v, lds, in, &a, my_tiles > 1 ? &nx : nullptr, geo.M, groups, pt);
    wg_barrier();
    for (uint32_t t = 0; t < my_tiles; t++) {
      const unsigned long long c0 = (LAB & 16) ? __builtin_amdgcn_s_memtime() : 0;
      if (t + 1 < my_tiles) {
        a = nx;
        if (t + 2 < my_tiles) nx = source(t + 2);
        stage_pass<CH, LAB>(v, lds + ((t + 1) & 1) * buffer_floats, in, &a, t + 2 < my_tiles ? &nx : nullptr, geo.M, groups, pt);
      }
      const unsigned long long c2 = (LAB & 16) ? __builtin_amdgcn_s_memtime() : 0;
      wg_barrier();
      if ((LAB & 16) && blockIdx.x == 0 && pt == 0) {
        const unsigned long long c3 = __builtin_amdgcn_s_memtime();
        g_rs_clock[2] += c2 - c0; g_rs_clock[4] += c3 - c2; g_rs_clock[5] += 1;
      }
    }
    return;
  }

  float b[STEPS];
#pragma unroll
  for (int s = 0; s < STEPS; s++) b[s] = live ? coef_b[((size_t)jb * STEPS + s) * 64 + lane] : 0.f;
  const int a_off = (live ? (block_k0[jb] - region_start) * kRows : 0) + lane;
  const bool p_live = 16 * jb + (lane & 15) < geo.L;
  const int out_at = 4 * (lane >> 4) * geo.L + 16 * jb + (lane & 15);
  Cursor cur;
  uint64_t tile;
  locate(cur, blockIdx.x, tile);
  wg_barrier();
  for (uint32_t t = 0; t < my_tiles; t++) {
    const unsigned long long c0 = (LAB & 16) ? __builtin_amdgcn_s_memtime() : 0;
    if (live) {
      const RsStream st = cur.st;
      const float *a_ptr = lds + (t & 1) * buffer_floats + a_off;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      constexpr int kSteps = (LAB & 4) ? kAhead : STEPS;
      float a[kSteps];
#pragma unroll
      for (int s = 0; s < kSteps; s++) a[s] = a_ptr[64 * s];
#pragma unroll
      for (int s = 0; s < kSteps; s++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[s], acc, 0, 0, 0);
      // the schedule: kAhead reads, then a read per MFMA, then the last kAhead MFMAs
      __builtin_amdgcn_sched_group_barrier(0x100, kAhead, 0);
#pragma unroll
      for (int s = 0; s < kSteps - kAhead; s++) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, kAhead, 0);
      // outputs: lane (j = lane & 15, rows 4 (lane >> 4) .. + 3); only a stream's last tile needs the bounds check
      const uint64_t tile_first = tile * (uint64_t)(kRows * geo.L);
      int16_t *dst = out + st.out_off + tile_first;
      const bool whole = tile_first + (uint64_t)(kRows * geo.L) <= st.n_out;  // uniform
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int at = out_at + r * geo.L;  // (row 4 (lane >> 4) + r) L + p
        const float v = __builtin_amdgcn_fmed3f(rintf(acc[r]), -32768.0f, 32767.0f);
        if (p_live && (whole || tile_first + (uint64_t)at < st.n_out) && (!(LAB & 8) || acc[r] == 12345.f)) dst[at] = (int16_t)v;
      }
    }
    if (t + 1 < my_tiles) locate(cur, blockIdx.x + (t + 1) * gridDim.x, tile);
    const unsigned long long c1 = (LAB & 16) ? __builtin_amdgcn_s_memtime() : 0;
    wg_barrier();  // every wave is done with this buffer; the next one is complete
    if ((LAB & 16) && blockIdx.x == 0 && my_role == 0 && lane == 0) {
      const unsigned long long c2 = __builtin_amdgcn_s_memtime();
      g_rs_clock[0] += c1 - c0; g_rs_clock[1] += c2 - c1;
    }
  }
}

}  // namespace mfma_rs
