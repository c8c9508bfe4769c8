// Resampler on the matrix cores: the kernel for decimation steps M >= 64 with M % 4 == 0 (48, 32, 24, 16, 8 kHz ...).
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
// Staging is the job of kProducers waves of the workgroup (the others multiply): thread pt of them takes row pt % 16 and
// every (4 kProducers)-th group of four samples of it: one 8- or 16-byte load, down-mix, four ds_write_b32 sixteen words
// apart (the 32 lanes of an LDS lane group then differ in the row -- 16 banks -- and in two groups: 2-way, which costs a
// store nothing).  LDS holds two tiles: while tile t is multiplied, tile t + 1 goes from the staging threads' registers
// to the other buffer and the loads of tile t + 2 take its place in the registers (stage_pass), one barrier per tile.
// Which SIMD a wave runs on matters more than anything else here: see the role layout in resample.hip.
#pragma once

namespace mfma_rs {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kRows = 16;        // rows of a tile = the M dimension of the MFMA
constexpr int kProducers = 3;    // waves of a workgroup that stage samples; the others multiply
constexpr int kConsumers = 10;   // multiplying waves of a workgroup at most (see the role layout in resample.hip)
constexpr int kPrefetch = 14;    // groups of its row a staging thread holds in registers for the tile after the next
constexpr int kCovered = kPrefetch * 4 * kProducers;  // groups of a row the staging threads of that row move themselves

struct Geom {
  int L, M, half, delta;
  int nblocks;         // blocks of sixteen outputs per row: ceil(L / 16)
  int blocks_per_wg;   // = waves per workgroup
  int splits;          // workgroups per tile
  // what wave w of a workgroup does: 0..15 multiplies block jb0 + role[w], 0x80 + k is staging wave k, 0xFF leaves at once
  // (waves w, w + 4, w + 8 ... share a SIMD: the layout decides which waves compete for one)
  unsigned char role[16];
  // A row's groups from kCovered on are the next row's groups dup_lo .. dup_hi - 1 (rows lie M samples apart): whoever
  // converts one of those writes it to both places; the last row's come from the sixteen rows' successor ("row 16"),
  // one extra group for the first dup_hi - dup_lo staging threads.  dup_hi = 0: every row fits kCovered groups.
  int m_groups;        // M / 4
  int dup_lo, dup_hi;
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
    typedef float v2f __attribute__((ext_vector_type(2)));
    auto sum = [](int w) { return (float)((int)(int16_t)w + (w >> 16)); };
    const v2f lo = v2f{sum(v.x), sum(v.y)} * 0.5f, hi = v2f{sum(v.z), sum(v.w)} * 0.5f;  // (v_pk_mul_f32)
    return float4{__builtin_truncf(lo.x), __builtin_truncf(lo.y), __builtin_truncf(hi.x), __builtin_truncf(hi.y)};
  }
}

// where a tile's samples come from
struct TileSrc {
  uint64_t in_off;     // the stream's first value, in s16 values from the arena's start
  uint64_t n_in;
  long long first0;    // input sample at [0][row 0] of the region
  bool fast;           // the tile lies inside its stream and the stream is 16-byte aligned: staged by aligned groups
};

__device__ __forceinline__ bool tile_is_fast(const int16_t *src, uint64_t n_in, long long first0, const Geom &geo) {
  const long long tile_last = first0 + max((long long)(kRows - 1) * geo.M + 4ll * kCovered, (long long)kRows * geo.M + 4ll * geo.dup_hi);
  return (reinterpret_cast<uintptr_t>(src) & 15) == 0 && first0 >= 0 && tile_last <= (long long)n_in;
}

// what a staging thread carries from the loads of a tile to its LDS writes
template <int CH>
struct Staged {
  typename Raw<CH>::type v[kPrefetch];
  typename Raw<CH>::type tail;  // its group of "row 16"
};

// One pass of a staging thread over its groups: group u of tile `cur` goes from its register to LDS (the wait is for that
// load only), and the same register receives group u of tile `next` at once -- a thread always has its loads in flight,
// each with a whole tile period to arrive, and the memory pipeline never drains while a tile is written.  (Separate
// passes -- write everything, then issue everything -- left HBM idle for the writes and the barrier: 40 % of the time.)
// Every thread moves exactly kPrefetch groups of its row, g0 + u gstep: no clamps and no conditions, so the loads are
// one address register plus immediate offsets.  A sample is converted ONCE: the part of a row's window that is the
// start of the next row's is written to both (the conversions are VALU work on a SIMD that also multiplies, and there
// they add to the MFMA time; staging every row's whole window converted 1.4 samples per sample).
// LAB: 1 no loads, 2 no writes.
template <int CH, int LAB>
__device__ __forceinline__ void stage_pass(Staged<CH> &sg, float *lds, const int16_t *__restrict__ in, bool have_cur,
                                           const TileSrc cur, bool have_next, const TileSrc next, const Geom &geo,
                                           int groups, int pt) {
  using raw_t = typename Raw<CH>::type;
  const int row = pt & 15, g0 = pt >> 4;
  constexpr int gstep = 4 * kProducers;
  const uint32_t row_groups = (uint32_t)(row * geo.M) >> 2;  // M is a multiple of 4
  // LDS stores as single ds_write_b32 with the slot's distance as the instruction's 16-bit offset: one address register for
  // all slots (the compiler's ds_write2_b32 reaches 1 KB, so it kept an address per slot -- and spilled them)
  const uint32_t lds_at = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) float *)lds;
  const uint32_t own_at = lds_at + (uint32_t)((4 * g0) * kRows + row) * 4u;
  const uint32_t dup_at = own_at - 4u + (uint32_t)(4 * geo.m_groups * kRows) * 4u;  // the row before, M samples later
#define NEEDLE_RS_PUT(at, f, off)                                                                                      \
  do {                                                                                                                 \
    asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(at), "v"((f).x), "n"((off)) : "memory");                         \
    asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(at), "v"((f).y), "n"((off) + kRows * 4) : "memory");             \
    asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(at), "v"((f).z), "n"((off) + 2 * kRows * 4) : "memory");         \
    asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(at), "v"((f).w), "n"((off) + 3 * kRows * 4) : "memory");         \
  } while (0)
  const int tail_count = geo.dup_hi - geo.dup_lo;  // > 0: rows are longer than kCovered groups
  const bool dup_row = row >= 1;
#define NEEDLE_RS_WRITE(u)                                                                                             \
  do {                                                                                                                 \
    const float4 f = group_f32<CH>(sg.v[u]);                                                                            \
    NEEDLE_RS_PUT(own_at, f, (u) * gstep * 4 * kRows * 4);                                                              \
    if ((u) * gstep < geo.dup_hi && ((u) + 1) * gstep > geo.dup_lo) { /* (uniform: the slot meets the duplicated range) */ \
      const int g = g0 + (u) * gstep;                                                                                   \
      if (dup_row && g >= geo.dup_lo && g < geo.dup_hi) NEEDLE_RS_PUT(dup_at, f, (u) * gstep * 4 * kRows * 4);          \
    }                                                                                                                   \
  } while (0)
  auto write_tail = [&]() {
    if (tail_count > 0) {
      const float4 f = group_f32<CH>(sg.tail);
      const uint32_t at = lds_at + (uint32_t)(4 * (geo.dup_lo + pt + geo.m_groups) * kRows + kRows - 1) * 4u;
      if (pt < tail_count) NEEDLE_RS_PUT(at, f, 0);
    }
  };
  // (the tiles by value and flags, not by pointer: a pointer that may be null keeps both structures in scratch memory)
  const bool cur_fast = have_cur && cur.fast, next_fast = have_next && next.fast;
  // (derived from the kernel argument in every path, so that the loads are global_load, not flat_load)
  const raw_t *tile_base = reinterpret_cast<const raw_t *>(in + (next_fast ? next.in_off + (uint64_t)CH * next.first0 : 0));
  const raw_t *base = tile_base + row_groups;
  const raw_t *tail_at = tile_base + (uint32_t)(kRows * geo.m_groups + geo.dup_lo + min(pt, max(tail_count, 1) - 1));
  if (cur_fast && next_fast) {  // the steady state
#pragma unroll
    for (int u = 0; u < kPrefetch; u++) {
      if (!(LAB & 2)) NEEDLE_RS_WRITE(u);
      else asm volatile("" ::"v"(sg.v[u].x), "v"(sg.v[u].y));  // (lab: the loads stay although nothing reads them)
      if (LAB & 128) sg.v[u] = tile_base[pt + 64 * kProducers * u];  // (lab: the same bytes per tile, fully coalesced)
      else if (!(LAB & 1)) sg.v[u] = base[g0 + u * gstep];
      else if (LAB & 32) asm volatile("v_mov_b32 %0, %1" : "=v"(sg.v[u].x) : "v"(pt + u));  // (lab: values the compiler cannot fold)
    }
    if (!(LAB & 2)) write_tail();
    if (tail_count > 0 && !(LAB & 1)) sg.tail = *tail_at;
    return;
  }
  if (cur_fast && !(LAB & 2)) {
#pragma unroll
    for (int u = 0; u < kPrefetch; u++) NEEDLE_RS_WRITE(u);
    write_tail();
  } else if (have_cur && !cur_fast && !(LAB & 2)) {  // first / last tiles of a stream, unaligned streams: sample by sample
    const long long from = cur.first0 + (long long)row * geo.M;
    for (int m = g0; m < 4 * groups; m += gstep)
      lds[(size_t)m * kRows + row] = (float)downmixed<CH>(in + cur.in_off, cur.n_in, from + m);
  }
  if (next_fast && !(LAB & 1)) {
#pragma unroll
    for (int u = 0; u < kPrefetch; u++) sg.v[u] = base[g0 + u * gstep];
    if (tail_count > 0) sg.tail = *tail_at;
  }
#undef NEEDLE_RS_WRITE
#undef NEEDLE_RS_PUT
}

// The workgroup barrier of this kernel: LDS traffic of the wave complete, then s_barrier -- and NOT __syncthreads(),
// whose fence also waits for the wave's global loads (vmcnt(0)): the staging waves arrive with the next tile's loads in
// flight, that is the point of them.
__device__ __forceinline__ void wg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// LAB & 16: s_memtime ticks every wave of block 0 spends working and at the barrier: [2 w] work, [2 w + 1] barrier of wave w,
// [32] tiles, [33..35] s_memrealtime (100 MHz) of block 0's first multiplying wave at entry, after the prologue barrier, at the end;
// [36..37] the same wave of the LAST block: entry and end
__device__ unsigned long long g_rs_clock[40];

// gridDim.x is a multiple of geo.splits; block_k0 is [2][nblocks]: window starts, then steps; coef_b is [nblocks][STEPS][64].
// blockDim.x = 1024: what each of the sixteen waves does is geo.role.
// LAB (timing experiments, wrong results; NEEDLE_HIP_LAB_BUILD only): 1 no global loads, 2 no LDS writes, 4 no MFMA loop,
// 8 no output stores, 16 clocks, 32 (with 1) opaque values in place of the loads: the conversions stay
template <int CH, int STEPS, int LAB = 0>
__global__ __launch_bounds__(1024) void resample_mfma_kernel(
    const int16_t *__restrict__ in, const RsStream *__restrict__ streams, int num_streams,
    const float *__restrict__ coef_b, const int *__restrict__ block_k0, Geom geo, uint32_t total_blocks, int buffer_floats,
    int16_t *__restrict__ out) {
  extern __shared__ float lds[];  // two buffers of [sample][row]
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;  // (uniform, and known to be)
  const int my_role = geo.role[wave];
  if (my_role == 0xFF) return;
  const bool stamp = (LAB & 16) && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1) && my_role == 0 && lane == 0;
  if (stamp) g_rs_clock[blockIdx.x == 0 ? 33 : 36] = __builtin_amdgcn_s_memrealtime();
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
    Staged<CH> sg;
    if (LAB & 1) {
      for (int u = 0; u < kPrefetch; u++) sg.v[u] = typename Raw<CH>::type{};
      sg.tail = typename Raw<CH>::type{};
    }
    Cursor cur;
    auto source = [&](uint32_t t) {
      uint64_t tile;
      locate(cur, blockIdx.x + t * gridDim.x, tile);
      TileSrc ts;
      ts.in_off = cur.st.in_off;
      ts.n_in = cur.st.n_in;
      ts.first0 = first_of((LAB & 64) ? (tile & 31) + 1 : tile);  // (lab 64: every load hits the L2)
      ts.fast = (LAB & 1) || tile_is_fast(in + ts.in_off, ts.n_in, ts.first0, geo);
      return ts;
    };
    TileSrc a = source(0), nx = a;
    stage_pass<CH, LAB>(sg, lds, in, false, a, true, a, geo, groups, pt);
    if (my_tiles > 1) nx = source(1);
    stage_pass<CH, LAB>(sg, lds, in, true, a, my_tiles > 1, nx, geo, groups, pt);
    wg_barrier();
    unsigned long long work = 0, wait = 0;
    for (uint32_t t = 0; t < my_tiles; t++) {
      const unsigned long long c0 = (LAB & 16) ? __builtin_amdgcn_s_memtime() : 0;
      if (t + 1 < my_tiles) {
        a = nx;
        if (t + 2 < my_tiles) nx = source(t + 2);
        stage_pass<CH, LAB>(sg, lds + ((t + 1) & 1) * buffer_floats, in, true, a, t + 2 < my_tiles, nx, geo, groups, pt);
      }
      const unsigned long long c2 = (LAB & 16) ? __builtin_amdgcn_s_memtime() : 0;
      wg_barrier();
      if (LAB & 16) {
        const unsigned long long c3 = __builtin_amdgcn_s_memtime();
        work += c2 - c0; wait += c3 - c2;
      }
    }
    if ((LAB & 16) && blockIdx.x == 0 && lane == 0) {
      g_rs_clock[2 * wave] = work; g_rs_clock[2 * wave + 1] = wait; g_rs_clock[32] = my_tiles;
    }
    return;
  }

  float b[STEPS];
#pragma unroll
  for (int s = 0; s < STEPS; s++) b[s] = live ? coef_b[((size_t)jb * STEPS + s) * 64 + lane] : 0.f;
  const int a_off = (live ? (block_k0[jb] - region_start) * kRows : 0) + lane;
  const int my_steps = live ? block_k0[geo.nblocks + jb] : 0;  // (the table's second half: steps per block, multiples of 4)
  const bool p_live = 16 * jb + (lane & 15) < geo.L;
  const int out_at = 4 * (lane >> 4) * geo.L + 16 * jb + (lane & 15);
  Cursor cur;
  uint64_t tile;
  locate(cur, blockIdx.x, tile);
  wg_barrier();
  if (stamp && blockIdx.x == 0) g_rs_clock[34] = __builtin_amdgcn_s_memrealtime();
  unsigned long long work = 0, wait = 0;
  for (uint32_t t = 0; t < my_tiles; t++) {
    const unsigned long long c0 = (LAB & 16) ? __builtin_amdgcn_s_memtime() : 0;
    if (live) {
      const RsStream st = cur.st;
      const float *a_ptr = lds + (t & 1) * buffer_floats + a_off;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      // Chunks of four steps, the next chunk's A operands read while this one is multiplied; a block whose windows
      // span fewer samples (the last one of a row: L % 16 outputs) stops early.
      float a[2][4];
#pragma unroll
      for (int k = 0; k < 4; k++) a[0][k] = a_ptr[64 * k];
#pragma unroll
      for (int c = 0; c < STEPS / 4; c++) {
        if (4 * c >= my_steps || ((LAB & 4) && c >= 2)) break;  // uniform
        if (c + 1 < STEPS / 4) {
#pragma unroll
          for (int k = 0; k < 4; k++) a[(c + 1) & 1][k] = a_ptr[64 * (4 * (c + 1) + k)];
        }
#pragma unroll
        for (int k = 0; k < 4; k++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c & 1][k], b[4 * c + k], acc, 0, 0, 0);
      }
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
    if (LAB & 16) {
      const unsigned long long c2 = __builtin_amdgcn_s_memtime();
      work += c1 - c0; wait += c2 - c1;
    }
  }
  if ((LAB & 16) && blockIdx.x == 0 && lane == 0) {
    g_rs_clock[2 * wave] = work; g_rs_clock[2 * wave + 1] = wait;
  }
  if (stamp) g_rs_clock[blockIdx.x == 0 ? 35 : 37] = __builtin_amdgcn_s_memrealtime();
}

}  // namespace mfma_rs
