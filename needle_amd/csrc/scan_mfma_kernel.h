// The sampled scan's first stage on the matrix pipe, second form (round 5).  Included by search.hip inside its anonymous
// namespace, after SearchProblem, mfma_windows() and the mfma_v4i / mfma_v16i types.
//
// Same aligned windows, same candidates and same runs as hamming_runs_sampled_kernel (comparator.rs:176-200 is what all of
// them reproduce).  The Hamming distances of a window's head rows against every destination position are int8 matrix
// products: a hash as 32 bytes of +-1 gives dot(a, b) = 32 - 2 d(a, b).  A tile is 32 aligned windows x 32 destination
// positions and costs four v_mfma_i32_32x32x32_i8 (head rows 0, 2, 4, 7 of a window of 8).
//
// Round 4's form kept the matrix pipe 27 % busy: beside its 4 products a tile cost 88 vector and 16 scalar instructions, a
// wave runs its instructions in order at 5 - 8 cycles each, and three waves per SIMD (166 registers) cannot cover that.
// What was measured on the way to this form (profiles/NOTES.md, round 5): what costs most is not arithmetic but BRANCHES
// per tile (every tile with a survivor took a scalar detour of ~100 instructions) and trips to global memory inside the
// tile loop.  Hence:
//  * The products only FLAG.  The four products of a tile accumulate into two register sets -- rows (0 + 2) and rows
//    (4 + 7), each preset so that its sign bit is SET where the SUM of two rows' distances is <= 2 t, a necessary
//    condition of "both <= t" (0.20 % of the window-diagonals pass on synthetic audio, 0.05 % the exact head test).
//    16 + 16 result registers are folded into four words (one per group of four windows) and the four sign bits are
//    shifted into a per-lane flag word (v_alignbit): 20 vector instructions, no compare, no branch, eight tiles in a
//    straight line.
//  * Every eight tiles the flag words become ITEMS = (group of four windows, destination position), handed out one per
//    lane, 64 at a time, whatever lane flagged them (a position that looks like many windows -- a sustained sound --
//    flags the same lane again and again: left to that lane, the wave waits for it).  A lane tests its item's four
//    windows' head rows EXACTLY with popcounts (the vector form's test), then the tail rows of what passes, then
//    kM2Probe rows on either side (a whole window whose run ends inside them is shorter than any min_len this path
//    takes: dropped) -- all from LDS.  What remains (a few windows per pair: real runs) is resolved by the wave against
//    the source sequence in global memory, two rows per lane and direction in one trip.
//  * One workgroup = one destination x up to EIGHT sources (as many as its share of the CU's LDS holds): staging and the
//    expansion of the destination's hashes into B fragments are shared by 7 - 19 row tiles instead of 5; an A fragment
//    read serves both column blocks of a unit; waves take units from a counter in LDS, so none idles while another still
//    has units; a workgroup finds its group with one load (round 4: a binary search of the table, 17 dependent loads).
//  * Default shape (mfma_waves(), search.hip): workgroups of 8 waves, two per CU, four waves per SIMD (128 registers: one
//    accumulator pair, a tile folded before the next is multiplied).  Measured against it: 16 waves x 1 (same), 12 waves
//    x 1 with two accumulator pairs and the next tile's products issued before the fold (5 % slower), 4 waves x 3 (45 %).
//  * Columns outside the table get all-zero B fragments and rows beyond the last window read a row of zeros
//    (product 0 + preset > 0: never flagged).
// Operand maps of the instruction: tools/mfma_i8_layout.hip.  Needs t <= 15.
#ifndef NEEDLE_M2_LAB
#define NEEDLE_M2_LAB 0                       // timing laboratory (tools/build_variant.sh), WRONG results: 1 flags ignored (the tile loop
                                              // alone), 2 head survivors dropped, 64 a workgroup's setup alone
#endif
constexpr int kM2Heads = 4;
constexpr int kM2Members = 8;                 // sources a workgroup takes at most (of one destination)
constexpr int kM2Pitch = kM2Heads * 8 + 4;    // words of a window's row in the A image: 4 x 32 bytes of +-1, the window's member << 28 | w0, 3 spare
                                              // (36: lanes 32 words apart would all meet in two groups of LDS banks)
constexpr int kM2Probe = 4;                   // rows tested per lane on either side of a whole window
constexpr int kM2Rows = kSampleW + 2 * kM2Probe;  // source hashes kept per window: rows w0 - 4 .. w0 + 11
constexpr int kM2ColBlocks = 2;               // column blocks of 32 positions a wave takes per unit
constexpr int kM2Batch = 4;                   // row tiles (x 2 column blocks = 8 tiles = 32 flag bits) between two looks at the flags
constexpr int kM2Queue = 128;                 // items a wave can hold: < 64 waiting + the <= 64 one turn adds
constexpr int kM2CtlWords = 36;               // [0] unit counter, [1 + g] first row of member g (g = members: all rows), [10 + g] source offset, [18 + g] source length, [26 + g] first window of its image
static_assert((kM2Batch & (kM2Batch - 1)) == 0 && 8 * kM2Batch <= 32, "a batch's flags fill at most one word");
static_assert(kM2Probe == 4 && kM2Rows == 16, "a window's sixteen source hashes are read as 16-byte words");
static_assert(kSampleW + 2 * kM2Probe - 2 < 2 * kSampleW - 1 + 8, "a run that ends inside the probed rows must be shorter than any min_len the sampled path takes");

// LDS words of a workgroup: staged destination (+ 64 zeros), tables, per-window source hashes, A image
__host__ __device__ constexpr size_t m2_round4(size_t x) { return (x + 3) & ~(size_t)3; }
__host__ __device__ constexpr size_t m2_lds_words(uint64_t m, uint64_t windows, int waves) {
  return m2_round4(m + 64) + kM2CtlWords + 16 + (size_t)waves * kM2Queue + (size_t)kM2Rows * windows + (size_t)(windows + 1) * kM2Pitch;
}

// A source sequence's windows as a workgroup wants them in LDS -- per window kM2Pitch words of the A image (its four head
// hashes as 32 negated +-1 bytes each, then w0) and kM2Rows source hashes around it -- built ONCE per launch in global
// memory (m2_window_images_kernel) instead of by every workgroup that takes the sequence as a source: at 2000 videos a
// sequence is a source 1999 times, and gathering + expanding its windows was a tenth of the scan (a workgroup's setup
// alone: 0.44 of 4.0 ms at 79 800 pairs).  A workgroup copies its members' images with 16-byte loads.
constexpr int kM2ImageWords = kM2Pitch + kM2Rows;   // per window
struct M2ImageSeq {
  uint32_t src_off, n;     // the sequence in the hash arena
  uint32_t first_window;   // of its image, in windows
  uint32_t windows;
};
template <int W>
__global__ __launch_bounds__(256) void m2_window_images_kernel(const uint32_t *__restrict__ hashes, const M2ImageSeq *__restrict__ seqs,
                                                               int num_seqs, uint32_t total_windows, uint32_t min_len,
                                                               uint32_t *__restrict__ images) {
  constexpr int H = kM2Heads, PITCH = kM2Pitch, E = kM2Probe, NR = kM2Rows;
  const uint32_t P = min_len - W + 1;
  for (uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total_windows * NR; idx += gridDim.x * blockDim.x) {
    const uint32_t k = idx / NR;
    const int s = (int)(idx % NR) - E;             // row w0 + s
    int lo = 0, hi = num_seqs - 1;                 // the sequence whose image holds window k
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (seqs[mid].first_window <= k) lo = mid; else hi = mid - 1;
    }
    const M2ImageSeq q = seqs[lo];
    const uint32_t w0 = 1u + (k - q.first_window) * P;
    const int row = (int)w0 + s;
    const uint32_t hsh = row >= 0 && row < (int)q.n ? hashes[q.src_off + (uint32_t)row] : 0u;
    uint32_t *img = images + (size_t)k * kM2ImageWords;
    img[PITCH + s + E] = hsh;
    if (s == 0) img[8 * H] = w0;
    if (s == 1 || s == 3 || s == 5) img[8 * H + (s + 1) / 2] = 0u;   // (the row's three spare words)
    if (s == 0 || s == 2 || s == 4 || s == 7) {
      uint32_t *o = img + 8 * (s == 7 ? 3 : s >> 1);
#pragma unroll
      for (int b = 0; b < 8; b++) {
        const uint32_t nib = (~hsh >> (4 * b)) & 0xFu;   // negated: a set bit becomes -1
        uint32_t w = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) w |= (((nib >> i) & 1u) ? 0x01u : 0xFFu) << (8 * i);
        o[b] = w;
      }
    }
  }
}

// WAVES per workgroup; PER_SIMD waves the registers have to allow on a SIMD (3: up to 168 registers, the tiles software-
// pipelined over two accumulator pairs; 4: up to 128, one pair, a tile folded before the next is multiplied).
// (Measured and dropped, commit d27d279: a 16-wave form with the members' WHOLE source sequences staged in LDS, so that the
// resolution of real runs never leaves the CU -- half as many sources per workgroup and their staging cost more than the
// trips to global memory it saved: 4.59 against 3.89 ms at 79 800 pairs of 45-minute windows, 1.13 against 0.98 at 39 060
// pairs of 24-minute ones.)
template <int W, int WAVES, int PER_SIMD>
__global__ __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(PER_SIMD, PER_SIMD))) void hamming_runs_mfma2_kernel(
    const uint32_t *__restrict__ hashes, const SearchProblem *__restrict__ problems, int num_problems, uint32_t threshold,
    NeedleHipRun *__restrict__ runs, uint32_t capacity, uint32_t *__restrict__ count, int splits,
    const uint32_t *__restrict__ images) {
  static_assert(W == 8, "head rows {0, 2, 4, 7} and tail rows {1, 3, 5, 6} of a window of 8");
  constexpr int H = kM2Heads, PITCH = kM2Pitch, CB = kM2ColBlocks, E = kM2Probe, NR = kM2Rows;
  extern __shared__ uint32_t lds[];
  __builtin_amdgcn_s_setprio(3);
  // Which group: workgroup / splits.  Its first entry's index lies in the pad field (bits 8 .. 30) of the table entry whose
  // POSITION is the group's number (build_plan) -- one load, not a binary search of the table (17 dependent trips to
  // global memory at 80 000 pairs: half the life of a workgroup of 24-minute windows).
  const int group = (int)(blockIdx.x / (uint32_t)splits);
  const int lo = (int)((problems[group].pad >> 8) & 0x7FFFFFu);
  (void)num_problems;
  const SearchProblem pr = problems[lo];
  const int members = min((int)(pr.pad & 0xFFu), kM2Members - 1) + 1;  // pad of a group's first entry: how many followers (build_plan)
  const int m = (int)pr.m, min_len = (int)pr.min_len;
  const int P = min_len - W + 1;
  const uint32_t *__restrict__ dst = hashes + pr.dst_off;
  int nW = 0;  // rows of the tiles: the windows of all members
  for (int g = 0; g < members; g++) nW += mfma_windows((int)problems[lo + g].n, min_len, W);

  const int dst_words = (int)m2_round4((size_t)m + 64);
  uint32_t *ldst = lds;                          // ldst[j] = dst[j], zeros behind
  uint32_t *ctl = lds + dst_words;
  uint32_t *ntab = ctl + kM2CtlWords;            // 4 bits -> 4 bytes of +-1
  uint32_t *queues = ntab + 16;                  // per wave: items waiting for their exact test
  // per window the source hashes of rows w0 - E .. w0 + W + E - 1 (zero where there is none)
  uint32_t *wsrc = queues + WAVES * kM2Queue;
  uint32_t *aimg = wsrc + NR * nW;               // 16-byte aligned: every size above is a multiple of 4 words

  if (threadIdx.x == 0) {
    ctl[0] = 0u;
    int rows = 0;
    for (int g = 0; g < members; g++) {
      ctl[1 + g] = (uint32_t)rows;
      ctl[10 + g] = problems[lo + g].src_off;
      ctl[18 + g] = problems[lo + g].n;
      ctl[26 + g] = problems[lo + g].block_base;
      rows += mfma_windows((int)problems[lo + g].n, min_len, W);
    }
    ctl[1 + members] = (uint32_t)rows;
  }
  if (threadIdx.x < 16) {
    uint32_t w = 0;
    for (int i = 0; i < 4; i++) w |= (((threadIdx.x >> i) & 1) ? 0x01u : 0xFFu) << (8 * i);
    ntab[threadIdx.x] = w;
  }
  {
    constexpr int kU = 4;
    const int nt = 64 * WAVES;
    int k = threadIdx.x;
    for (; k + (kU - 1) * nt < m; k += kU * nt) {
      uint32_t v[kU];
#pragma unroll
      for (int u = 0; u < kU; u++) v[u] = dst[k + u * nt];
#pragma unroll
      for (int u = 0; u < kU; u++) ldst[k + u * nt] = v[u];
    }
    for (; k < dst_words; k += nt) ldst[k] = k < m ? dst[k] : 0u;
  }
  __syncthreads();
  if (images != nullptr) {
    // the members' window images, built once per launch (m2_window_images_kernel): SearchProblem::block_base of an entry =
    // first window of its source's image.  16-byte pieces: 9 of a window's A-image row, 4 of its source hashes.
    constexpr int kPieces = kM2ImageWords / 4;
    static_assert(kM2ImageWords % 4 == 0 && PITCH % 4 == 0, "");
    constexpr int kU = 4;                         // pieces in flight per thread
    for (int base = threadIdx.x; base < nW * kPieces; base += kU * 64 * WAVES) {
      mfma_v4i v[kU];
      uint32_t *to[kU];
#pragma unroll
      for (int u = 0; u < kU; u++) {
        const int idx = min(base + u * 64 * WAVES, nW * kPieces - 1);
        const int k = idx / kPieces, piece = idx % kPieces;
        int g = 0;
        for (int i = 1; i < members; i++) g += (uint32_t)k >= ctl[1 + i] ? 1 : 0;
        const uint32_t from = ctl[26 + g] + ((uint32_t)k - ctl[1 + g]);
        v[u] = *reinterpret_cast<const mfma_v4i *>(images + (size_t)from * kM2ImageWords + 4 * piece);
        if (piece == PITCH / 4 - 1) v[u][0] |= (int)((uint32_t)g << 28);   // w0 -> member << 28 | w0
        to[u] = piece < PITCH / 4 ? aimg + k * PITCH + 4 * piece : wsrc + NR * k + 4 * (piece - PITCH / 4);
      }
#pragma unroll
      for (int u = 0; u < kU; u++)
        if (base + u * 64 * WAVES < nW * kPieces) *reinterpret_cast<mfma_v4i *>(to[u]) = v[u];
    }
  } else {
  // The windows' source hashes and the A image: one thread per (window, one of its NR rows), four windows per turn with
  // their four loads in flight together (a workgroup of 24-minute windows spent as long here, one dependent trip to
  // global memory per window, as on its tiles).  64 WAVES is a multiple of NR: a thread keeps its row s.
  {
    static_assert((64 * WAVES) % NR == 0, "");
    constexpr int kU = 4, kStep = 64 * WAVES / NR;                   // windows a turn of the workgroup covers: kU kStep
    const int s = (int)(threadIdx.x % NR) - E;                       // row w0 + s, s = -E .. W + E - 1
    for (int k0 = (int)(threadIdx.x / NR); k0 < nW; k0 += kU * kStep) {
      uint32_t hv[kU], wm[kU];
#pragma unroll
      for (int u = 0; u < kU; u++) {
        const int k = min(k0 + u * kStep, nW - 1);
        int g = 0;
        for (int i = 1; i < members; i++) g += (uint32_t)k >= ctl[1 + i] ? 1 : 0;
        const int w0 = 1 + (k - (int)ctl[1 + g]) * P;
        const int row = w0 + s;
        wm[u] = ((uint32_t)g << 28) | (uint32_t)w0;
        hv[u] = row >= 0 && row < (int)ctl[18 + g] ? hashes[ctl[10 + g] + (uint32_t)row] : 0u;
      }
#pragma unroll
      for (int u = 0; u < kU; u++) {
        const int k = k0 + u * kStep;
        if (k >= nW) break;
        wsrc[NR * k + s + E] = hv[u];
        if (s == 0) aimg[k * PITCH + 8 * H] = wm[u];
        if (s == 0 || s == 2 || s == 4 || s == 7) {                  // head rows 0 .. 3
          uint32_t *o = aimg + k * PITCH + 8 * (s == 7 ? 3 : s >> 1);
#pragma unroll
          for (int q = 0; q < 8; q++) o[q] = ntab[(~hv[u] >> (4 * q)) & 0xFu];   // negated: a set bit becomes -1
        }
      }
    }
  }
  }
  for (int q = threadIdx.x; q < PITCH; q += 64 * WAVES) aimg[nW * PITCH + q] = 0u;  // the row the last tile reads beyond the last window
  __syncthreads();

  const int lane = (int)(threadIdx.x & 63);
  uint32_t *queue = queues + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) * kM2Queue;
  const int last_j = m - W;                      // valid destination positions of a window's first row: 1 .. m - W
  if (nW <= 0 || last_j < 1) return;
  const int r = lane & 31, h = lane >> 5;
  const int t = (int)min(threshold, 15u);
  // The A image holds the NEGATED head hashes (bit set: -1), so an accumulator holds 2 (d + d') - 64 + preset over its two
  // rows; with preset = 63 - 4 t it is NEGATIVE exactly where d + d' <= 2 t: a set sign bit = "look here", and the flag
  // words need no inversion.  (The preset lives in 16 registers.  As a literal it can be the instruction's inline constant
  // -- 63 - 4 t lies in 3 .. 63 -- but the compiler folds only a literal with a single use, and with several call sites
  // of the products it built the constant in registers per tile instead: measured, dropped.)
  const int preset = 63 - 4 * t;
  const int row_tiles = (nW + 31) / 32;
  const int col_blocks = last_j / 32 + 1;        // positions 0 .. last_j
  const int units = (col_blocks + CB - 1) / CB;
  const int b_in_group = (int)(blockIdx.x - (uint32_t)group * (uint32_t)splits);

  // Exact resolution of one window whose W cells all match on diagonal d, by the whole wave: the same function of the same
  // cells as the vector form's resolve(), read differently.  The source sequence lies in global memory and a trip there
  // costs a wave microseconds, so one trip fetches two rows per lane in EACH direction: the 128 rows behind the window
  // reach the end of the next aligned window whenever P <= 120 (the usual case: one trip decides whether the run goes on
  // into the next window, which then reports it), and 128 rows in front of it end most runs that stop here; a run that
  // goes further back is followed 256 rows per trip.  (Measured and dropped: collecting such windows and resolving four
  // per trip -- the arrays of the four went to scratch memory and the kernel took 1.4 x as long.)
  auto resolve = [&](const int w0, const int d, const int g) {  // (wave-uniform arguments)
    const uint32_t *__restrict__ sp = hashes + __builtin_amdgcn_readfirstlane(ctl[10 + g]);
    const int ns = (int)__builtin_amdgcn_readfirstlane(ctl[18 + g]);
    const int ilo = d < 0 ? 1 - d : 1;
    const int ihi = min(ns - 1, m - 1 - d);
    if (w0 < ilo || w0 + W - 1 > ihi) return;
    const int fwd_limit = min(ihi, w0 + P + W - 1);  // last row of the NEXT aligned window
    // Does row `row` of the diagonal mismatch?  (false outside `in`)  The load is UNCONDITIONAL (of row w0 outside `in`) and
    // the result combined without a branch: written as `in && ...` every row's load sat in a branch of its own behind
    // the wait for the previous one -- four trips to global memory instead of one.
    auto cell_at = [&](const int row, const bool in) {
      const int rr = in ? row : w0;
      return sp[rr] ^ ldst[rr + d];
    };
    auto bad = [&](const uint32_t x, const bool in) { return (int)in & (int)((uint32_t)__popc(x) > threshold); };
    int e = w0 + W;
    bool ended = false;
    int a = -1;                                   // first row of the run, once known
    int q = w0 - 1;                               // next row to look at going back
    {
      const int f0 = e + lane, f1 = e + 64 + lane, b0 = q - lane, b1 = q - 64 - lane;
      const uint32_t cf0 = cell_at(f0, f0 <= fwd_limit), cf1 = cell_at(f1, f1 <= fwd_limit);
      const uint32_t cb0 = cell_at(b0, b0 >= ilo), cb1 = cell_at(b1, b1 >= ilo);
      const unsigned long long mf0 = __builtin_amdgcn_ballot_w64(bad(cf0, f0 <= fwd_limit) != 0);
      const unsigned long long mf1 = __builtin_amdgcn_ballot_w64(bad(cf1, f1 <= fwd_limit) != 0);
      const unsigned long long mb0 = __builtin_amdgcn_ballot_w64(bad(cb0, b0 >= ilo) != 0);
      const unsigned long long mb1 = __builtin_amdgcn_ballot_w64(bad(cb1, b1 >= ilo) != 0);
      if (mf0) {
        e += __ffsll((long long)mf0) - 1;
        ended = true;
      } else if (mf1) {
        e += 64 + __ffsll((long long)mf1) - 1;
        ended = true;
      } else {
        e += 128;
      }
      if (mb0) a = q - (__ffsll((long long)mb0) - 1) + 1;
      else if (mb1) a = q - 64 - (__ffsll((long long)mb1) - 1) + 1;
      else if (q - 128 < ilo) a = ilo;            // every row down to the first one of the diagonal matches
      q -= 128;
    }
    while (!ended && e <= fwd_limit) {            // (only when P > 120)
      const int row = e + lane;
      const unsigned long long mm = __builtin_amdgcn_ballot_w64(bad(cell_at(row, row <= fwd_limit), row <= fwd_limit) != 0);
      if (mm) {
        e += __ffsll((long long)mm) - 1;
        ended = true;
        break;
      }
      e += 64;
    }
    if (!ended) {
      if (fwd_limit == w0 + P + W - 1) return;   // the run also covers the next window: that one reports it
      e = ihi + 1;                                // the run reaches the table edge (comparator.rs:197)
    }
    const int b = e - 1;
    while (a < 0) {                               // rows q, q - 1, ... still to be looked at, 256 per trip
      unsigned long long mm[4];
      uint32_t cell[4];
#pragma unroll
      for (int i = 0; i < 4; i++) cell[i] = cell_at(q - 64 * i - lane, q - 64 * i - lane >= ilo);
#pragma unroll
      for (int i = 0; i < 4; i++) mm[i] = __builtin_amdgcn_ballot_w64(bad(cell[i], q - 64 * i - lane >= ilo) != 0);
      int hit = -1;
#pragma unroll
      for (int i = 3; i >= 0; i--)
        if (mm[i]) hit = i;
      if (hit >= 0) {
        const unsigned long long mh = hit == 0 ? mm[0] : hit == 1 ? mm[1] : hit == 2 ? mm[2] : mm[3];
        a = q - 64 * hit - (__ffsll((long long)mh) - 1) + 1;
      } else if (q - 256 < ilo) {
        a = ilo;
      }
      q -= 256;
    }
    const int len = b - a + 1;
    if (len >= min_len && lane == 0) {
      const uint32_t slot = atomicAdd(count, 1u);
      if (slot < capacity)
        runs[slot] = NeedleHipRun{(uint32_t)(lo + g), (uint32_t)b, (uint32_t)(b + d), (uint32_t)len, 0u, 0u};
    }
  };

  // What the flags point at.  An ITEM = (group of four windows, destination position) that may hold a survivor.  Items
  // are handed out one per lane, 64 at a time, whatever lane flagged them (a destination position that looks like many
  // windows -- a sustained sound -- flags the same lane again and again: left to that lane, the wave would wait for it).
  // A lane tests its item's four windows' head rows EXACTLY with popcounts out of LDS (the vector form's test), then the
  // tail rows of what passes, then kM2Probe rows on either side; what remains is resolved by the wave.
  const uint32_t bias = 31u - (uint32_t)t;        // popcount + bias has bit 5 set exactly when the cell does NOT match
  auto process = [&](const int first, const int cnt) {
    wave_lds_fence_search();
    uint32_t passm = 0u;                          // windows of the group (bit i) whose four head cells all match
    int kbase = 0, j = 0;
    if (lane < cnt) {
      const uint32_t item = queue[first + lane];
      kbase = (int)(item & 0xFFFFu);
      j = (int)(item >> 16);
      const uint32_t d0 = ldst[j], d2 = ldst[j + 2], d4 = ldst[j + 4], d7 = ldst[j + 7];
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int k = min(kbase + i, nW - 1);
        const mfma_v4i lo4 = *reinterpret_cast<const mfma_v4i *>(wsrc + NR * k + E);       // rows 0 .. 3 of the window
        const mfma_v4i hi4 = *reinterpret_cast<const mfma_v4i *>(wsrc + NR * k + E + 4);   // rows 4 .. 7
        const uint32_t s0 = (uint32_t)lo4[0], s2 = (uint32_t)lo4[2], s4 = (uint32_t)hi4[0], s7 = (uint32_t)hi4[3];
        const uint32_t miss = ((uint32_t)__popc(s0 ^ d0) + bias) | ((uint32_t)__popc(s2 ^ d2) + bias) |
                              ((uint32_t)__popc(s4 ^ d4) + bias) | ((uint32_t)__popc(s7 ^ d7) + bias);
        passm |= (miss < 32u && kbase + i < nW) ? 1u << i : 0u;
      }
    }
#if NEEDLE_M2_LAB & 2   // laboratory: head survivors dropped (threshold < 2^20: the mask is zero, but not for the compiler)
    passm &= threshold >> 20;
#endif
    if (__builtin_amdgcn_ballot_w64(passm != 0u) == 0ull) return;
    // the rest of the window, its neighbourhood, and the resolution of what stays: window by window (rare)
#pragma unroll 1
    for (int i = 0; i < 4; i++) {
      bool whole = false;
      int w0 = 0, d = 0, g = 0;
      if ((passm >> i) & 1u) {
        const int k = kbase + i;
        const uint32_t wm = aimg[k * PITCH + 8 * H];
        g = (int)(wm >> 28);
        w0 = (int)(wm & 0x0FFFFFFFu);
        d = j - w0;
        const uint32_t *rows = wsrc + NR * k;     // the window's 16 rows w0 - E .. w0 + W + E - 1
        const uint32_t miss = ((uint32_t)__popc(rows[E + 1] ^ ldst[j + 1]) + bias) | ((uint32_t)__popc(rows[E + 3] ^ ldst[j + 3]) + bias) |
                              ((uint32_t)__popc(rows[E + 5] ^ ldst[j + 5]) + bias) | ((uint32_t)__popc(rows[E + 6] ^ ldst[j + 6]) + bias);
        if (miss < 32u) {
          const int ns = (int)ctl[18 + g];
          const int ilo = d < 0 ? 1 - d : 1;
          const int ihi = min(ns - 1, m - 1 - d);
          if (w0 >= ilo && w0 + W - 1 <= ihi) {   // window inside the table on this diagonal (resolve()'s first test)
            bool f_end = false, b_end = false;    // the run ends inside the probed rows: a mismatch there, or the table's edge
#pragma unroll
            for (int e = 0; e < E; e++) {
              const int fr = w0 + W + e, br = w0 - E + e;
              f_end |= fr > ihi || (uint32_t)__popc(rows[E + W + e] ^ ldst[min(fr, ihi) + d]) > threshold;
              b_end |= br < ilo || (uint32_t)__popc(rows[e] ^ ldst[max(br, ilo) + d]) > threshold;
            }
            whole = !(f_end && b_end);            // ended on both sides: at most W + 2 E - 2 rows, below every min_len of this path
          }
        }
      }
      unsigned long long cand = __builtin_amdgcn_ballot_w64(whole);
      while (cand) {
        const int src_lane = __ffsll((long long)cand) - 1;
        cand &= cand - 1;
        resolve(__builtin_amdgcn_readfirstlane(__shfl(w0, src_lane)), __builtin_amdgcn_readfirstlane(__shfl(d, src_lane)),
                __builtin_amdgcn_readfirstlane(__shfl(g, src_lane)));
      }
    }
  };
  // A batch's flags become items.  flags: bit 4 n - 1 - (4 s + g) set <=> tile slot s (s = 2 (row tile - rt0) + column block,
  // n slots in all), group g.  One item per lane and turn (turns = the most set bits any lane holds), 64 are processed
  // as soon as they are there.
  int qn = 0;                                     // wave-uniform: items waiting in this wave's queue (< 64 between turns)
  auto enqueue = [&](uint32_t flags, const int rt0, const int n_slots, const int j0) {
    for (;;) {
      const bool act = flags != 0u;
      const unsigned long long ball = __builtin_amdgcn_ballot_w64(act);
      if (ball == 0ull) break;
      if (act) {
        const int idx = 4 * n_slots - 1 - (__ffs((int)flags) - 1);
        flags &= flags - 1u;
        const int s = idx >> 2, g = idx & 3;
        const uint32_t kbase = (uint32_t)(32 * (rt0 + (s >> 1)) + 8 * g + 4 * h);
        const uint32_t j = (uint32_t)(j0 + 32 * (s & 1));
        const uint32_t before = __builtin_amdgcn_mbcnt_hi((uint32_t)(ball >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ball, 0u));
        queue[qn + (int)before] = (j << 16) | kbase;
      }
      qn += (int)__popcll(ball);
      if (qn >= 64) {
        process(qn - 64, 64);
        qn -= 64;
      }
    }
  };

  const uint32_t *arow = aimg + r * PITCH + 4 * h;                                         // rows of a full tile: + rt 32 PITCH
  const uint32_t *arow_last = aimg + min(32 * (row_tiles - 1) + r, nW) * PITCH + 4 * h;  // the last tile's may lie beyond the last window
  auto load_a = [&](const int rt, mfma_v4i (&fa)[H]) {
    const uint32_t *ap = rt == row_tiles - 1 ? arow_last : arow + rt * 32 * PITCH;
#pragma unroll
    for (int kb = 0; kb < H; kb++) fa[kb] = *reinterpret_cast<const mfma_v4i *>(ap + 8 * kb);
  };
  mfma_v16i presets;
#pragma unroll
  for (int q = 0; q < 16; q++) presets[q] = preset;
  // rows (0 + 2) accumulate in ua, rows (4 + 7) in ub; the two chains alternate, so a product never waits for its own accumulator
  auto products = [&](const mfma_v4i (&fa)[H], const mfma_v4i (&fbk)[H], mfma_v16i &ua, mfma_v16i &ub) {
    asm volatile("" : "+v"(presets));            // stays in its 16 registers (otherwise re-built from scalars for every tile)
    ua = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[0], fbk[0], presets, 0, 0, 0);
    ub = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[2], fbk[2], presets, 0, 0, 0);
    ua = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[1], fbk[1], ua, 0, 0, 0);
    ub = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[3], fbk[3], ub, 0, 0, 0);
  };
  // a group's word is negative iff one of its four windows passes both sums; its sign goes into the flags
  auto fold = [&](const mfma_v16i &ua, const mfma_v16i &ub, uint32_t &flags) {
#pragma unroll
    for (int g = 0; g < 4; g++) {
      int a = ua[4 * g] & ub[4 * g];
#pragma unroll
      for (int q = 4 * g + 1; q < 4 * g + 4; q++) a = __builtin_amdgcn_bitop3_b32(ua[q], ub[q], a, 0xEA);   // (ua & ub) | a: one instruction
      flags = __builtin_amdgcn_alignbit(flags, (uint32_t)a, 31);  // (written as C the compiler makes it and, and, and, and_or, or3)
    }
  };

#if NEEDLE_M2_LAB & 64   // laboratory: a workgroup's setup alone
  if (threshold < (1u << 20)) return;
#endif
  for (;;) {
    int c = 0;
    if (lane == 0) c = (int)__hip_atomic_fetch_add(&ctl[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    c = __builtin_amdgcn_readfirstlane(c);
    const int unit = b_in_group + splits * c;     // (every wave sees the counter pass `units`: the loop ends for all)
    if (unit >= units) break;
    // B fragments: lane (r, h) holds half h of dst[j + s] for the H head rows s as +-1 bytes, four bits at a time through a
    // table of 16 words (16 banks: different entries never collide, equal ones are one broadcast)
    mfma_v4i fb[CB][H];
    const int j0 = 32 * unit * CB + r;            // this lane's position in the unit's first column block; + 32 in the second
#pragma unroll
    for (int cbk = 0; cbk < CB; cbk++) {
      const int jr = min(j0 + 32 * cbk, m);       // (a unit's second block may lie beyond the table: zeros)
#pragma unroll
      for (int kb = 0; kb < H; kb++) {
        const uint32_t half = ldst[jr + (kb * (W - 1)) / (H - 1)] >> (16 * h);   // head rows 0, 2, 4, 7
#pragma unroll
        for (int q = 0; q < 4; q++) fb[cbk][kb][q] = (int)ntab[(half >> (4 * q)) & 0xFu];
      }
    }
    if (unit == 0 || unit * CB + CB > last_j / 32) {  // (wave-uniform) a unit with positions outside 1 .. last_j: their columns
#pragma unroll                                        // become zeros -- product 0 + preset: positive, never flagged
      for (int cbk = 0; cbk < CB; cbk++) {
        const bool ok = j0 + 32 * cbk >= 1 && j0 + 32 * cbk <= last_j;
#pragma unroll
        for (int kb = 0; kb < H; kb++)
#pragma unroll
          for (int q = 0; q < 4; q++) fb[cbk][kb][q] = ok ? fb[cbk][kb][q] : 0;
      }
    }
    // The unit's tiles in the order (row tile, column block), software-pipelined: a tile's four products are issued before
    // the tile in front of it is folded (two accumulator pairs), and the next row tile's A fragments are asked for behind
    // the last product that reads this row tile's.
    mfma_v4i fa[H];
    uint32_t flags = 0u;                                             // sign bits of the groups' words: set = look here
    load_a(0, fa);
    if constexpr (PER_SIMD <= 3) {
      // Software-pipelined: a tile's four products are issued before the tile in front of it is folded (two accumulator
      // pairs), and the next row tile's A fragments are asked for behind the last product that reads this row tile's.
      // (Measured and dropped: pinning "one product, six instructions of the fold" with __builtin_amdgcn_sched_group_barrier
      // -- 4.69 against 4.48 ms at 79 800 pairs; the compiler copied fragments to make room for the early reads.)
      mfma_v16i a0, b0, a1, b1;
      products(fa, fb[0], a0, b0);                                   // tile (row 0, block 0)
#pragma unroll 1
      for (int rt = 0; rt + 1 < row_tiles; rt++) {                   // every row tile but the last: the next one exists, nothing in
        products(fa, fb[1], a1, b1);                                 // the loop is conditional (a conditional product made the  (rt, 1)
        load_a(rt + 1, fa);                                          // compiler copy both accumulator pairs twice per turn)
        fold(a0, b0, flags);                                         // (rt, 0)
        products(fa, fb[0], a0, b0);                                 // (rt + 1, 0)
        fold(a1, b1, flags);                                         // (rt, 1)
        if ((rt & (kM2Batch - 1)) == kM2Batch - 1) {
#if !(NEEDLE_M2_LAB & 1)   // laboratory: flags ignored -- the tile loop alone
          enqueue(flags, rt - (kM2Batch - 1), 2 * kM2Batch, j0);
#else
          if (flags == 0x12345u + threshold) atomicAdd(count, 1u);
#endif
          flags = 0u;
        }
      }
      products(fa, fb[1], a1, b1);                                   // the last row tile
      fold(a0, b0, flags);
      fold(a1, b1, flags);
    } else {
      // One accumulator pair, and the compiler held to it: left alone it issues the second tile's products inside the first
      // tile's fold on a SECOND pair, and at 128 registers that costs B fragments copied and spilled in every turn.
      mfma_v16i a0, b0;
#pragma unroll 1
      for (int rt = 0; rt + 1 < row_tiles; rt++) {
        products(fa, fb[0], a0, b0);
        fold(a0, b0, flags);
        __builtin_amdgcn_sched_barrier(0);
        products(fa, fb[1], a0, b0);
        load_a(rt + 1, fa);
        fold(a0, b0, flags);
        __builtin_amdgcn_sched_barrier(0);
        if ((rt & (kM2Batch - 1)) == kM2Batch - 1) {
#if !(NEEDLE_M2_LAB & 1)
          enqueue(flags, rt - (kM2Batch - 1), 2 * kM2Batch, j0);
#else
          if (flags == 0x12345u + threshold) atomicAdd(count, 1u);
#endif
          flags = 0u;
        }
      }
      products(fa, fb[0], a0, b0);
      fold(a0, b0, flags);
      products(fa, fb[1], a0, b0);
      fold(a0, b0, flags);
    }
#if !(NEEDLE_M2_LAB & 1)
    enqueue(flags, (row_tiles - 1) & ~(kM2Batch - 1), 2 * (((row_tiles - 1) & (kM2Batch - 1)) + 1), j0);
#else
    if (flags == 0x12345u + threshold) atomicAdd(count, 1u);
#endif
  }
  if (qn > 0) process(0, qn);
}
