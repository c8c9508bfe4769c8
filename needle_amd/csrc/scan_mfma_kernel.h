// The sampled scan's first stage on the matrix pipe (round 5; FP4 products since the second half of that round).  Included
// by search.hip inside its anonymous namespace, after SearchProblem, mfma_windows() and the mfma_v4i / _v8i / _v16f types.
//
// Same aligned windows, same candidates and same runs as hamming_runs_sampled_kernel (comparator.rs:176-200 is what all of
// them reproduce).  The Hamming distances of a window's head rows against every destination position are matrix products:
// a hash as 32 values of +-1 gives dot(a, b) = 32 - 2 d(a, b).  +-1 is exact in FP4 (e2m1: 0x2, 0xA), and gfx950's
// v_mfma_f32_32x32x64_f8f6f4 with both operands FP4 takes K = 64 -- TWO hash rows -- in the 32 cycles that
// v_mfma_i32_32x32x32_i8 takes for one (tools/mfma_fp4_probe.hip: exact sums, 33 - 35 ticks either way).  A tile is 32
// aligned windows x 32 destination positions and costs two instructions (head rows 0 | 2 and 4 | 7 of a window of 8 by lane
// half); fragments are 16 bytes per lane and operand; the f32 accumulators hold small integers exactly.
//
// Round 4's form (int8, one product per head row) kept the matrix pipe 27 % busy: beside its 4 products a tile cost 88
// vector and 16 scalar instructions, a wave runs its instructions in order at 5 - 8 cycles each, and three waves per SIMD
// (166 registers) cannot cover that.  What was measured since (profiles/NOTES.md, round 5): what costs is not arithmetic but
// BRANCHES per tile, trips to global memory, wave-uniform code run by 64 lanes in step, returning atomics on one address, and
// rare paths inlined so often that waves wait for instructions.  Hence:
//  * The products only FLAG.  A tile's two products go into two register sets -- rows (0 + 2) and rows (4 + 7), each
//    preset so that its sign bit is SET where the SUM of two rows' distances is <= 2 t, a necessary condition of "both
//    <= t" (0.20 % of the window-diagonals pass on synthetic audio, 0.05 % the exact head test).  16 + 16 result registers
//    are folded into four words (one per group of four windows) and the four sign bits are shifted into a per-lane flag
//    word (v_alignbit): 20 vector instructions, no compare, no branch.
//  * Every eight tiles the flag words become ITEMS = (group of four windows, destination position), handed out one per
//    lane, 64 at a time, whatever lane flagged them (a position that looks like many windows -- a sustained sound --
//    flags the same lane again and again: left to that lane, the wave waits for it).  A lane tests its item's four
//    windows on all eight rows with popcounts out of LDS.
//  * What passes (a window in thousands) is looked at by the WAVE, window by window, sixteen lanes reading its sixteen
//    rows in LDS (window_whole_wave): whole, and the probes of one side (rows up to half a minimum run away) all matching?  Consecutive whole
//    windows on a diagonal form a CHAIN; only its last window resolves it -- one trip to global memory fetches 512 rows
//    backward and 128 forward, every stretch of >= min_len matching rows in it is a run (resolve()).
//  * Runs go to a buffer in LDS; the workgroup asks for their slots in the run list with one atomic at its end.
//  * (Round 6.)  The kernel is bound by vector-instruction issue (80 % of the SIMDs' cycles, the matrix pipe 34 %): an item
//    enters the queue as a TOKEN naming its flag bit, decoded 64 at a time by process(); and a workgroup's END is shared work:
//    chains found late, and chains found in a crowd (a block of equal hashes), go to a ring in LDS that the waves which are through
//    with their units empty (kM2Chains, drain()).
//  * One workgroup = one destination x up to kM2Members sources (as many as its share of the CU's LDS holds); the sources'
//    windows (A image rows + sixteen hashes each) are built once per launch (m2_window_images_kernel) and copied in; an A
//    fragment read serves both column blocks of a unit; waves take units from a counter in LDS, so none idles while another
//    still has units; a workgroup finds its group with one load (round 4: a binary search of the table, 17 dependent loads).
//  * Default shape (mfma_waves(), search.hip): workgroups of 8 waves, two per CU, four waves per SIMD, ~100 registers, one
//    accumulator pair, a tile folded before the next is multiplied.  Measured against it: 16 waves x 1 (3.2 against 2.8 ms
//    at 79 800 pairs of 45-minute windows), 12 x 1 and 4 x 3 (slower still).
//  * Columns outside the table get all-zero B fragments and rows beyond the last window read a row of zeros
//    (product 0 + preset > 0: never flagged).
// ONE call site of enqueue(), process() and resolve() each (always_inline; at three sites each the kernel was 7500 lines of
// ISA and every item-path stage 40 - 70 % slower).  Needs t <= 15.
#ifndef NEEDLE_M2_LAB
#define NEEDLE_M2_LAB 0                       // timing laboratory (tools/build_variant.sh), WRONG results: 1 flags ignored (the tile loop
                                              // alone), 2 the items' survivors dropped, 4 whole windows not resolved, 8 event counts, 16 a resolution's trip and no more, 32 a walk ends at its first mismatch, 64 a workgroup's setup alone
#endif
#if NEEDLE_M2_LAB & 8   // laboratory: event counts (search.hip prints them after the launch)
__device__ unsigned long long m2_dbg[8];
#define M2_COUNT(i) do { if (lane == 0) atomicAdd(&m2_dbg[i], 1ull); } while (0)
#define M2_COUNT_LANES(i, n) atomicAdd(&m2_dbg[i], (unsigned long long)(n))
#else
#define M2_COUNT(i) do { } while (0)
#define M2_COUNT_LANES(i, n) do { } while (0)
#endif
constexpr int kM2Heads = 4;
#ifndef NEEDLE_M2_MEMBERS
#define NEEDLE_M2_MEMBERS 12
#endif
constexpr int kM2Members = NEEDLE_M2_MEMBERS; // sources a workgroup takes at most (of one destination; its share of the CU's LDS decides: nine 24-minute
                                              // windows, four 45-minute ones); < 16: a window's member rides in 4 bits.  (8 until round 6: the ninth -- room
                                              // made by a run buffer of 32 instead of 64 -- is 2.2 % at 39 060 pairs of 24-minute windows: one more pair per setup,
                                              // 98 instead of 94 % of the row tiles' rows in use)
constexpr int kM2Pitch = kM2Heads * 4 + 4;    // words of a window's row in the A image: 4 x 32 FP4 nibbles of +-1, the window's member << 28 | w0, 3 spare
                                              // (20: sixteen lanes' 16-byte reads of one instruction fall into sixteen different groups of four banks)
#ifndef NEEDLE_M2_RUNBUF
#define NEEDLE_M2_RUNBUF 32
#endif
constexpr int kM2RunBuf = NEEDLE_M2_RUNBUF;   // runs a workgroup collects in LDS before it asks for room in the run list (one atomic)
#ifndef NEEDLE_M2_OVERFLOW
#define NEEDLE_M2_OVERFLOW 16                  // (0: measurements -- one atomic per run beyond the workgroup's buffer, as before round 6)
#endif
constexpr int kM2Overflow = NEEDLE_M2_OVERFLOW;  // ... and beyond those, runs a WAVE collects before it asks (round 6: stretches of one repeated hash --
                                              // silence against silence -- give a workgroup thousands of runs; one returning atomic per run on the
                                              // list's one counter was 46 of 52 ms at 39 060 pairs of the hostile corpus, 4.65 M runs).  They lie in the
                                              // 64 words of the wave's item queue whose items process() has just taken: no LDS of their own (a buffer of
                                              // 4 KB per workgroup cost 45-minute windows a source per workgroup and the scan 6 %)
constexpr int kM2Table = 256;                 // a byte of hash bits -> its eight FP4 nibbles (bit set: +1 = 0x2, clear: -1 = 0xA)
constexpr int kM2Probe = 4;                   // rows looked at on either side of a whole window
constexpr int kM2Rows = kSampleW + 2 * kM2Probe;  // source hashes kept per window: its W rows and four probes on either side (m2_probe_offset)
constexpr int kM2ColBlocks = 2;               // column blocks of 32 positions a wave takes per unit
constexpr int kM2Batch = 4;                   // row tiles (x 2 column blocks = 8 tiles = 32 flag bits) between two looks at the flags
constexpr int kM2Queue = 128;                 // items a wave can hold: < 64 waiting + the <= 64 one turn adds
// The workgroup's control words (one table in LDS)
constexpr int kCtlUnits = 0;                              // units handed out
constexpr int kCtlRows = 1;                               // [+ g] first row (window) of member g; [+ members]: all rows
constexpr int kCtlSrc = kCtlRows + kM2Members + 1;        // [+ g] member g's source offset in the hash arena
constexpr int kCtlLen = kCtlSrc + kM2Members;             // [+ g] its length
constexpr int kCtlImage = kCtlLen + kM2Members;           // [+ g] first window of its image
constexpr int kCtlRuns = kCtlImage + kM2Members;          // runs collected in the workgroup's buffer
constexpr int kCtlSlot = kCtlRuns + 1;                    // their first slot in the run list
constexpr int kCtlTaken = kCtlSlot + 1;                   // chain queue: chains taken,
constexpr int kCtlPushed = kCtlTaken + 1;                 // chains pushed,
constexpr int kCtlLock = kCtlPushed + 1;                  // its lock
constexpr int kCtlThrough = kCtlLock + 1;                 // waves that have flushed their items
constexpr int kM2CtlWords = (kCtlThrough + 1 + 3) & ~3;   // (a multiple of 4: what follows stays 16-byte aligned)
// The workgroup's CHAIN QUEUE (round 6).  The chains of a destination's intro against its eight or nine sources end within one
// or two column units: ONE wave found them all and resolved them one after the other, a trip to global memory each.  While the
// workgroup has units to hand out that costs nothing (the other waves take them); near its end the other seven ran out of
// units and waited -- a third of the resolution's 0.24 of 0.74 ms at 39 060 pairs of 24-minute windows.  So a chain found
// within the last kM2LateUnits units per wave goes to a queue in LDS (a ring behind a lock), which every wave empties at its
// end (drain()); one found earlier, or finding the queue full, is resolved by its finder as before.  0.738 -> 0.648 ms there;
// 79 800 pairs of 45-minute windows unchanged inside +-1 % (the same with the queue drained between units, with every chain
// queued, with 64 entries).  And a wave in a CROWD of chains (a block of equal hashes: silence against silence) hands them
// over too, once it has resolved kM2Crowd of a process() call itself: the waves whose units were cheap are through by then and
// stay in drain() until every wave has flushed its items.
#ifndef NEEDLE_M2_LATE_UNITS
#define NEEDLE_M2_LATE_UNITS 2
#endif
constexpr int kM2LateUnits = NEEDLE_M2_LATE_UNITS;
#ifndef NEEDLE_M2_CHAINS
#define NEEDLE_M2_CHAINS 32
#endif
constexpr int kM2Chains = NEEDLE_M2_CHAINS;   // entries of the ring
static_assert((kM2Chains & (kM2Chains - 1)) == 0, "");
#ifndef NEEDLE_M2_CROWD
#define NEEDLE_M2_CROWD 4
#endif
constexpr int kM2Crowd = NEEDLE_M2_CROWD;     // chains one process() call resolves itself before it hands the rest to the queue
static_assert((kM2Batch & (kM2Batch - 1)) == 0 && 8 * kM2Batch <= 32, "a batch's flags fill at most one word");
static_assert(kM2Members >= 1 && kM2Members <= 15, "");
static_assert(kM2Probe == 4 && kM2Rows == 16, "a window's sixteen source hashes are read as 16-byte words");
// Which rows the PROBES of a window are (round 6): slot s = -kM2Probe .. -1 below the window, kSampleW .. kSampleW + kM2Probe - 1 above it,
// as an offset from the window's first row.  A run of min_len rows or more that holds the window's W rows has min_len - W others, so at
// least REACH = ceil((min_len - W) / 2) of them on one side: the probes of a side lie at distances 1 .. REACH from the window, evenly,
// and "all probes of one side match" stays a NECESSARY condition of such a run -- but one that a run of a dozen rows no longer meets
// (the four rows next to the window, as until now, match or fail together on audio; a window with a run of 9 - 14 rows around it cost a
// trip to global memory to find that out: a quarter of all resolutions at 39 060 pairs).
__host__ __device__ constexpr int m2_probe_offset(int s, int min_len) {
  const int reach = (min_len - kSampleW + 1) / 2;
  const int i = s < 0 ? -1 - s : s - kSampleW;                    // 0 .. kM2Probe - 1, the nearest first
  const int dist = 1 + i * (reach - 1) / (kM2Probe - 1);
  return s < 0 ? -dist : kSampleW - 1 + dist;
}
__host__ __device__ constexpr int m2_row_offset(int s, int min_len) { return s >= 0 && s < kSampleW ? s : m2_probe_offset(s, min_len); }
static_assert(m2_probe_offset(-1, 23) == -1 && m2_probe_offset(-4, 23) == -8 && m2_probe_offset(8, 67) == 8 && m2_probe_offset(11, 67) == 37, "");

// LDS words of a workgroup: staged destination (+ 64 zeros), tables, per-window source hashes, A image
__host__ __device__ constexpr size_t m2_round4(size_t x) { return (x + 3) & ~(size_t)3; }
__host__ __device__ constexpr size_t m2_lds_words(uint64_t m, uint64_t windows, int waves) {
  return m2_round4(m + 64) + kM2CtlWords + kM2Table + 4 * kM2RunBuf + 2 * kM2Chains + (size_t)waves * kM2Queue + (size_t)(windows + 1) * (kM2Pitch + kM2Rows);
}

// A source sequence's windows as a workgroup wants them in LDS -- per window kM2Pitch words of the A image (its four head
// hashes as 32 negated +-1 nibbles each, then w0) and kM2Rows source hashes around it -- built ONCE per launch in global
// memory (m2_window_images_kernel) instead of by every workgroup that takes the sequence as a source: at 2000 videos a
// sequence is a source 1999 times, and gathering + expanding its windows was a tenth of the scan (a workgroup's setup
// alone: 0.44 of 4.0 ms at 79 800 pairs).  A workgroup copies its members' images with 16-byte loads.
constexpr int kM2ImageWords = kM2Pitch + kM2Rows;   // per window
// the eight low bits of x as FP4 (e2m1) nibbles: bit i set -> +1.0 (0x2), clear -> -1.0 (0xA), in nibble i
__host__ __device__ constexpr uint32_t m2_nibbles(uint32_t x) {
  uint32_t w = 0;
  for (int i = 0; i < 8; i++) w |= (((x >> i) & 1u) ? 0x2u : 0xAu) << (4 * i);
  return w;
}
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
    const int row = (int)w0 + m2_row_offset(s, (int)min_len);
    const uint32_t hsh = row >= 0 && row < (int)q.n ? hashes[q.src_off + (uint32_t)row] : 0u;
    uint32_t *img = images + (size_t)k * kM2ImageWords;
    img[PITCH + s + E] = hsh;
    if (s == 0) img[4 * H] = w0;
    if (s == 1 || s == 3 || s == 5) img[4 * H + (s + 1) / 2] = 0u;   // (the row's three spare words)
    if (s == 0 || s == 2 || s == 4 || s == 7) {
      uint32_t *o = img + 4 * (s == 7 ? 3 : s >> 1);
#pragma unroll
      for (int b = 0; b < 4; b++) o[b] = m2_nibbles(~hsh >> (8 * b));   // negated: a set bit becomes -1
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
  static_assert(W == 8, "head rows {0, 2, 4, 7} of a window of 8: two pairs for two products");
  constexpr int H = kM2Heads, HP = H / 2, PITCH = kM2Pitch, STRIDE = kM2ImageWords, CB = kM2ColBlocks, E = kM2Probe, NR = kM2Rows;
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
  uint32_t *ntab = ctl + kM2CtlWords;            // 8 bits -> 8 nibbles of +-1
  uint32_t *runbuf = ntab + kM2Table;            // the workgroup's runs: (pair, last row, last column, length)
  uint32_t *chains = runbuf + 4 * kM2RunBuf;     // the workgroup's chains waiting for a wave: (member << 28 | w0, diagonal); word 0 == 0: not written yet
  uint32_t *queues = chains + 2 * kM2Chains;     // per wave: items waiting for their exact test
  // per window the source hashes of its W rows and of its 2 E probe rows (m2_row_offset; zero where there is none)
  // Per window ONE row of kM2ImageWords = 36 words: its A image row (the head hashes' nibbles, then w0), then its sixteen
  // source hashes.  36 = 4 x 9: sixteen lanes' 16-byte reads of sixteen different windows -- the A fragments of a tile, the
  // rows of the items' windows -- fall into sixteen different groups of four banks.  (Two arrays, the hashes at a pitch of
  // 16 words: lanes with different windows met in FOUR groups; SQ_LDS_BANK_CONFLICT was 61 % of SQ_LDS_IDX_ACTIVE.)
  uint32_t *wimg = queues + WAVES * kM2Queue;    // 16-byte aligned: every size above is a multiple of 4 words
  static_assert(4 * kM2Overflow <= 64 && kM2Queue >= 128, "the overflow runs lie in the queue words of the 64 items a process() call has taken");
  uint32_t *overflow = nullptr;                  // this wave's overflow runs: set by process() (below)
  int overflowed = 0;                            // runs in it (wave-uniform)

  if (threadIdx.x == 0) {
    ctl[kCtlUnits] = 0u;
    ctl[kCtlRuns] = 0u;
    ctl[kCtlTaken] = 0u;
    ctl[kCtlPushed] = 0u;
    ctl[kCtlLock] = 0u;
    ctl[kCtlThrough] = 0u;
    int rows = 0;
    for (int g = 0; g < members; g++) {
      ctl[kCtlRows + g] = (uint32_t)rows;
      ctl[kCtlSrc + g] = problems[lo + g].src_off;
      ctl[kCtlLen + g] = problems[lo + g].n;
      ctl[kCtlImage + g] = problems[lo + g].block_base;
      rows += mfma_windows((int)problems[lo + g].n, min_len, W);
    }
    ctl[kCtlRows + members] = (uint32_t)rows;
  }
  static_assert(64 * WAVES >= kM2Table, "");
  if (threadIdx.x < kM2Table) ntab[threadIdx.x] = m2_nibbles(threadIdx.x);
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
        for (int i = 1; i < members; i++) g += (uint32_t)k >= ctl[kCtlRows + i] ? 1 : 0;
        const uint32_t from = ctl[kCtlImage + g] + ((uint32_t)k - ctl[kCtlRows + g]);
        v[u] = *reinterpret_cast<const mfma_v4i *>(images + (size_t)from * kM2ImageWords + 4 * piece);
        if (piece == PITCH / 4 - 1) v[u][0] |= (int)((uint32_t)g << 28);   // w0 -> member << 28 | w0
        to[u] = wimg + k * STRIDE + 4 * piece;
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
        for (int i = 1; i < members; i++) g += (uint32_t)k >= ctl[kCtlRows + i] ? 1 : 0;
        const int w0 = 1 + (k - (int)ctl[kCtlRows + g]) * P;
        const int row = w0 + m2_row_offset(s, min_len);
        wm[u] = ((uint32_t)g << 28) | (uint32_t)w0;
        hv[u] = row >= 0 && row < (int)ctl[kCtlLen + g] ? hashes[ctl[kCtlSrc + g] + (uint32_t)row] : 0u;
      }
#pragma unroll
      for (int u = 0; u < kU; u++) {
        const int k = k0 + u * kStep;
        if (k >= nW) break;
        wimg[k * STRIDE + PITCH + s + E] = hv[u];
        if (s == 0) wimg[k * STRIDE + 4 * H] = wm[u];
        if (s == 0 || s == 2 || s == 4 || s == 7) {                  // head rows 0 .. 3
          uint32_t *o = wimg + k * STRIDE + 4 * (s == 7 ? 3 : s >> 1);
#pragma unroll
          for (int q = 0; q < 4; q++) o[q] = ntab[(~hv[u] >> (8 * q)) & 0xFFu];   // negated: a set bit becomes -1
        }
      }
    }
  }
  }
  for (int q = threadIdx.x; q < PITCH; q += 64 * WAVES) wimg[nW * STRIDE + q] = 0u;  // the row the last tile reads beyond the last window
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

  const uint32_t p_magic = (uint32_t)(0xFFFFFFFFull / (uint32_t)P);   // ~ 2^32 / P from below (P >= W: the host takes this path from min_len >= 2 W - 1)
  const uint32_t bias = 31u - (uint32_t)t;        // popcount + bias has bit 5 set exactly when the cell does NOT match
  // Is window k (member g, first row w0) WHOLE at destination position j: inside the table on its diagonal, its W cells all
  // match, and the probes of at least ONE side all match (m2_probe_offset: a run of min_len rows that holds the window reaches
  // the farthest probe of one side; any deterministic predicate with that property serves the chain rule)?  All from LDS, asked by
  // the whole wave about ONE window (wave-uniform arguments): lane s < 16 looks at the image row's s-th hash.  One predicate for
  // three askers who must agree: a window about itself, about its successor, and the walk below about the windows it passes.  (Run per lane by 64 lanes in step it was 0.33 of the scan's
  // 1.0 ms at 39 060 pairs.)
  const int whole_row_off = m2_row_offset((lane & (NR - 1)) - E, min_len);
  auto window_whole_wave = [&](const int k, const int g, const int w0, const int j) __attribute__((always_inline)) {
    const int d = j - w0;
    const int ns = (int)__builtin_amdgcn_readfirstlane(ctl[kCtlLen + g]);
    const int ilo = d < 0 ? 1 - d : 1;
    const int ihi = min(ns - 1, m - 1 - d);
    const int row = w0 + whole_row_off;        // lane s < 16: the window's rows, and its probes (m2_probe_offset)
    const bool in = row >= ilo && row <= ihi;
    const bool is_bad = !in || (uint32_t)__popc(wimg[k * STRIDE + PITCH + (lane & (NR - 1))] ^ ldst[min(max(row, ilo), ihi) + d]) > threshold;
    const uint32_t mask = (uint32_t)__builtin_amdgcn_ballot_w64(is_bad) & 0xFFFFu;
    constexpr uint32_t kWindow = ((1u << W) - 1u) << E, kBelow = (1u << E) - 1u, kAbove = kBelow << (E + W);
    return (mask & kWindow) == 0u && !((mask & kBelow) != 0u && (mask & kAbove) != 0u);
  };

  // the wave's overflow buffer into the run list: one request for room, one run per lane
  auto flush_overflow = [&]() __attribute__((always_inline)) {
    if (overflowed == 0) return;
    wave_lds_fence_search();
    uint32_t base = 0u;
    if (lane == 0) base = atomicAdd(count, (uint32_t)overflowed);
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    if (lane < overflowed && base + (uint32_t)lane < capacity) {
      const uint32_t *e = overflow + 4 * lane;
      runs[base + (uint32_t)lane] = NeedleHipRun{e[0], e[1], e[2], e[3], 0u, 0u};
    }
    wave_lds_fence_search();
    overflowed = 0;
  };

  // Exact resolution, by the whole wave, of a CHAIN: consecutive aligned windows of one member, all whole on diagonal d.
  // Only the chain's LAST window (the one whose successor is not whole) comes here; the others do nothing -- they used to
  // spend a trip to global memory each on finding out that the run goes on into the next window, and with two runs of
  // five windows per pair those trips were 0.43 of 0.97 ms at 39 060 pairs of 24-minute windows.  From the last window the
  // wave follows the diagonal forward to the run's end, then backward: every maximal stretch of matching cells of at
  // least min_len rows is a run (the same function of the same cells as the table walk, comparator.rs:178-236).  At a
  // mismatch in row x the walk is over -- unless x lies in the GAP above an aligned window that is whole: that window
  // has left its run to its successor's chain, that is to this walk, which goes on below x.  (A mismatch INSIDE an aligned
  // window: the window is not whole, so the one below it is the last of a chain of its own.)
  // A trip fetches eight rows per lane backward (and the first one two forward): a 90-second run is two trips.
  auto resolve = [&](const int w0, const int d, const int g) __attribute__((always_inline)) {  // (wave-uniform arguments)
    const uint32_t *__restrict__ sp = hashes + __builtin_amdgcn_readfirstlane(ctl[kCtlSrc + g]);
    const int ns = (int)__builtin_amdgcn_readfirstlane(ctl[kCtlLen + g]);
    const int kg = (int)__builtin_amdgcn_readfirstlane(ctl[kCtlRows + g]);   // the member's first window
    const int ilo = d < 0 ? 1 - d : 1;
    const int ihi = min(ns - 1, m - 1 - d);
    if (w0 < ilo || w0 + W - 1 > ihi) return;
    M2_COUNT(0);
#if NEEDLE_M2_LAB & 4    // laboratory: whole windows found, none resolved
    if (threshold < (1u << 20)) return;
#endif
    // The cell of row `row` (of row w0 outside `in`): the load is UNCONDITIONAL and the result combined without a branch.
    // Written as `in && ...` every row's load sat in a branch of its own behind the wait for the previous one -- a trip each.
    auto cell_at = [&](const int row, const bool in) {
      const int rr = in ? row : w0;
      return sp[(uint32_t)rr] ^ ldst[rr + d];     // (unsigned: one address register per load beside the scalar base)
    };
    auto bad = [&](const uint32_t x, const bool in) { return ((int)in & (int)((uint32_t)__popc(x) > threshold)) != 0; };
    // A run goes to the workgroup's buffer in LDS; the buffer asks for its slots of the run list with ONE atomic at the end
    // (returning atomics on one address go one after the other: at 39 060 pairs the 79 027 of them WERE the resolution's
    // 0.43 ms, whatever the walk did).  A workgroup with more runs than the buffer holds emits the rest directly.
    // (wave-uniform arguments.)  The workgroup's buffer first; when that is full, the wave's own, emptied into the run list
    // kM2Overflow runs at a time with one atomic.
    auto emit = [&](const int a, const int b) {
      if (b - a + 1 < min_len) return;
      uint32_t at = 0u;
      if (lane == 0) at = __hip_atomic_fetch_add(&ctl[kCtlRuns], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      at = (uint32_t)__builtin_amdgcn_readfirstlane((int)at);
      if (at < (uint32_t)kM2RunBuf) {
        if (lane == 0)
          runbuf[4 * at] = (uint32_t)(lo + g), runbuf[4 * at + 1] = (uint32_t)b, runbuf[4 * at + 2] = (uint32_t)(b + d),
          runbuf[4 * at + 3] = (uint32_t)(b - a + 1);
        return;
      }
      if (kM2Overflow == 0) {
        if (lane == 0) {
          const uint32_t slot = atomicAdd(count, 1u);
          if (slot < capacity) runs[slot] = NeedleHipRun{(uint32_t)(lo + g), (uint32_t)b, (uint32_t)(b + d), (uint32_t)(b - a + 1), 0u, 0u};
        }
        return;
      }
      if (lane == 0)
        overflow[4 * overflowed] = (uint32_t)(lo + g), overflow[4 * overflowed + 1] = (uint32_t)b, overflow[4 * overflowed + 2] = (uint32_t)(b + d),
        overflow[4 * overflowed + 3] = (uint32_t)(b - a + 1);
      if (++overflowed == kM2Overflow) {
        flush_overflow();
      }
    };
    constexpr int kBack = 8;                      // rows per lane of a backward trip
    int e = w0 + W;                               // forward: first row not known to match
    int q = w0 - 1;                               // backward: next row to look at
    int b = 0;                                    // last row of the stretch the backward walk is in
    // Rows q - 64 i - lane, i < kBack, have been looked at (bit i of a lane's badmask: its row of block i mismatches); true
    // when the walk is over.  Every mismatch of the block is handled from these bits (audio-like runs are ragged: a row over
    // the threshold every hundred rows or so).  One ballot at a time: eight live masks cost the kernel its scalar registers
    // (spilled to lanes all over the item path: the head tests took 0.13 instead of 0.08 ms).
    auto back_step = [&](const uint32_t badmask) __attribute__((always_inline)) {
      const int q0 = q;
      int pos = q0;                               // rows above pos are done
      for (;;) {
        int x = -1;                               // the mismatch nearest to pos
#pragma unroll 1
        for (int i = 0; i < kBack && x < 0; i++) {
          const unsigned long long mm = __builtin_amdgcn_ballot_w64(((badmask >> i) & 1u) != 0u && q0 - 64 * i - lane <= pos);
          if (mm) x = q0 - 64 * i - (__ffsll((long long)mm) - 1);
        }
        if (x < 0) {
          if (q0 - 64 * kBack < ilo) {            // every row down to the first one of the diagonal matches
            emit(ilo, b);
            return true;
          }
          q = q0 - 64 * kBack;
          return false;
        }
        emit(x + 1, b);
        M2_COUNT(2);
        int idx = (int)__umulhi((uint32_t)(x - 1), p_magic);   // (x - 1) / P: the aligned window that starts at or below x
        idx += ((idx + 1) * P <= x - 1) - (idx * P > x - 1);   // (x >= ilo >= 1; the reciprocal is off by one at most)
        const int ws = 1 + idx * P;
        if (x <= ws + W - 1) return true;         // x inside it
#if NEEDLE_M2_LAB & 32   // laboratory: the walk ends at its first mismatch
        if (threshold < (1u << 20)) return true;
#endif
        if (!window_whole_wave(kg + idx, g, ws, ws + d)) return true;
        b = x - 1;                                // x in the gap above a whole window: its run is this walk's
        pos = x - 1;
        M2_COUNT(3);
      }
    };
    // A trip: kBack rows per lane backward -- and on the first one two forward, which end the run unless it goes on for
    // more than 128 rows behind its chain's last window (min_len > 128, or the member's last window).
    bool over = false, first = true, ended = false;
#pragma unroll 1
    while (!over) {
      uint32_t cf[2], cb[kBack];
      M2_COUNT(1);
#pragma unroll
      for (int i = 0; i < 2; i++) cf[i] = cell_at(e + 64 * i + lane, first && e + 64 * i + lane <= ihi);
#pragma unroll
      for (int i = 0; i < kBack; i++) cb[i] = cell_at(q - 64 * i - lane, q - 64 * i - lane >= ilo);
      if (first) {
        const unsigned long long mf0 = __builtin_amdgcn_ballot_w64(bad(cf[0], e + lane <= ihi));
        const unsigned long long mf1 = __builtin_amdgcn_ballot_w64(bad(cf[1], e + 64 + lane <= ihi));
        ended = true;
        if (mf0) e += __ffsll((long long)mf0) - 1;
        else if (mf1) e += 64 + __ffsll((long long)mf1) - 1;
        else e += 128, ended = false;
#pragma unroll 1
        while (!ended && e <= ihi) {
          uint32_t c[4];
          M2_COUNT(4);
#pragma unroll
          for (int i = 0; i < 4; i++) c[i] = cell_at(e + 64 * i + lane, e + 64 * i + lane <= ihi);
          int step = 256;
#pragma unroll
          for (int i = 3; i >= 0; i--) {
            const unsigned long long mf = __builtin_amdgcn_ballot_w64(bad(c[i], e + 64 * i + lane <= ihi));
            if (mf) step = 64 * i + __ffsll((long long)mf) - 1, ended = true;
          }
          e += step;
        }
        if (!ended) e = ihi + 1;                  // the run reaches the table edge (comparator.rs:197)
        b = e - 1;
        first = false;
      }
      uint32_t badmask = 0u;
#pragma unroll
      for (int i = 0; i < kBack; i++) badmask |= bad(cb[i], q - 64 * i - lane >= ilo) ? 1u << i : 0u;
#if NEEDLE_M2_LAB & 16   // laboratory: the trip made, nothing done with it
      if (threshold < (1u << 20)) { if (badmask == 0x12345u) atomicAdd(count, 1u); return; }
#endif
      over = back_step(badmask);
    }
  };

  // The workgroup's chain queue: a ring of kM2Chains entries behind a lock (ctl[kCtlLock]); ctl[kCtlTaken] counts the chains taken, ctl[kCtlPushed]
  // the chains pushed.  All wave-uniform; lane 0 does the work, a handful of LDS operations under the lock.
  auto chain_lock = [&]() __attribute__((always_inline)) {
    while (__hip_atomic_exchange(&ctl[kCtlLock], 1u, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != 0u) __builtin_amdgcn_s_sleep(1);
  };
  auto chain_unlock = [&]() __attribute__((always_inline)) {
    __hip_atomic_store(&ctl[kCtlLock], 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  };
  auto push_chain = [&](const uint32_t wm, const int d) __attribute__((always_inline)) {   // false: the queue is full
    uint32_t ok = 0u;
    if (lane == 0) {
      chain_lock();
      const uint32_t taken = __hip_atomic_load(&ctl[kCtlTaken], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      const uint32_t pushed = __hip_atomic_load(&ctl[kCtlPushed], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (pushed - taken < (uint32_t)kM2Chains) {
        const uint32_t at = pushed & (uint32_t)(kM2Chains - 1);
        __hip_atomic_store(&chains[2 * at], wm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_store(&chains[2 * at + 1], (uint32_t)d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_store(&ctl[kCtlPushed], pushed + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        ok = 1u;
      }
      chain_unlock();
    }
    return __builtin_amdgcn_readfirstlane((int)ok) != 0;
  };
  auto pop_chain = [&](uint32_t &wm, uint32_t &d) __attribute__((always_inline)) {
    uint32_t e0 = 0u, e1 = 0u;                    // (w0 >= 1: a chain's first word is never 0)
    if (lane == 0 && __hip_atomic_load(&ctl[kCtlPushed], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) !=
                         __hip_atomic_load(&ctl[kCtlTaken], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
      chain_lock();
      const uint32_t taken = __hip_atomic_load(&ctl[kCtlTaken], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (taken != __hip_atomic_load(&ctl[kCtlPushed], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
        const uint32_t at = taken & (uint32_t)(kM2Chains - 1);
        e0 = __hip_atomic_load(&chains[2 * at], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        e1 = __hip_atomic_load(&chains[2 * at + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_store(&ctl[kCtlTaken], taken + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      chain_unlock();
    }
    wm = (uint32_t)__builtin_amdgcn_readfirstlane((int)e0);
    d = (uint32_t)__builtin_amdgcn_readfirstlane((int)e1);
    return wm != 0u;
  };

  // What the flags point at.  An ITEM = (group of four windows, destination position) that may hold a survivor.  Items
  // are handed out one per lane, 64 at a time, whatever lane flagged them (a destination position that looks like many
  // windows -- a sustained sound -- flags the same lane again and again: left to that lane, the wave would wait for it).
  // A lane tests its item's four windows on all eight rows with popcounts out of LDS; what passes is looked at by the wave
  // (window_whole_wave, the chain rule) and, if it is the last window of its chain, resolved.
  auto process = [&](const int first, const int cnt) __attribute__((always_inline)) {
    wave_lds_fence_search();
    uint32_t passm = 0u;                          // windows of the group (bit i) whose W cells all match
    int kbase = 0, j = 0;
    if (lane < cnt) {
      // the item's token (enqueue(), below): flag bit | finder's lane << 5 | batch << 11 | unit << 17.  Decoded HERE, 64 items per
      // instruction; in enqueue() the same arithmetic served the five or so lanes of a turn that held a flag.
      const uint32_t token = queue[first + lane];
      const int f = (int)(token & 31u), fr = (int)((token >> 5) & 31u), fh = (int)((token >> 10) & 1u);
      const int brt0 = (int)((token >> 11) & 63u) * kM2Batch, bunit = (int)(token >> 17);
      const int idx = 8 * min(kM2Batch, row_tiles - brt0) - 1 - f;   // flag bit -> tile slot s = 2 (row tile - rt0) + column block, group g
      const int s = idx >> 2, g = idx & 3;
      kbase = 32 * (brt0 + (s >> 1)) + 8 * g + 4 * fh;
      j = 32 * CB * bunit + fr + 32 * (s & 1);
      // All eight rows at once: the window's rows come as two 16-byte reads anyway, the destination's eight serve the four
      // windows.  (Head rows first and the tail rows of the survivors in a loop of their own -- one turn per window with
      // four lanes in a hundred busy -- was a quarter more instructions.)
      uint32_t dr[W];
#pragma unroll
      for (int s = 0; s < W; s++) dr[s] = ldst[j + s];
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int k = min(kbase + i, nW - 1);
        const mfma_v4i lo4 = *reinterpret_cast<const mfma_v4i *>(wimg + k * STRIDE + PITCH + E);       // rows 0 .. 3 of the window
        const mfma_v4i hi4 = *reinterpret_cast<const mfma_v4i *>(wimg + k * STRIDE + PITCH + E + 4);   // rows 4 .. 7
        uint32_t miss = 0u;
#pragma unroll
        for (int s = 0; s < 4; s++)
          miss |= ((uint32_t)__popc((uint32_t)lo4[s] ^ dr[s]) + bias) | ((uint32_t)__popc((uint32_t)hi4[s] ^ dr[4 + s]) + bias);
        passm |= (miss < 32u && kbase + i < nW) ? 1u << i : 0u;
      }
    }
#if NEEDLE_M2_LAB & 2   // laboratory: survivors dropped (threshold < 2^20: the mask is zero, but not for the compiler)
    passm &= threshold >> 20;
#endif
    M2_COUNT_LANES(6, lane < cnt ? 1 : 0);
    M2_COUNT_LANES(7, __popc(passm));
    if (__builtin_amdgcn_ballot_w64(passm != 0u) == 0ull) return;
    wave_lds_fence_search();
    overflow = queue + first;                     // the items are in registers: their (and the following, unused) 64 words hold overflow runs
    overflowed = 0;
    int resolved_here = 0;                        // by this call (wave-uniform)
    // What passes is rare (a window in three thousand of the items') and is looked at by the WAVE, window by window:
    // whole?  the last of its chain -- no successor (the member's last window), or one that is not whole on this diagonal?
    // Then the chain is resolved here, or goes to the WORKGROUP's queue (drain(), below).
#pragma unroll 1
    for (int i = 0; i < 4; i++) {
      const bool tails = ((passm >> i) & 1u) != 0u;
      const int k = kbase + i;
      unsigned long long cand = __builtin_amdgcn_ballot_w64(tails);
      M2_COUNT_LANES(5, tails ? 1 : 0);
      while (cand) {
        const int src_lane = __ffsll((long long)cand) - 1;
        cand &= cand - 1;
        const int ck = __builtin_amdgcn_readlane(k, src_lane), cj = __builtin_amdgcn_readlane(j, src_lane);
        const uint32_t wm = __builtin_amdgcn_readfirstlane(wimg[ck * STRIDE + 4 * H]);
        const int g = (int)(wm >> 28), w0 = (int)(wm & 0x0FFFFFFFu);
        if (!window_whole_wave(ck, g, w0, cj)) continue;
        if (ck + 1 < (int)__builtin_amdgcn_readfirstlane(ctl[kCtlRows + 1 + g]) && window_whole_wave(ck + 1, g, w0 + P, cj + P)) continue;
        // Found while the workgroup still has plenty of units to hand out: resolved here, beside the other waves' tiles.  Found
        // LATE -- the units left are fewer than the waves would take during a cluster's eight resolutions -- or in a CROWD (this
        // call has resolved kM2Crowd already: a block of equal hashes, every diagonal through it a chain) it goes to the queue,
        // which the waves that have run out of units empty (drain()); resolved here after all when the queue is full.
        const bool late = ((int)__builtin_amdgcn_readfirstlane(ctl[kCtlUnits]) + kM2LateUnits * WAVES) * splits >= units;   // (ctl[kCtlUnits]: this workgroup's share)
        if ((late || resolved_here >= kM2Crowd) && push_chain(wm, cj - w0)) continue;
        resolve(w0, cj - w0, g);
        resolved_here++;
      }
    }
    flush_overflow();                             // (the words are the queue's again when this call returns)
  };
  // Chains from the workgroup's queue, whoever found them -- behind a wave's last unit, when it has flushed its items and will
  // push nothing any more.  It says so (ctl[kCtlThrough]) and STAYS until every wave has: the waves of a workgroup whose units are cheap
  // finish first and then take what the waves in a block of equal hashes keep pushing (the hostile corpus, 280 files: scan
  // 15.3 -> 8.5 ms; the tonal figures unchanged).  (No items wait in the wave's queue here: its upper 64 words hold the overflow runs.)
  auto drain = [&]() __attribute__((always_inline)) {
    overflow = queue + 64;
    overflowed = 0;
    if (lane == 0) __hip_atomic_fetch_add(&ctl[kCtlThrough], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    for (;;) {
      const bool all = __builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&ctl[kCtlThrough], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) == WAVES;
      uint32_t wm = 0u, dd = 0u;
      while (pop_chain(wm, dd)) resolve((int)(wm & 0x0FFFFFFFu), (int)dd, (int)(wm >> 28));
      if (all) break;                             // (read BEFORE the queue was found empty: nothing can have come since)
      __builtin_amdgcn_s_sleep(8);
    }
    flush_overflow();
  };
  // A batch's flags become items.  flags: bit 4 n - 1 - (4 s + g) set <=> tile slot s (s = 2 (row tile - rt0) + column block,
  // n slots in all), group g.  One item per lane and turn (turns = the most set bits any lane holds), 64 are processed
  // as soon as they are there.  An item goes into the queue as a TOKEN that names its flag bit (round 6): a turn serves the few
  // lanes that hold a flag, and the arithmetic from bit to (windows, position) -- 9 of a turn's 17 vector instructions, 2.6 per
  // tile beside the fold's 20 -- is done by process() for 64 items at a time instead.
  int qn = 0;                                     // wave-uniform: items waiting in this wave's queue (< 64 between turns)
  // (ONE call site, and one of process() inside: the rare paths' code, inlined at three sites each, had grown to where the
  // waves waited for instructions -- the same head tests took 0.12 instead of 0.08 ms.  flush: the wave's last call.)
  auto enqueue = [&](uint32_t flags, const int rt0, const int unit, const bool flush) __attribute__((always_inline)) {
    // (unit < 32768 and rt0 / kM2Batch < 64: m < 65536 and fewer than 8192 windows per group -- search.hip build_plan stages nothing else)
    const uint32_t mine = ((uint32_t)unit << 17) | ((uint32_t)(rt0 / kM2Batch) << 11) | ((uint32_t)lane << 5);
    for (;;) {
      const bool act = flags != 0u;
      const unsigned long long ball = __builtin_amdgcn_ballot_w64(act);
      if (act) {
        const uint32_t f = (uint32_t)(__ffs((int)flags) - 1);
        flags &= flags - 1u;
        const uint32_t before = __builtin_amdgcn_mbcnt_hi((uint32_t)(ball >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ball, 0u));
        queue[qn + (int)before] = mine | f;
      }
      qn += (int)__popcll(ball);
      const int cnt = qn >= 64 ? 64 : (ball == 0ull && flush) ? qn : 0;
      if (cnt > 0) {
        process(qn - cnt, cnt);
        qn -= cnt;
      }
      if (ball == 0ull) break;
    }
  };

  const uint32_t *arow = wimg + r * STRIDE + 4 * h;                                         // rows of a full tile: + rt 32 STRIDE
  const uint32_t *arow_last = wimg + min(32 * (row_tiles - 1) + r, nW) * STRIDE + 4 * h;  // the last tile's may lie beyond the last window
  // (Measured and dropped, round 6: the last tile = the LAST 32 windows, overlapping its neighbour instead of reaching beyond the
  // last window -- a tile's address is then one per-lane base plus a scalar, 41 instead of 43 vector instructions per two tiles --
  // with the overlap's second flags dropped in enqueue(): 2.81 against 2.66 ms at 79 800 pairs of 45-minute windows, 0.665
  // against 0.650 at 39 060 pairs: the overlap's flags and the second ballot per turn cost more than the two instructions.)
  auto load_a = [&](const int rt, mfma_v4i (&fa)[HP]) {
    const uint32_t *ap = rt == row_tiles - 1 ? arow_last : arow + rt * 32 * STRIDE;
#pragma unroll
    for (int kb = 0; kb < HP; kb++) fa[kb] = *reinterpret_cast<const mfma_v4i *>(ap + 8 * kb);
  };
  mfma_v16f presets;
#pragma unroll
  for (int q = 0; q < 16; q++) presets[q] = (float)preset;
  // One product = TWO head rows: lane half h of the A and B fragments holds the 32 nibbles of the first (h = 0) or second
  // (h = 1) row of the pair, K = 64 sums over both.  Rows (0, 2) -> ua, rows (4, 7) -> ub: independent, no chain.
  // (Both operands FP4, scale operands 0: the compiler picks the unscaled v_mfma_f32_32x32x64_f8f6f4, whose scale is 2^0 --
  // tools/mfma_fp4_probe.hip.  The fragments are 4 registers; the builtin's type is 8 wide, the upper half is dropped.)
  auto product = [&](const mfma_v4i &a, const mfma_v4i &b) {
    const mfma_v8i a8 = {a[0], a[1], a[2], a[3], 0, 0, 0, 0}, b8 = {b[0], b[1], b[2], b[3], 0, 0, 0, 0};
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, presets, 4, 4, 0, 0, 0, 0);
  };
  auto products = [&](const mfma_v4i (&fa)[HP], const mfma_v4i (&fbk)[HP], mfma_v16f &ua, mfma_v16f &ub) {
    asm volatile("" : "+v"(presets));            // stays in its 16 registers (otherwise re-built from scalars for every tile)
    ua = product(fa[0], fbk[0]);
    ub = product(fa[1], fbk[1]);
  };
  // a group's word is negative iff one of its four windows passes both sums; its sign goes into the flags
  auto fold = [&](const mfma_v16f &ua, const mfma_v16f &ub, uint32_t &flags) {
#pragma unroll
    for (int g = 0; g < 4; g++) {
      uint32_t a = __float_as_uint(ua[4 * g]) & __float_as_uint(ub[4 * g]);
#pragma unroll
      for (int q = 4 * g + 1; q < 4 * g + 4; q++)                // (ua & ub) | a: one instruction
        a = (uint32_t)__builtin_amdgcn_bitop3_b32((int)__float_as_uint(ua[q]), (int)__float_as_uint(ub[q]), (int)a, 0xEA);
      flags = __builtin_amdgcn_alignbit(flags, a, 31);           // (written as C the compiler makes it and, and, and, and_or, or3)
    }
  };

#if NEEDLE_M2_LAB & 64   // laboratory: a workgroup's setup alone
  if (threshold < (1u << 20)) return;
#endif
  // One accumulator pair, and the compiler held to it: left alone it issues the second tile's products inside the first
  // tile's fold on a SECOND pair, and at 128 registers that costs B fragments copied and spilled in every turn.  (Round 5's
  // int8 form also had a software-pipelined loop over two pairs for the 3-waves-per-SIMD shapes: 5 % slower, gone.)
  bool done = false;
  while (!done) {
    int c = 0;
    if (lane == 0) c = (int)__hip_atomic_fetch_add(&ctl[kCtlUnits], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    c = __builtin_amdgcn_readfirstlane(c);
    const int unit = b_in_group + splits * c;     // (every wave sees the counter pass `units`: the loop ends for all)
    done = unit >= units;                         // then one more turn with nothing in it: the queue's last items

    // B fragments: lane (r, h) holds dst[j + s] for the head rows s = (0 | 2), (4 | 7) of its half h as 32 nibbles of +-1,
    // eight bits at a time through a table of 256 words
    mfma_v4i fb[CB][HP], fa[HP];
    const int j0 = 32 * unit * CB + r;            // this lane's position in the unit's first column block; + 32 in the second
    if (!done) {
#pragma unroll
      for (int cbk = 0; cbk < CB; cbk++) {
        const int jr = min(j0 + 32 * cbk, m);     // (a unit's second block may lie beyond the table: zeros)
#pragma unroll
        for (int kb = 0; kb < HP; kb++) {
          const uint32_t hv = ldst[jr + (kb == 0 ? 2 * h : 4 + 3 * h)];
#pragma unroll
          for (int q = 0; q < 4; q++) fb[cbk][kb][q] = (int)ntab[(hv >> (8 * q)) & 0xFFu];
        }
      }
      if (unit == 0 || unit * CB + CB > last_j / 32) {  // (wave-uniform) a unit with positions outside 1 .. last_j: their columns
#pragma unroll                                          // become zeros -- product 0 + preset: positive, never flagged
        for (int cbk = 0; cbk < CB; cbk++) {
          const bool ok = j0 + 32 * cbk >= 1 && j0 + 32 * cbk <= last_j;
#pragma unroll
          for (int kb = 0; kb < HP; kb++)
#pragma unroll
            for (int q = 0; q < 4; q++) fb[cbk][kb][q] = ok ? fb[cbk][kb][q] : 0;
        }
      }
      load_a(0, fa);
    }
    // the unit's tiles in the order (row tile, column block), kM2Batch row tiles between two looks at the flags
    int rt0 = 0;
    do {
      uint32_t flags = 0u;                        // sign bits of the groups' words: set = look here
      const int nb = done ? 0 : min(kM2Batch, row_tiles - rt0);

      mfma_v16f a0, b0;
#pragma unroll 1
      for (int rt = rt0; rt < rt0 + nb; rt++) {
        products(fa, fb[0], a0, b0);
        fold(a0, b0, flags);
        __builtin_amdgcn_sched_barrier(0);
        products(fa, fb[1], a0, b0);
        load_a(min(rt + 1, row_tiles - 1), fa);   // (behind the last product that reads this row tile's; the last one's again)
        fold(a0, b0, flags);
        __builtin_amdgcn_sched_barrier(0);
      }
#if !(NEEDLE_M2_LAB & 1)   // laboratory: flags ignored -- the tile loop alone
      enqueue(flags, rt0, unit, done);
#else
      if (flags == 0x12345u + threshold) atomicAdd(count, 1u);
#endif
      rt0 += kM2Batch;
    } while (!done && rt0 < row_tiles);
    if (done) drain();                            // the workgroup's chains, by all its waves
  }
  // the workgroup's runs: one request for room, then the copy
  __syncthreads();
  const uint32_t n_runs = min(ctl[kCtlRuns], (uint32_t)kM2RunBuf);
  if (n_runs == 0u) return;
  if (threadIdx.x == 0) ctl[kCtlSlot] = atomicAdd(count, n_runs);
  __syncthreads();
  if (threadIdx.x < n_runs) {
    const uint32_t slot = ctl[kCtlSlot] + threadIdx.x;
    const uint32_t *e = runbuf + 4 * threadIdx.x;
    if (slot < capacity) runs[slot] = NeedleHipRun{e[0], e[1], e[2], e[3], 0u, 0u};
  }
}
