// GPU replacement of Comparator::longest_common_hash_match's two table sweeps
// (needle/src/audio/comparator.rs:175-247).
//
// The reference fills an (n+1)x(m+1) table with t[i][j] = t[i-1][j-1]+1 when
// popcount(src[i]^dst[j]) <= threshold (rows/cols 0 forced to 0, :179-180) and then walks it backwards
// keeping cells that END a maximal diagonal run (:196-200).  Every diagonal is independent, so no table
// is ever built here: one lane walks one diagonal d = j - i over its valid cells (i >= 1, j >= 1),
// carrying the current run length in a register, and appends (i_end, j_end, L) for each maximal run
// with L >= min_len.  The duration test (:212-223) needs timestamps and stays on the host; min_len is
// the host's lower bound on the run length that can pass it, so the output list stays short.
//
// Integer-only (xor, popcount, compare, add): results are exact by construction.
#include "hipctx.h"

#include <algorithm>
#include <cstring>
#include <mutex>

namespace needle {

namespace {

constexpr int kDiagsPerBlock = 256;

struct SearchProblem {
  uint32_t src_off, n;  // hash arena offset + length of the source sequence
  uint32_t dst_off, m;
  uint32_t min_len, tag;
  uint32_t block_base;  // first workgroup of this problem in the grid
  uint32_t pad;
};

// One workgroup = 256 consecutive diagonals of one problem; both sequences staged in LDS.
__global__ __launch_bounds__(256) void hamming_runs_kernel(const uint32_t *__restrict__ hashes,
                                                           const SearchProblem *__restrict__ problems,
                                                           int num_problems, uint32_t threshold,
                                                           NeedleHipRun *__restrict__ runs, uint32_t capacity,
                                                           uint32_t *__restrict__ count) {
  extern __shared__ uint32_t lds[];
  // problem lookup: last problem with block_base <= blockIdx.x
  int lo = 0, hi = num_problems - 1;
  while (lo < hi) {
    int mid = (lo + hi + 1) >> 1;
    if (problems[mid].block_base <= blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const SearchProblem pr = problems[lo];
  const int n = (int)pr.n, m = (int)pr.m;
  uint32_t *s = lds, *t = lds + n;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s[i] = hashes[pr.src_off + i];
  for (int j = threadIdx.x; j < m; j += blockDim.x) t[j] = hashes[pr.dst_off + j];
  __syncthreads();

  // diagonals d = j - i with at least one cell i>=1, j>=1: d in [-(n-2), m-2]
  const int dd = (int)(blockIdx.x - pr.block_base) * kDiagsPerBlock + (int)threadIdx.x;
  const int num_diags = n + m - 3;
  if (dd >= num_diags) return;
  const int d = dd - (n - 2);
  const int i_lo = d < 0 ? 1 - d : 1;
  const int i_hi = min(n - 1, m - 1 - d);
  const uint32_t min_len = pr.min_len;
  uint32_t run = 0;
  for (int i = i_lo; i <= i_hi; i++) {
    const bool match = (uint32_t)__popc(s[i] ^ t[i + d]) <= threshold;
    if (match) {
      run++;
    } else {
      if (run >= min_len) {  // run ended at the previous cell
        const uint32_t slot = atomicAdd(count, 1u);
        if (slot < capacity) runs[slot] = NeedleHipRun{pr.tag, (uint32_t)(i - 1), (uint32_t)(i - 1 + d), run};
      }
      run = 0;
    }
  }
  if (run >= min_len) {  // run reaches the table edge (i == n-1 or j == m-1, comparator.rs:197)
    const uint32_t slot = atomicAdd(count, 1u);
    if (slot < capacity) runs[slot] = NeedleHipRun{pr.tag, (uint32_t)i_hi, (uint32_t)(i_hi + d), run};
  }
}

struct SearchWorkspace {
  DeviceBuffer<SearchProblem> problems;
  PinnedStage stage;
};
std::mutex g_ws_mu;
std::map<int, SearchWorkspace *> g_ws;

SearchWorkspace *workspace() {
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lock(g_ws_mu);
  auto it = g_ws.find(dev);
  if (it != g_ws.end()) return it->second;
  SearchWorkspace *w = new SearchWorkspace();
  g_ws[dev] = w;
  return w;
}

bool g_lds_attr_set = false;

}  // namespace

Status gpu_hamming_runs_device(const uint32_t *d_hashes, const NeedleHipSeq *seqs, size_t num_seqs,
                               const NeedleHipProblem *problems, size_t num_problems, uint32_t threshold,
                               NeedleHipRun *d_runs, uint32_t capacity, uint32_t *d_count, bool sync) {
  Status s = ensure_device();
  if (!s.ok()) return s;
  hipStream_t stream = library_stream();
  NEEDLE_HIP_TRY(hipMemsetAsync(d_count, 0, sizeof(uint32_t), stream));
  std::vector<SearchProblem> meta;
  meta.reserve(num_problems);
  uint64_t blocks = 0;
  size_t max_lds = 0;
  for (size_t p = 0; p < num_problems; p++) {
    const NeedleHipProblem &pr = problems[p];
    if (pr.src_seq >= num_seqs || pr.dst_seq >= num_seqs)
      return Status::Make(NeedleError_InvalidArgument, "hamming_runs: problem references a missing sequence");
    if (pr.min_len == 0) return Status::Make(NeedleError_InvalidArgument, "hamming_runs: min_len must be >= 1");
    const NeedleHipSeq &a = seqs[pr.src_seq], &b = seqs[pr.dst_seq];
    if (a.len < 2 || b.len < 2) continue;  // no cell with i >= 1 and j >= 1 (comparator.rs:165-167,179)
    SearchProblem m;
    m.src_off = a.offset;
    m.n = a.len;
    m.dst_off = b.offset;
    m.m = b.len;
    m.min_len = pr.min_len;
    m.tag = pr.tag;
    m.block_base = (uint32_t)blocks;
    m.pad = 0;
    const uint64_t diags = (uint64_t)a.len + b.len - 3;
    blocks += (diags + kDiagsPerBlock - 1) / kDiagsPerBlock;
    max_lds = std::max(max_lds, ((size_t)a.len + b.len) * sizeof(uint32_t));
    meta.push_back(m);
  }
  if (blocks > 0x7FFFFFFFull) return Status::Make(NeedleError_InvalidArgument, "hamming_runs: too many problems for one launch");
  if (max_lds > 160 * 1024)
    return Status::Make(NeedleError_InvalidArgument,
                        "hamming_runs: a sequence pair exceeds the 160 KiB LDS staging limit (40960 hashes)");
  if (!meta.empty()) {
    SearchWorkspace *ws = workspace();
    if (!(s = ws->problems.reserve(meta.size())).ok()) return s;
    if (!(s = ws->stage.acquire(meta.size() * sizeof(SearchProblem))).ok()) return s;
    std::memcpy(ws->stage.ptr, meta.data(), meta.size() * sizeof(SearchProblem));
    NEEDLE_HIP_TRY(hipMemcpyAsync(ws->problems.ptr, ws->stage.ptr, meta.size() * sizeof(SearchProblem),
                                  hipMemcpyHostToDevice, stream));
    ws->stage.mark(stream);
    if (max_lds > 64 * 1024 && !g_lds_attr_set) {
      NEEDLE_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(hamming_runs_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      g_lds_attr_set = true;
    }
    {
      KernelTimer timer("hamming_runs");
      hipLaunchKernelGGL(hamming_runs_kernel, dim3((uint32_t)blocks), dim3(256), max_lds, stream, d_hashes,
                         ws->problems.ptr, (int)meta.size(), threshold, d_runs, capacity, d_count);
    }
    NEEDLE_HIP_TRY(hipGetLastError());
  }
  if (sync) NEEDLE_HIP_TRY(hipStreamSynchronize(stream));
  return Status::Ok();
}

Status gpu_hamming_runs_host(const uint32_t *hashes, size_t num_hashes, const NeedleHipSeq *seqs, size_t num_seqs,
                             const NeedleHipProblem *problems, size_t num_problems, uint32_t threshold,
                             std::vector<NeedleHipRun> *runs) {
  Status s = ensure_device();
  if (!s.ok()) return s;
  for (size_t i = 0; i < num_seqs; i++)
    if ((uint64_t)seqs[i].offset + seqs[i].len > num_hashes)
      return Status::Make(NeedleError_InvalidArgument, "hamming_runs: sequence outside the hash arena");
  hipStream_t stream = library_stream();
  DeviceBuffer<uint32_t> d_hashes, d_count;
  DeviceBuffer<NeedleHipRun> d_runs;
  if (!(s = d_hashes.reserve(std::max<size_t>(num_hashes, 1))).ok()) return s;
  if (!(s = d_count.reserve(1)).ok()) return s;
  NEEDLE_HIP_TRY(hipMemcpyAsync(d_hashes.ptr, hashes, num_hashes * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
  uint32_t capacity = 1u << 16;
  for (int attempt = 0; attempt < 2; attempt++) {
    if (!(s = d_runs.reserve(capacity)).ok()) return s;
    s = gpu_hamming_runs_device(d_hashes.ptr, seqs, num_seqs, problems, num_problems, threshold, d_runs.ptr,
                                capacity, d_count.ptr, false);
    if (!s.ok()) return s;
    uint32_t found = 0;
    NEEDLE_HIP_TRY(hipMemcpyAsync(&found, d_count.ptr, sizeof(found), hipMemcpyDeviceToHost, stream));
    NEEDLE_HIP_TRY(hipStreamSynchronize(stream));
    if (found <= capacity) {
      runs->resize(found);
      if (found)
        NEEDLE_HIP_TRY(hipMemcpy(runs->data(), d_runs.ptr, found * sizeof(NeedleHipRun), hipMemcpyDeviceToHost));
      return Status::Ok();
    }
    capacity = found;  // the scan is deterministic: a second pass with the exact size fits
  }
  return Status::Make(NeedleError_Unknown, "hamming_runs: run list did not fit after resize");
}

}  // namespace needle
