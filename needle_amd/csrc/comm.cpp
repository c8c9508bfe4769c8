// Transport of the multi-GPU path (comm.h): RCCL through dlopen, or host-staged shared memory.
#include "comm.h"

#include <dlfcn.h>
#include <fcntl.h>
#include <rccl/rccl.h>  // types and enums only: every function is resolved with dlsym
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <random>
#include <string>

#include "hipctx.h"

namespace needle {

namespace {

static_assert(sizeof(ncclUniqueId) == 128, "the public id is one ncclUniqueId");

struct RcclApi {
  void *handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*GetVersion)(int *) = nullptr;
  // optional (comm_all_to_all_v): absent -> the callers keep their all-gathers
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
};

Status load_rccl(RcclApi *api) {
  static std::mutex mu;
  static RcclApi loaded;
  std::lock_guard<std::mutex> lock(mu);
  if (!loaded.handle) {
    const char *names[] = {getenv("NEEDLE_HIP_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    std::string tried;
    for (const char *name : names) {
      if (!name || !*name) continue;
      loaded.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (loaded.handle) break;
      tried += std::string(tried.empty() ? "" : "; ") + dlerror();
    }
    if (!loaded.handle) return Status::Make(NeedleError_Unknown, "cannot load librccl: " + tried);
    auto sym = [&](const char *n) { return dlsym(loaded.handle, n); };
    loaded.GetUniqueId = reinterpret_cast<decltype(loaded.GetUniqueId)>(sym("ncclGetUniqueId"));
    loaded.CommInitRank = reinterpret_cast<decltype(loaded.CommInitRank)>(sym("ncclCommInitRank"));
    loaded.CommDestroy = reinterpret_cast<decltype(loaded.CommDestroy)>(sym("ncclCommDestroy"));
    loaded.AllGather = reinterpret_cast<decltype(loaded.AllGather)>(sym("ncclAllGather"));
    loaded.GetErrorString = reinterpret_cast<decltype(loaded.GetErrorString)>(sym("ncclGetErrorString"));
    loaded.GetVersion = reinterpret_cast<decltype(loaded.GetVersion)>(sym("ncclGetVersion"));
    loaded.Send = reinterpret_cast<decltype(loaded.Send)>(sym("ncclSend"));
    loaded.Recv = reinterpret_cast<decltype(loaded.Recv)>(sym("ncclRecv"));
    loaded.GroupStart = reinterpret_cast<decltype(loaded.GroupStart)>(sym("ncclGroupStart"));
    loaded.GroupEnd = reinterpret_cast<decltype(loaded.GroupEnd)>(sym("ncclGroupEnd"));
    if (!loaded.GetUniqueId || !loaded.CommInitRank || !loaded.CommDestroy || !loaded.AllGather ||
        !loaded.GetErrorString) {
      dlclose(loaded.handle);
      loaded = RcclApi{};
      return Status::Make(NeedleError_Unknown, "librccl lacks a required symbol");
    }
  }
  *api = loaded;
  return Status::Ok();
}

// ---- host-staged backend: ranks of one node exchange through a POSIX shared-memory segment --------------------------
constexpr char kHostMagic[8] = {'N', 'H', 'C', 'O', 'M', 'M', '1', 0};

struct ShmHeader {
  uint32_t world;
  uint32_t pad;
  uint64_t slot_bytes;
  std::atomic<uint32_t> arrived;
  std::atomic<uint32_t> generation;
};
constexpr size_t kShmDataOffset = 4096;

size_t host_slot_bytes() {
  size_t v = 4u << 20;
  if (const char *e = getenv("NEEDLE_HIP_COMM_SLOT_BYTES")) v = (size_t)std::max(64ll, atoll(e));
  return (v + 63) & ~(size_t)63;
}

bool use_host_backend() {
  const char *e = getenv("NEEDLE_HIP_COMM");
  return e && std::strcmp(e, "host") == 0;
}

}  // namespace

struct Comm {
  int rank = 0, world = 1;
  bool host = false;
  // rccl
  RcclApi api;
  ncclComm_t nccl[2] = {nullptr, nullptr};
  // host
  std::string shm_name;
  char *shm = nullptr;
  size_t shm_bytes = 0;
  // staging for comm_all_gather_host
  DeviceBuffer<uint8_t> stage;

  ShmHeader *header() const { return reinterpret_cast<ShmHeader *>(shm); }
  char *slot(int r) const { return shm + kShmDataOffset + (size_t)r * header()->slot_bytes; }

  Status barrier_host() {
    ShmHeader *h = header();
    const uint32_t gen = h->generation.load(std::memory_order_acquire);
    if (h->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)world) {
      h->arrived.store(0, std::memory_order_relaxed);
      h->generation.store(gen + 1, std::memory_order_release);
      return Status::Ok();
    }
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    while (h->generation.load(std::memory_order_acquire) == gen) {
      if (++spins < 2000) continue;
      sched_yield();
      if ((spins & 0xfff) == 0 &&
          std::chrono::steady_clock::now() - t0 > std::chrono::seconds(300))
        return Status::Make(NeedleError_Unknown, "host communicator: a rank did not reach the barrier within 300 s");
    }
    return Status::Ok();
  }
};

namespace {
Comm *g_comm = nullptr;
std::mutex g_comm_mu;

Status nccl_status(const RcclApi &api, ncclResult_t r, const char *what) {
  if (r == ncclSuccess) return Status::Ok();
  return Status::Make(NeedleError_Unknown, std::string("RCCL error in ") + what + ": " + api.GetErrorString(r));
}
}  // namespace

// Rank processes sharing this node's CPUs: the launcher's LOCAL_WORLD_SIZE when it exports one (torchrun does), else the
// world size (this transport is single-node).  Sizes every rank's host thread pool (common.h host_threads()).
unsigned node_ranks(int world) {
  if (const char *e = getenv("LOCAL_WORLD_SIZE"))
    if (atoi(e) > 0) return (unsigned)atoi(e);
  return (unsigned)std::max(world, 1);
}

Comm *comm_get() { return g_comm; }
int comm_rank() { return g_comm ? g_comm->rank : 0; }
int comm_world() { return g_comm ? g_comm->world : 1; }
const char *comm_backend() { return !g_comm ? "none" : g_comm->host ? "host" : "rccl"; }

Status comm_create_id(uint8_t id[128]) {
  std::memset(id, 0, 128);
  if (use_host_backend()) {
    std::random_device rd;
    char name[96];
    std::snprintf(name, sizeof(name), "/needle_comm_%d_%08x%08x", (int)getpid(), (unsigned)rd(), (unsigned)rd());
    std::memcpy(id, kHostMagic, 8);
    std::memcpy(id + 8, name, std::strlen(name) + 1);
    return Status::Ok();
  }
  RcclApi api;
  Status s = load_rccl(&api);
  if (!s.ok()) return s;
  ncclUniqueId uid;
  if (!(s = nccl_status(api, api.GetUniqueId(&uid), "ncclGetUniqueId")).ok()) return s;
  std::memcpy(id, &uid, 128);
  return Status::Ok();
}

Status comm_init(const uint8_t id[128], int rank, int world) {
  if (world < 1 || rank < 0 || rank >= world) return Status::Make(NeedleError_InvalidArgument, "invalid rank / world size");
  std::lock_guard<std::mutex> lock(g_comm_mu);
  if (g_comm) return Status::Make(NeedleError_InvalidArgument, "a communicator already exists: call needle_hip_comm_finalize first");
  Status s;
  Comm *c = new Comm();
  c->rank = rank;
  c->world = world;
  c->host = std::memcmp(id, kHostMagic, 8) == 0;
  if (c->host) {
    c->shm_name = reinterpret_cast<const char *>(id + 8);
    const size_t slot = host_slot_bytes();
    c->shm_bytes = kShmDataOffset + (size_t)world * slot;
    int fd = -1;
    if (rank == 0) {
      fd = shm_open(c->shm_name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
      if (fd < 0 || ftruncate(fd, (off_t)c->shm_bytes) != 0) {
        if (fd >= 0) close(fd);
        delete c;
        return Status::Make(NeedleError_IOError, "IO error: cannot create the shared-memory segment of the host communicator");
      }
    } else {
      const auto t0 = std::chrono::steady_clock::now();
      for (;;) {  // rank 0 may not have created (or sized) it yet
        fd = shm_open(c->shm_name.c_str(), O_RDWR, 0600);
        struct stat st;
        if (fd >= 0 && fstat(fd, &st) == 0 && (size_t)st.st_size >= c->shm_bytes) break;
        if (fd >= 0) close(fd);
        fd = -1;
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) {
          delete c;
          return Status::Make(NeedleError_IOError, "IO error: the host communicator's shared-memory segment never appeared");
        }
        usleep(1000);
      }
    }
    void *p = mmap(nullptr, c->shm_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) {
      delete c;
      return Status::Make(NeedleError_IOError, "IO error: cannot map the host communicator's shared memory");
    }
    c->shm = static_cast<char *>(p);
    if (rank == 0) {  // a fresh segment is zero-filled: counters start at 0
      c->header()->slot_bytes = slot;
      std::atomic_thread_fence(std::memory_order_release);
      c->header()->world = (uint32_t)world;
    } else {
      const auto t0 = std::chrono::steady_clock::now();
      while (reinterpret_cast<volatile uint32_t &>(c->header()->world) != (uint32_t)world) {
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) {
          munmap(c->shm, c->shm_bytes);
          delete c;
          return Status::Make(NeedleError_InvalidArgument, "host communicator: world size differs between ranks");
        }
        usleep(200);
      }
      std::atomic_thread_fence(std::memory_order_acquire);
    }
    s = c->barrier_host();
    if (rank == 0) shm_unlink(c->shm_name.c_str());  // every rank has it mapped: the name can go
    if (!s.ok()) {
      munmap(c->shm, c->shm_bytes);
      delete c;
      return s;
    }
    g_comm = c;
    set_node_ranks(node_ranks(world));
    return Status::Ok();
  }

  if (!(s = ensure_device()).ok() || !(s = load_rccl(&c->api)).ok()) {
    delete c;
    return s;
  }
  ncclUniqueId uid;
  std::memcpy(&uid, id, 128);
  s = nccl_status(c->api, c->api.CommInitRank(&c->nccl[kData], world, uid, rank), "ncclCommInitRank");
  if (!s.ok()) {
    delete c;
    return s;
  }
  g_comm = c;
  // ONE RCCL communicator by default.  RCCL serialises the collectives of one communicator in the order they were
  // enqueued even when they sit on different streams, and every rank enqueues them in the same program order (hash
  // rows of job k, run slabs of job k, hash rows of job k + 1, ...), so nothing can deadlock; the price is that job
  // k + 1's row gather starts behind job k's (small) slab gather.  Two communicators on one device run their kernels
  // concurrently, which NCCL documents as deadlock-prone unless the device can co-schedule both on every rank:
  // NEEDLE_HIP_COMM_DUAL=1 opts into that (a second communicator for the download stream) for measurements.
  const char *dual = getenv("NEEDLE_HIP_COMM_DUAL");
  if (world > 1 && dual && atoi(dual) != 0) {
    // the side communicator's id travels over the first one: every rank contributes 128 bytes, rank 0's count
    std::vector<uint8_t> mine(128, 0), all((size_t)world * 128, 0);
    if (rank == 0) {
      ncclUniqueId side;
      s = nccl_status(c->api, c->api.GetUniqueId(&side), "ncclGetUniqueId");
      std::memcpy(mine.data(), &side, 128);
    }
    c->nccl[kSide] = c->nccl[kData];  // comm_all_gather_host runs on the side channel
    if (s.ok()) s = comm_all_gather_host(mine.data(), all.data(), 128);
    c->nccl[kSide] = nullptr;
    if (s.ok()) {
      std::memcpy(&uid, all.data(), 128);
      s = nccl_status(c->api, c->api.CommInitRank(&c->nccl[kSide], world, uid, rank), "ncclCommInitRank (side)");
    }
    if (!s.ok()) {
      g_comm = nullptr;
      (void)c->api.CommDestroy(c->nccl[kData]);
      delete c;
      return s;
    }
  } else {
    c->nccl[kSide] = c->nccl[kData];
  }
  set_node_ranks(node_ranks(world));
  return Status::Ok();
}

void comm_finalize() {
  std::lock_guard<std::mutex> lock(g_comm_mu);
  Comm *c = g_comm;
  if (!c) return;
  g_comm = nullptr;
  set_node_ranks(1);
  (void)hipDeviceSynchronize();
  if (c->host) {
    if (c->shm) munmap(c->shm, c->shm_bytes);
  } else {
    if (c->nccl[kSide] && c->nccl[kSide] != c->nccl[kData]) (void)c->api.CommDestroy(c->nccl[kSide]);
    if (c->nccl[kData]) (void)c->api.CommDestroy(c->nccl[kData]);
  }
  delete c;
}

Status comm_all_gather(CommChannel ch, const void *d_send, void *d_recv, size_t bytes, hipStream_t stream) {
  Comm *c = g_comm;
  if (bytes % 4) return Status::Make(NeedleError_InvalidArgument, "all-gather size must be a multiple of 4 bytes");
  char *recv = static_cast<char *>(d_recv);
  const bool in_place = d_send == recv + (size_t)comm_rank() * bytes;
  // NEEDLE_HIP_COMM_FORCE_COLLECTIVES: a 1-rank RCCL communicator still goes through ncclAllGather (tests on one GPU)
  if (!c || (c->world == 1 && (c->host || !getenv("NEEDLE_HIP_COMM_FORCE_COLLECTIVES")))) {
    if (!in_place && bytes) NEEDLE_HIP_TRY(hipMemcpyAsync(recv, d_send, bytes, hipMemcpyDeviceToDevice, stream));
    return Status::Ok();
  }
  if (bytes == 0) return Status::Ok();
  if (!c->host)
    return nccl_status(c->api, c->api.AllGather(d_send, d_recv, bytes / 4, ncclUint32, c->nccl[ch], stream), "ncclAllGather");

  // host-staged: device -> my slot, barrier, every other slot -> device, barrier (slots are reused by the next chunk)
  const size_t slot = c->header()->slot_bytes;
  if (!in_place) NEEDLE_HIP_TRY(hipMemcpyAsync(recv + (size_t)c->rank * bytes, d_send, bytes, hipMemcpyDeviceToDevice, stream));
  for (size_t off = 0; off < bytes; off += slot) {
    const size_t len = std::min(slot, bytes - off);
    NEEDLE_HIP_TRY(hipMemcpyAsync(c->slot(c->rank), static_cast<const char *>(d_send) + off, len, hipMemcpyDeviceToHost, stream));
    NEEDLE_HIP_TRY(hipStreamSynchronize(stream));
    Status s = c->barrier_host();
    if (!s.ok()) return s;
    for (int r = 0; r < c->world; r++)
      if (r != c->rank)
        NEEDLE_HIP_TRY(hipMemcpyAsync(recv + (size_t)r * bytes + off, c->slot(r), len, hipMemcpyHostToDevice, stream));
    NEEDLE_HIP_TRY(hipStreamSynchronize(stream));
    if (!(s = c->barrier_host()).ok()) return s;
  }
  return Status::Ok();
}

Status comm_all_to_all_v(CommChannel ch, const void *d_send, const size_t *send_off, const size_t *send_bytes, void *d_recv,
                         const size_t *recv_off, const size_t *recv_bytes, size_t max_block_bytes, hipStream_t stream) {
  Comm *c = g_comm;
  const int world = comm_world(), rank = comm_rank();
  const char *send = static_cast<const char *>(d_send);
  char *recv = static_cast<char *>(d_recv);
  for (int q = 0; q < world; q++)
    if (send_bytes[q] % 4 || recv_bytes[q] % 4) return Status::Make(NeedleError_InvalidArgument, "all-to-all block sizes must be multiples of 4 bytes");
  // this rank's own block never leaves the device
  if (send_bytes[rank]) {
    if (send_bytes[rank] != recv_bytes[rank]) return Status::Make(NeedleError_InvalidArgument, "all-to-all: a rank's block to itself has two sizes");
    NEEDLE_HIP_TRY(hipMemcpyAsync(recv + recv_off[rank], send + send_off[rank], send_bytes[rank], hipMemcpyDeviceToDevice, stream));
  }
  if (!c || world == 1) return Status::Ok();
  if (!c->host) {
    if (!c->api.Send || !c->api.Recv || !c->api.GroupStart || !c->api.GroupEnd)
      return Status::Make(NeedleError_InvalidArgument, "all-to-all unsupported: librccl has no ncclSend / ncclRecv / ncclGroupStart / ncclGroupEnd");
    Status s = nccl_status(c->api, c->api.GroupStart(), "ncclGroupStart");
    for (int q = 0; q < world && s.ok(); q++) {
      if (q == rank) continue;
      if (send_bytes[q]) s = nccl_status(c->api, c->api.Send(send + send_off[q], send_bytes[q] / 4, ncclUint32, q, c->nccl[ch], stream), "ncclSend");
      if (s.ok() && recv_bytes[q])
        s = nccl_status(c->api, c->api.Recv(recv + recv_off[q], recv_bytes[q] / 4, ncclUint32, q, c->nccl[ch], stream), "ncclRecv");
    }
    Status e = nccl_status(c->api, c->api.GroupEnd(), "ncclGroupEnd");
    return s.ok() ? e : s;
  }
  // host-staged: world - 1 rounds; in round k a rank sends to rank + k and receives from rank - k, a slot's worth at a time
  // (every rank makes the same number of steps per round: the largest block of the call decides)
  const size_t slot = c->header()->slot_bytes;
  const size_t steps = (max_block_bytes + slot - 1) / slot;
  for (int k = 1; k < world; k++) {
    const int to = (rank + k) % world, from = (rank - k + world) % world;
    for (size_t step = 0; step < steps; step++) {
      const size_t off = step * slot;
      const size_t out = off < send_bytes[to] ? std::min(slot, send_bytes[to] - off) : 0;
      const size_t in = off < recv_bytes[from] ? std::min(slot, recv_bytes[from] - off) : 0;
      if (out) NEEDLE_HIP_TRY(hipMemcpyAsync(c->slot(rank), send + send_off[to] + off, out, hipMemcpyDeviceToHost, stream));
      NEEDLE_HIP_TRY(hipStreamSynchronize(stream));
      Status s = c->barrier_host();
      if (!s.ok()) return s;
      if (in) NEEDLE_HIP_TRY(hipMemcpyAsync(recv + recv_off[from] + off, c->slot(from), in, hipMemcpyHostToDevice, stream));
      NEEDLE_HIP_TRY(hipStreamSynchronize(stream));
      if (!(s = c->barrier_host()).ok()) return s;
    }
  }
  return Status::Ok();
}

Status comm_all_gather_host(const void *send, void *recv, size_t bytes) {
  Comm *c = g_comm;
  if (!c || (c->world == 1 && (c->host || !getenv("NEEDLE_HIP_COMM_FORCE_COLLECTIVES")))) {
    if (send != recv && bytes) std::memmove(recv, send, bytes);
    return Status::Ok();
  }
  if (bytes == 0) return Status::Ok();
  if (c->host) {  // no device involved
    const size_t slot = c->header()->slot_bytes;
    for (size_t off = 0; off < bytes; off += slot) {
      const size_t len = std::min(slot, bytes - off);
      std::memcpy(c->slot(c->rank), static_cast<const char *>(send) + off, len);
      Status s = c->barrier_host();
      if (!s.ok()) return s;
      for (int r = 0; r < c->world; r++) std::memcpy(static_cast<char *>(recv) + (size_t)r * bytes + off, c->slot(r), len);
      if (!(s = c->barrier_host()).ok()) return s;
    }
    return Status::Ok();
  }
  std::lock_guard<std::recursive_mutex> gpu_lock(gpu_mutex());
  const size_t padded = (bytes + 3) & ~(size_t)3;
  Status s = c->stage.reserve(padded * (size_t)c->world);
  if (!s.ok()) return s;
  hipStream_t stream = download_stream();
  uint8_t *mine = c->stage.ptr + (size_t)c->rank * padded;
  NEEDLE_HIP_TRY(hipMemcpyAsync(mine, send, bytes, hipMemcpyHostToDevice, stream));
  if (!(s = comm_all_gather(kSide, mine, c->stage.ptr, padded, stream)).ok()) return s;
  for (int r = 0; r < c->world; r++)
    NEEDLE_HIP_TRY(hipMemcpyAsync(static_cast<char *>(recv) + (size_t)r * bytes, c->stage.ptr + (size_t)r * padded, bytes,
                                  hipMemcpyDeviceToHost, stream));
  NEEDLE_HIP_TRY(hipStreamSynchronize(stream));
  return Status::Ok();
}

Status comm_barrier() {
  Comm *c = g_comm;
  if (!c || c->world == 1) return Status::Ok();
  if (c->host) return c->barrier_host();
  uint32_t one = 1;
  std::vector<uint32_t> all((size_t)c->world);
  return comm_all_gather_host(&one, all.data(), sizeof(one));
}

}  // namespace needle
