// libfakerccl.so — TEST INFRASTRUCTURE, never shipped, never loaded unless a test names its path
// (plonk_comm_set_library).  A stand-in for librccl that lets several ranks SHARE ONE GPU: it exports the nine
// nccl* entry points comm.hip resolves and runs ncclAllGather / ncclAllToAll on DEVICE pointers between
// processes, stream-ordered like the real collectives:
//
//   * every rank owns a device "mailbox" (hipMalloc); the hipIpcMemHandles are exchanged once, at
//     ncclCommInitRank, through a POSIX shared-memory segment whose name travels inside the 128-byte
//     ncclUniqueId, with a host barrier (the one place where the host waits for its peers);
//   * a collective is queued on the CALLER'S stream and returns at once:
//       1. a one-wave kernel waits until every peer has finished reading this rank's mailbox (previous collective),
//       2. hipMemcpyAsync  send buffer -> own mailbox,
//       3. a one-wave kernel publishes the sequence number (system-scope store into the shared segment, which every
//          process has registered with hipHostRegister) and spins until every peer has published the same number,
//       4. hipMemcpyAsync  peer mailbox (IPC mapping) -> receive buffer, one copy per peer,
//       5. a one-wave kernel marks the peers' mailboxes as consumed;
//   * a spinning kernel gives up when the communicator's abort flag is raised (ncclCommAbort, as RCCL's kernels do) or
//     after FAKE_RCCL_KERNEL_TIMEOUT_S seconds of device wall clock (default 120), so a dead peer can never hang the GPU.
//
// What this exercises in the product: the staging offsets of comm_allgather_host, the in-place all-gather of
// comm_allgather_dev (send = recv + rank * bytes), the [peer][bytes] block layout of comm_alltoall_dev, the ordering of
// the collectives against the kernels and copies around them on the library's stream, comm_sync's polling / time-out /
// abort path, and plonk_comm_info.  What it does NOT tell anyone: anything about xGMI, RCCL's protocols or their speed.
//
// Fault injection for the peer-failure test: FAKE_RCCL_DIE_RANK=r with FAKE_RCCL_DIE_AT=allgather:k | alltoall:k makes
// rank r _exit(17) on entering its k-th collective of that kind.
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

namespace {

constexpr int MAX_RANKS = 16;
constexpr uint32_t MAGIC = 0x46524343;   // "FRCC"

struct alignas(128) Slot {
  std::atomic<uint32_t> joined;     // host: mailbox handle below is valid
  std::atomic<uint32_t> opened;     // host: this rank has mapped every peer's mailbox
  std::atomic<uint32_t> left;       // host: communicator destroyed / aborted
  int32_t pid;
  hipIpcMemHandle_t mailbox;
  alignas(64) uint32_t published;   // device-written: last collective whose data sits in this rank's mailbox
  alignas(64) uint32_t consumed;    // device-written: last collective whose peer data this rank has copied out
};

struct Segment {
  uint32_t magic;
  uint32_t world;
  Slot slot[MAX_RANKS];
};

struct Comm {
  int rank = 0, world = 1;
  char shm_name[64] = {0};
  Segment* seg = nullptr;           // host mapping of the shared segment
  Segment* seg_dev = nullptr;       // the same pages as the GPU sees them (hipHostRegister)
  size_t seg_bytes = 0;
  uint8_t* mailbox = nullptr;
  size_t mailbox_bytes = 0;
  uint8_t* peer_box[MAX_RANKS] = {nullptr};
  uint32_t* abort_flag = nullptr;   // pinned, process-local: raised by ncclCommAbort
  uint32_t* abort_dev = nullptr;
  uint32_t* timed_out = nullptr;    // pinned: a spin kernel gave up on its own
  uint32_t seq = 0;
  hipStream_t last_stream = nullptr;
  int n_allgather = 0, n_alltoall = 0;
};

size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

double env_seconds(const char* name, double dflt) {
  const char* e = getenv(name);
  const double v = e ? atof(e) : 0.0;
  return v > 0 ? v : dflt;
}

// one wave: optional publish, then wait until every rank's flag has reached `wait_for`
__global__ void fake_rccl_sync(uint32_t* publish_to, uint32_t value, const uint32_t* flags, uint32_t stride_words, int world,
                               uint32_t wait_for, int do_wait, const uint32_t* abort_flag, uint32_t* timed_out, uint64_t max_ticks) {
  const int t = threadIdx.x;
  if (publish_to && t == 0) {
    __threadfence_system();
    __hip_atomic_store(publish_to, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (!do_wait || t >= world) return;
  const uint64_t t0 = wall_clock64();
  const uint32_t* f = flags + (size_t)stride_words * t;
  for (uint32_t spin = 0;; ++spin) {
    const uint32_t v = __hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
    if ((int32_t)(v - wait_for) >= 0) break;
    if ((spin & 63) == 63) {
      if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) break;
      if (wall_clock64() - t0 > max_ticks) { __hip_atomic_store(timed_out, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
    }
    __builtin_amdgcn_s_sleep(32);
  }
}

#define HIP_OK(expr)                                                                                     \
  do {                                                                                                   \
    hipError_t _e = (expr);                                                                              \
    if (_e != hipSuccess) {                                                                              \
      fprintf(stderr, "[fake_rccl] %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return ncclUnhandledCudaError;                                                                     \
    }                                                                                                    \
  } while (0)

template <class F>
bool wait_until(F&& done, double seconds) {
  const auto t0 = std::chrono::steady_clock::now();
  while (!done()) {
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > seconds) return false;
    std::this_thread::sleep_for(std::chrono::microseconds(200));
  }
  return true;
}

uint64_t kernel_timeout_ticks() {   // wall_clock64 counts at 100 MHz on gfx9
  return (uint64_t)(env_seconds("FAKE_RCCL_KERNEL_TIMEOUT_S", 120.0) * 1e8);
}

// Fault injection.  FAKE_RCCL_DIE_RANK / FAKE_RCCL_DIE_AT=<kind>:<k> make that rank _exit(17) on entering its k-th collective of
// that kind — counted from communicator creation, or, after fake_rccl_arm() (called by the test harness once the prover is
// built), from the moment of arming: "the z commitment's all-gather of the first proof" then stays the same index however
// many collectives prover creation needs (round 6 added three: mode agreement, the Lagrange-key check).
static int g_arm_allgather = 0, g_arm_alltoall = 0, g_seen_allgather = 0, g_seen_alltoall = 0;
extern "C" void fake_rccl_arm(void) { g_arm_allgather = g_seen_allgather; g_arm_alltoall = g_seen_alltoall; }

void maybe_die(Comm* c, const char* kind, int nth) {
  const bool ag = kind[3] == 'g';   // "allgather" / "alltoall"
  (ag ? g_seen_allgather : g_seen_alltoall) = nth;
  nth -= ag ? g_arm_allgather : g_arm_alltoall;
  const char* r = getenv("FAKE_RCCL_DIE_RANK");
  const char* at = getenv("FAKE_RCCL_DIE_AT");
  if (!r || !at || atoi(r) != c->rank) return;
  const size_t kl = strlen(kind);
  if (strncmp(at, kind, kl) == 0 && at[kl] == ':' && atoi(at + kl + 1) == nth) {
    fprintf(stderr, "[fake_rccl] rank %d: injected failure on entering %s #%d\n", c->rank, kind, nth);
    fflush(stderr);
    _exit(17);
  }
}

ncclResult_t launch_sync(Comm* c, hipStream_t st, uint32_t* publish_to, uint32_t value, const uint32_t* flags, uint32_t wait_for, int do_wait) {
  const uint32_t stride = sizeof(Slot) / sizeof(uint32_t);
  hipLaunchKernelGGL(fake_rccl_sync, dim3(1), dim3(64), 0, st, publish_to, value, flags, stride, c->world, wait_for, do_wait,
                     c->abort_dev, c->timed_out, kernel_timeout_ticks());
  HIP_OK(hipGetLastError());
  return ncclSuccess;
}

// the shared shape of both collectives: `mine` bytes of the send buffer go to the mailbox; from peer p the `bytes` at
// offset src_off(p) of its mailbox land at recv + p * bytes
ncclResult_t exchange(Comm* c, const void* send, size_t mine, void* recv, size_t bytes, bool from_my_block, hipStream_t st) {
  if (mine > c->mailbox_bytes) {
    fprintf(stderr, "[fake_rccl] message of %zu bytes exceeds the %zu-byte mailbox (FAKE_RCCL_MAILBOX_MB)\n", mine, c->mailbox_bytes);
    return ncclInvalidArgument;
  }
  if (__atomic_load_n(c->timed_out, __ATOMIC_ACQUIRE)) {   // an earlier collective's kernel gave up on its own: its data was garbage
    fprintf(stderr, "[fake_rccl] rank %d: a collective timed out on the device (a peer never arrived); refusing further collectives\n", c->rank);
    return ncclRemoteError;
  }
  c->last_stream = st;
  const uint32_t k = ++c->seq;
  Slot* sd = c->seg_dev->slot;
  ncclResult_t r;
  if ((r = launch_sync(c, st, nullptr, 0, &sd[0].consumed, k - 1, 1)) != ncclSuccess) return r;
  if (mine) HIP_OK(hipMemcpyAsync(c->mailbox, send, mine, hipMemcpyDeviceToDevice, st));
  if ((r = launch_sync(c, st, &sd[c->rank].published, k, &sd[0].published, k, 1)) != ncclSuccess) return r;
  for (int p = 0; p < c->world && bytes; ++p) {
    const uint8_t* src = (p == c->rank ? c->mailbox : c->peer_box[p]) + (from_my_block ? bytes * (size_t)c->rank : 0);
    HIP_OK(hipMemcpyAsync((uint8_t*)recv + bytes * (size_t)p, src, bytes, hipMemcpyDeviceToDevice, st));
  }
  return launch_sync(c, st, &sd[c->rank].consumed, k, nullptr, 0, 0);
}

size_t type_bytes(ncclDataType_t t) {
  switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
  }
}

void release(Comm* c, bool wait_for_peers) {
  (void)hipStreamSynchronize(c->last_stream);   // (the null stream when no collective ran)
  if (c->seg) {
    if (wait_for_peers) {   // nobody may still be reading this rank's mailbox
      const uint32_t last = c->seq;
      (void)wait_until([&] {
        for (int p = 0; p < c->world; ++p) {
          if (p == c->rank || c->seg->slot[p].left.load()) continue;
          const uint32_t v = __atomic_load_n(&c->seg->slot[p].consumed, __ATOMIC_ACQUIRE);
          if ((int32_t)(v - last) < 0) return false;
        }
        return true;
      }, 20.0);
    }
    c->seg->slot[c->rank].left.store(1);
  }
  for (int p = 0; p < c->world; ++p)
    if (c->peer_box[p]) (void)hipIpcCloseMemHandle(c->peer_box[p]);
  if (c->mailbox) (void)hipFree(c->mailbox);
  if (c->seg) {
    (void)hipHostUnregister(c->seg);
    munmap(c->seg, c->seg_bytes);
  }
  if (c->abort_flag) (void)hipHostFree(c->abort_flag);
  if (c->timed_out) (void)hipHostFree(c->timed_out);
  delete c;
}

}  // namespace

extern "C" {

// exported beside the nccl* names so that a test can ask the loaded library what it is
const char* fakeRcclName(void) { return "fake-rccl (hipIpc mailboxes on one GPU; test stand-in)"; }

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  if (!id) return ncclInvalidArgument;
  memset(id, 0, sizeof(*id));
  unsigned long long rnd = 0;
  if (FILE* f = fopen("/dev/urandom", "rb")) { (void)!fread(&rnd, sizeof rnd, 1, f); fclose(f); }
  char* name = (char*)id;
  snprintf(name, 64, "/fakerccl-%d-%016llx", (int)getpid(), rnd);
  const int fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
  if (fd < 0) { perror("[fake_rccl] shm_open(create)"); return ncclSystemError; }
  const size_t bytes = round_up(sizeof(Segment), 4096);
  if (ftruncate(fd, (off_t)bytes) != 0) { perror("[fake_rccl] ftruncate"); close(fd); shm_unlink(name); return ncclSystemError; }
  close(fd);   // zero-filled: magic is written by the first rank to join
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* out, int nranks, ncclUniqueId id, int rank) {
  if (!out || nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  Comm* c = new Comm;
  c->rank = rank;
  c->world = nranks;
  memcpy(c->shm_name, &id, 63);
  const int fd = shm_open(c->shm_name, O_RDWR, 0600);
  if (fd < 0) { perror("[fake_rccl] shm_open(join)"); delete c; return ncclSystemError; }
  c->seg_bytes = round_up(sizeof(Segment), 4096);
  void* m = mmap(nullptr, c->seg_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (m == MAP_FAILED) { perror("[fake_rccl] mmap"); delete c; return ncclSystemError; }
  c->seg = (Segment*)m;
  HIP_OK(hipHostRegister(c->seg, c->seg_bytes, hipHostRegisterMapped));
  HIP_OK(hipHostGetDevicePointer((void**)&c->seg_dev, c->seg, 0));
  HIP_OK(hipHostMalloc((void**)&c->abort_flag, 64, hipHostMallocMapped));
  HIP_OK(hipHostMalloc((void**)&c->timed_out, 64, hipHostMallocMapped));
  *c->abort_flag = 0;
  *c->timed_out = 0;
  HIP_OK(hipHostGetDevicePointer((void**)&c->abort_dev, c->abort_flag, 0));
  const char* mb = getenv("FAKE_RCCL_MAILBOX_MB");
  c->mailbox_bytes = (size_t)(mb && atoi(mb) > 0 ? atoi(mb) : 192) << 20;
  HIP_OK(hipMalloc((void**)&c->mailbox, c->mailbox_bytes));
  Slot& me = c->seg->slot[rank];
  if (nranks > 1) HIP_OK(hipIpcGetMemHandle(&me.mailbox, c->mailbox));
  me.pid = (int32_t)getpid();
  c->seg->magic = MAGIC;
  c->seg->world = (uint32_t)nranks;
  me.joined.store(1);
  const double join_s = env_seconds("FAKE_RCCL_JOIN_TIMEOUT_S", 120.0);
  if (!wait_until([&] { for (int p = 0; p < nranks; ++p) if (!c->seg->slot[p].joined.load()) return false; return true; }, join_s)) {
    fprintf(stderr, "[fake_rccl] rank %d: peers did not join within %.0f s\n", rank, join_s);
    release(c, false);
    return ncclSystemError;
  }
  for (int p = 0; p < nranks; ++p) {
    if (p == rank) continue;
    hipIpcMemHandle_t h = c->seg->slot[p].mailbox;
    HIP_OK(hipIpcOpenMemHandle((void**)&c->peer_box[p], h, hipIpcMemLazyEnablePeerAccess));
  }
  me.opened.store(1);
  if (!wait_until([&] { for (int p = 0; p < nranks; ++p) if (!c->seg->slot[p].opened.load()) return false; return true; }, join_s)) {
    release(c, false);
    return ncclSystemError;
  }
  if (rank == 0) shm_unlink(c->shm_name);   // every rank holds its mapping; the name can go
  *out = (ncclComm_t)c;
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
  if (!comm) return ncclInvalidArgument;
  release((Comm*)comm, true);
  return ncclSuccess;
}

ncclResult_t ncclCommAbort(ncclComm_t comm) {
  if (!comm) return ncclInvalidArgument;
  Comm* c = (Comm*)comm;
  // fault injection (tests only): an abort that does NOT release the spinning kernels — what a transport without a working
  // ncclCommAbort looks like to comm_sync, which must then mark the context unusable instead of waiting for the stream
  if (const char* e = getenv("FAKE_RCCL_ABORT_FAILS"); e && e[0] == '1') return ncclSystemError;
  __atomic_store_n(c->abort_flag, 1u, __ATOMIC_RELEASE);   // every spinning kernel of this rank gives up
  release(c, false);                                       // (drains the stream first)
  return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t comm, int* count) {
  if (!comm || !count) return ncclInvalidArgument;
  *count = ((const Comm*)comm)->world;
  return ncclSuccess;
}

ncclResult_t ncclCommUserRank(const ncclComm_t comm, int* rank) {
  if (!comm || !rank) return ncclInvalidArgument;
  *rank = ((const Comm*)comm)->rank;
  return ncclSuccess;
}

ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t dt, ncclComm_t comm, hipStream_t st) {
  Comm* c = (Comm*)comm;
  const size_t tb = type_bytes(dt);
  if (!c || !tb || (count && (!send || !recv))) return ncclInvalidArgument;
  maybe_die(c, "allgather", ++c->n_allgather);
  return exchange(c, send, count * tb, recv, count * tb, false, st);
}

ncclResult_t ncclAllToAll(const void* send, void* recv, size_t count, ncclDataType_t dt, ncclComm_t comm, hipStream_t st) {
  Comm* c = (Comm*)comm;
  const size_t tb = type_bytes(dt);
  if (!c || !tb || (count && (!send || !recv))) return ncclInvalidArgument;
  maybe_die(c, "alltoall", ++c->n_alltoall);
  return exchange(c, send, count * tb * (size_t)c->world, recv, count * tb, true, st);
}

const char* ncclGetErrorString(ncclResult_t r) {
  switch (r) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "fake-rccl: HIP call failed";
    case ncclSystemError: return "fake-rccl: system error (shared memory / peers did not join)";
    case ncclInvalidArgument: return "fake-rccl: invalid argument";
    case ncclRemoteError: return "fake-rccl: a peer never arrived (device-side time-out of an earlier collective)";
    default: return "fake-rccl: error";
  }
}

}  // extern "C"
