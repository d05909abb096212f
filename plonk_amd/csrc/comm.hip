// Multi-GPU exchange layer of libplonk_hip.so: one process per GPU, RCCL over xGMI.
//
// The prover needs two collectives:
//   * all-gather of a few hundred bytes per rank (MSM partial sums, partial evaluations, suffix-sum
//     carries) — EC addition / field addition is done locally after the gather, because a group
//     element is not an RCCL reduction type;
//   * one all-to-all of the per-class quotient remainders before the coefficient recombination
//     (the transpose of a four-step transform), 32 * n / world bytes per peer.
// Both run on the context's main stream.  RCCL is resolved with dlopen at plonk_comm_init, so the
// library loads (and the single-GPU path runs) on hosts without librccl.  A context without a
// communicator falls back to the caller-supplied host all-gather callback of plonk_prover_desc
// (tests drive several ranks on ONE device through it, which RCCL itself refuses).
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include <string>

#include "plonk_internal.hpp"

namespace plonk {

struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;                 // optional: the abort path of comm_sync
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;     // optional: plonk_comm_info
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllToAll)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

// The transport library.  Default: librccl from the loader path / /opt/rocm/lib.  plonk_comm_set_library names another file
// BEFORE the first communicator call of the process: a deployment whose RCCL lives elsewhere — and the test suite's
// stand-in (tests/fake_rccl: hipIpc collectives between ranks that share one GPU, which RCCL itself refuses).  Whatever
// was loaded is reported by plonk_comm_library, so a benchmark line can never pass a stand-in off as RCCL.
static std::mutex g_rccl_mu;
static std::string g_rccl_path;       // explicit path ("" = search the default names)
static std::string g_rccl_loaded;     // what dlopen resolved (dladdr of ncclAllGather)

static RcclApi* rccl_api() {
  static RcclApi api;
  std::lock_guard<std::mutex> lk(g_rccl_mu);
  if (api.lib) return &api;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
  void* h = nullptr;
  if (!g_rccl_path.empty()) h = dlopen(g_rccl_path.c_str(), RTLD_NOW | RTLD_LOCAL);
  else
    for (const char* nm : names)
      if ((h = dlopen(nm, RTLD_NOW | RTLD_LOCAL))) break;
  if (!h) { set_last_error(g_rccl_path.empty() ? "dlopen(librccl)" : "dlopen(plonk_comm_set_library path)", dlerror(), __FILE__, __LINE__); return nullptr; }
  api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(h, "ncclGetUniqueId");
  api.CommInitRank = (decltype(api.CommInitRank))dlsym(h, "ncclCommInitRank");
  api.CommDestroy = (decltype(api.CommDestroy))dlsym(h, "ncclCommDestroy");
  api.AllGather = (decltype(api.AllGather))dlsym(h, "ncclAllGather");
  api.AllToAll = (decltype(api.AllToAll))dlsym(h, "ncclAllToAll");
  api.GetErrorString = (decltype(api.GetErrorString))dlsym(h, "ncclGetErrorString");
  api.CommAbort = (decltype(api.CommAbort))dlsym(h, "ncclCommAbort");
  api.CommCount = (decltype(api.CommCount))dlsym(h, "ncclCommCount");
  api.CommUserRank = (decltype(api.CommUserRank))dlsym(h, "ncclCommUserRank");
  if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllGather || !api.AllToAll) {
    set_last_error("dlsym(librccl)", "missing RCCL entry point", __FILE__, __LINE__);
    dlclose(h);
    return nullptr;
  }
  Dl_info di;
  g_rccl_loaded = (dladdr((void*)api.AllGather, &di) && di.dli_fname) ? di.dli_fname : (g_rccl_path.empty() ? "librccl" : g_rccl_path);
  api.lib = h;
  return &api;
}

#define RCCL_TRY(api, expr)                                                                         \
  do {                                                                                              \
    ncclResult_t _r = (expr);                                                                       \
    if (_r != ncclSuccess) {                                                                        \
      set_last_error(#expr, (api)->GetErrorString ? (api)->GetErrorString(_r) : "rccl error", __FILE__, __LINE__); \
      return PLONK_ERR_HIP;                                                                         \
    }                                                                                               \
  } while (0)

static constexpr size_t COMM_STAGE = 16384;   // bytes per rank of a small (host-value) all-gather

// The host driver of this platform only supports dmabuf IPC: without HSA_ENABLE_IPC_MODE_LEGACY=0 in the environment
// BEFORE the HSA runtime starts, RCCL's device-memory exchange between processes fails (hipIpcGetMemHandle: invalid
// argument).  The variable is read when the runtime initialises — normally the first HIP call of the process — so it is
// the HOST PROGRAM has to export it before its first HIP call (bench.py and the Python binding do; include/plonk_hip.h,
// "Multi-GPU").  The library no longer touches the process environment when it is loaded (round 3 did, from a
// constructor): plonk_comm_init only REPORTS a missing variable when the communicator cannot be created.

// Wait for the main stream after a collective was queued on it.  A rank that failed (or never arrived) leaves its peers
// inside the collective for ever, and a plain hipStreamSynchronize would hang with them: with a communicator the stream
// is POLLED, and after PLONK_COMM_TIMEOUT_MS (default 120 s) the communicator is aborted (ncclCommAbort) and the call
// returns PLONK_ERR_STATE, so that every surviving rank gets an error instead of a hang.
static long comm_timeout_ms(const Ctx* c) { return c->cfg.comm_timeout_ms > 0 ? c->cfg.comm_timeout_ms : 120000L; }   // plonk_gpu_config.comm_timeout_ms
int comm_sync(Ctx* c, hipStream_t st) {
  if (!c->nccl_comm) { HIP_TRY(hipStreamSynchronize(st)); return PLONK_OK; }
  const auto t0 = std::chrono::steady_clock::now();
  bool slow = false;
  for (uint32_t spin = 0;; ++spin) {
    const hipError_t e = hipStreamQuery(st);
    if (e == hipSuccess) return PLONK_OK;
    if (e != hipErrorNotReady) { set_last_error("hipStreamQuery", hipGetErrorString(e), __FILE__, __LINE__); return PLONK_ERR_HIP; }
    // busy poll for the first 40 ms — a rank's longest device phase at 2^20 gates is under 10 ms, and a sleep's wake-up
    // latency (50-100 us) would be paid at each of the five synchronisations of a proof; after that (2^22 gates and two
    // ranks, or a peer that is late or dead) sleep between polls
    if ((spin & 255) == 255) slow = std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(40);
    if (slow) std::this_thread::sleep_for(std::chrono::microseconds(50));
    if ((spin & 1023) == 1023 &&
        std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count() > comm_timeout_ms(c)) {
      RcclApi* api = rccl_api();
      const bool aborted = api && api->CommAbort && api->CommAbort((ncclComm_t)c->nccl_comm) == ncclSuccess;
      c->nccl_comm = nullptr;   // aborted: the context falls back to "no communicator" and every later sharded call fails loudly
      // the abort releases the collective's kernel; what was queued BEHIND it on this stream (copies into caller-owned host
      // buffers) must not still be running when the error reaches the caller.  The drain is BOUNDED: without a
      // successful abort — or if the abort does not retire the kernel — a blocking synchronise would be the very hang this
      // function exists to prevent; the stream is polled for 10 s and, if it never drains, the context is marked unusable.
      bool drained = false;
      if (aborted) {
        const auto t1 = std::chrono::steady_clock::now();
        while (std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t1).count() < 10000) {
          if (hipStreamQuery(st) != hipErrorNotReady) { drained = true; break; }
          std::this_thread::sleep_for(std::chrono::microseconds(200));
        }
      }
      if (!drained) {
        c->comm_poisoned = true;   // work of a dead collective may still sit on the stream: every later entry point of this context refuses
        set_last_error("comm_sync", aborted ? "collective timed out; the communicator was aborted but the stream did not drain: context unusable"
                                            : "collective timed out and the transport has no working abort: context unusable", __FILE__, __LINE__);
        return PLONK_ERR_STATE;
      }
      set_last_error("comm_sync", "collective timed out (a peer rank failed or never joined): communicator aborted", __FILE__, __LINE__);
      return PLONK_ERR_STATE;
    }
  }
}

// All-gather of `bytes` host bytes per rank; recv is rank-major.  RCCL when the context has a
// communicator (staged through device buffers on the main stream), else the host callback.
// Loop-back — MEASUREMENT ONLY (tools/rank_alone.py): every collective hands a rank its own contribution back in place of
// its peers', with local copies and no transport, so that ONE rank of a W-rank job can be timed alone on one GPU (its
// kernels, its share of the points and coefficients, the same host sequence).  The values exchanged are wrong by
// construction: prove() ends in PLONK_ERR_UNSAT at its final identity check, after all of its work.  Enabled per context by
// an explicit call (plonk_comm_measure_loopback); an environment variable alone no longer switches it on (round 3's
// PLONK_COMM_LOOPBACK made every collective of a production process silently local).
static bool comm_loopback(const Ctx* c) { return c->comm_loopback; }

int comm_allgather_host(Ctx* c, const CommLink& l, const void* send, void* recv, size_t bytes) {
  if (l.world <= 1) { memcpy(recv, send, bytes); return PLONK_OK; }
  if (comm_loopback(c)) {
    for (int r = 0; r < l.world; ++r) memcpy((uint8_t*)recv + bytes * (size_t)r, send, bytes);
    return PLONK_OK;
  }
  if (c->nccl_comm) {
    RcclApi* api = rccl_api();
    if (!api || bytes > COMM_STAGE) return (set_last_error("comm_allgather_host", "message too large / no rccl", __FILE__, __LINE__), PLONK_ERR_ARG);
    // Both host legs go through PINNED staging buffers.  Round 5, found by the peer-failure test on the stand-in transport: with
    // the caller's pageable `recv` as the target, hipMemcpyAsync(DeviceToHost) does not return until the stream has reached the
    // copy — i.e. the host sat INSIDE that call behind the collective, where no time-out polls, and a dead peer hung the rank
    // (the stand-in's kernels give up after a minute; RCCL's never do).  Pinned copies are queued and return; the wait is
    // comm_sync's poll.
    memcpy(c->comm_send_host, send, bytes);
    HIP_TRY(hipMemcpyAsync(c->comm_send, c->comm_send_host, bytes, hipMemcpyHostToDevice, c->main_stream));
    RCCL_TRY(api, api->AllGather(c->comm_send, c->comm_recv, bytes, ncclUint8, (ncclComm_t)c->nccl_comm, c->main_stream));
    HIP_TRY(hipMemcpyAsync(c->comm_recv_host, c->comm_recv, bytes * (size_t)l.world, hipMemcpyDeviceToHost, c->main_stream));
    const int rs = comm_sync(c, c->main_stream);
    if (rs) return rs;
    memcpy(recv, c->comm_recv_host, bytes * (size_t)l.world);
    return PLONK_OK;
  }
  if (!l.fn) return (set_last_error("comm_allgather_host", "no communicator and no all-gather callback", __FILE__, __LINE__), PLONK_ERR_STATE);
  if (l.fn(l.user, send, recv, bytes) != 0) return (set_last_error("all-gather callback", "returned non-zero", __FILE__, __LINE__), PLONK_ERR_STATE);
  return PLONK_OK;
}

// All-gather of device buffers IN PLACE: rank r's contribution already sits at buf + r * bytes_per_rank; afterwards every
// rank holds all of them (round 4: the z evaluations of a sharded grand product).  RCCL: ncclAllGather in place on the main
// stream; callback transport: the slice goes through the host all-gather.
int comm_allgather_dev(Ctx* c, const CommLink& l, void* buf_dev, size_t bytes_per_rank) {
  if (l.world <= 1 || bytes_per_rank == 0) return PLONK_OK;
  uint8_t* buf = (uint8_t*)buf_dev;
  if (comm_loopback(c)) {
    // Peer r's place gets this rank's slice ROTATED by 32 r bytes (round 6, second session).  Until then it got the slice
    // itself, so the gathered array was periodic: its inverse transform — the z polynomial of a sharded grand product — had
    // W - 1 of every W coefficients ZERO, and the rank's z commitment measured an eighth of its real accumulation.
    const uint8_t* mine = buf + bytes_per_rank * (size_t)l.rank;
    for (int r = 0; r < l.world; ++r) {
      if (r == l.rank) continue;
      const size_t rot = (32 * (size_t)(r + 1)) % bytes_per_rank;
      HIP_TRY(hipMemcpyAsync(buf + bytes_per_rank * (size_t)r, mine + rot, bytes_per_rank - rot, hipMemcpyDeviceToDevice, c->main_stream));
      if (rot) HIP_TRY(hipMemcpyAsync(buf + bytes_per_rank * (size_t)r + (bytes_per_rank - rot), mine, rot, hipMemcpyDeviceToDevice, c->main_stream));
    }
    return PLONK_OK;
  }
  if (c->nccl_comm) {
    RcclApi* api = rccl_api();
    if (!api) return PLONK_ERR_STATE;
    RCCL_TRY(api, api->AllGather(buf + bytes_per_rank * (size_t)l.rank, buf, bytes_per_rank, ncclUint8, (ncclComm_t)c->nccl_comm, c->main_stream));
    return PLONK_OK;
  }
  if (!l.fn) return (set_last_error("comm_allgather_dev", "no communicator and no all-gather callback", __FILE__, __LINE__), PLONK_ERR_STATE);
  std::vector<uint8_t> hs(bytes_per_rank), hr(bytes_per_rank * (size_t)l.world);
  HIP_TRY(hipMemcpyAsync(hs.data(), buf + bytes_per_rank * (size_t)l.rank, bytes_per_rank, hipMemcpyDeviceToHost, c->main_stream));
  HIP_TRY(hipStreamSynchronize(c->main_stream));
  if (l.fn(l.user, hs.data(), hr.data(), bytes_per_rank) != 0) return (set_last_error("all-gather callback", "returned non-zero", __FILE__, __LINE__), PLONK_ERR_STATE);
  HIP_TRY(hipMemcpyAsync(buf, hr.data(), hr.size(), hipMemcpyHostToDevice, c->main_stream));
  HIP_TRY(hipStreamSynchronize(c->main_stream));   // hr leaves scope
  return PLONK_OK;
}

// All-to-all of device buffers: send = [peer][bytes_per_peer], recv = [source][bytes_per_peer].
// Callback transport: every rank contributes its whole send buffer to an all-gather and keeps the
// pieces addressed to it (world x the traffic — it is the functional fallback, not the fast path).
int comm_alltoall_dev(Ctx* c, const CommLink& l, const void* send_dev, void* recv_dev, size_t bytes_per_peer) {
  if (l.world <= 1) {
    HIP_TRY(hipMemcpyAsync(recv_dev, send_dev, bytes_per_peer, hipMemcpyDeviceToDevice, c->main_stream));
    return PLONK_OK;
  }
  if (comm_loopback(c)) {
    // Source `src`'s place gets the block this rank addressed TO src — W different blocks (round 6, second session).  Until
    // then every source's place got the block the rank keeps for itself: W identical class residues, whose recombination
    // leaves ONE non-zero part of the quotient and three ZERO ones, so three of the four t commitments of a rank measured
    // empty (tools/rank_alone.py read 0.8 ms too little for a rank of 8 at 2^20 gates).
    for (int src = 0; src < l.world; ++src)
      HIP_TRY(hipMemcpyAsync((uint8_t*)recv_dev + bytes_per_peer * (size_t)src, (const uint8_t*)send_dev + bytes_per_peer * (size_t)src,
                             bytes_per_peer, hipMemcpyDeviceToDevice, c->main_stream));
    return PLONK_OK;
  }
  if (c->nccl_comm) {
    RcclApi* api = rccl_api();
    if (!api) return PLONK_ERR_STATE;
    RCCL_TRY(api, api->AllToAll(send_dev, recv_dev, bytes_per_peer, ncclUint8, (ncclComm_t)c->nccl_comm, c->main_stream));
    return PLONK_OK;
  }
  if (!l.fn) return (set_last_error("comm_alltoall_dev", "no communicator and no all-gather callback", __FILE__, __LINE__), PLONK_ERR_STATE);
  const size_t mine = bytes_per_peer * (size_t)l.world;
  std::vector<uint8_t> hs(mine), hr(mine * (size_t)l.world);
  HIP_TRY(hipMemcpyAsync(hs.data(), send_dev, mine, hipMemcpyDeviceToHost, c->main_stream));
  HIP_TRY(hipStreamSynchronize(c->main_stream));
  if (l.fn(l.user, hs.data(), hr.data(), mine) != 0) return (set_last_error("all-gather callback", "returned non-zero", __FILE__, __LINE__), PLONK_ERR_STATE);
  for (int src = 0; src < l.world; ++src)
    HIP_TRY(hipMemcpyAsync((uint8_t*)recv_dev + bytes_per_peer * (size_t)src,
                           hr.data() + mine * (size_t)src + bytes_per_peer * (size_t)l.rank, bytes_per_peer, hipMemcpyHostToDevice, c->main_stream));
  HIP_TRY(hipStreamSynchronize(c->main_stream));   // hr leaves scope
  return PLONK_OK;
}

}  // namespace plonk

using namespace plonk;

extern "C" {

int plonk_comm_measure_loopback(plonk_ctx* ctx, int on) {
  const char* const api_fn = __func__;
  return plonk::api_guard(api_fn, [&]() -> int {
  if (!ctx) return PLONK_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->c.mu);
  ctx->c.comm_loopback = on != 0;
  return PLONK_OK;
  });
}

int plonk_comm_set_library(const char* path) {
  const char* const api_fn = __func__;
  return plonk::api_guard(api_fn, [&]() -> int {
  if (!path || !path[0]) return PLONK_ERR_ARG;
  std::lock_guard<std::mutex> lk(g_rccl_mu);
  if (!g_rccl_loaded.empty()) return (set_last_error("plonk_comm_set_library", "a transport library is already loaded in this process", __FILE__, __LINE__), PLONK_ERR_STATE);
  g_rccl_path = path;
  return PLONK_OK;
  });
}

int plonk_comm_library(char* out, uint64_t cap) {
  const char* const api_fn = __func__;
  return plonk::api_guard(api_fn, [&]() -> int {
  if (!out || cap == 0) return PLONK_ERR_ARG;
  std::lock_guard<std::mutex> lk(g_rccl_mu);
  if (g_rccl_loaded.empty()) return (set_last_error("plonk_comm_library", "no transport library loaded yet", __FILE__, __LINE__), PLONK_ERR_STATE);
  snprintf(out, (size_t)cap, "%s", g_rccl_loaded.c_str());
  return PLONK_OK;
  });
}

int plonk_comm_unique_id(uint8_t out[128]) {
  const char* const api_fn = __func__;
  return plonk::api_guard(api_fn, [&]() -> int {
  if (!out) return PLONK_ERR_ARG;
  RcclApi* api = rccl_api();
  if (!api) return PLONK_ERR_STATE;
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId size");
  ncclUniqueId id;
  RCCL_TRY(api, api->GetUniqueId(&id));
  memcpy(out, &id, 128);
  return PLONK_OK;
  });
}

int plonk_comm_init(plonk_ctx* ctx, const uint8_t id128[128], int rank, int world) {
  const char* const api_fn = __func__;
  return plonk::api_guard(api_fn, [&]() -> int {
  if (!ctx || !id128 || world < 1 || rank < 0 || rank >= world) return PLONK_ERR_ARG;
  Ctx& c = ctx->c;
  std::lock_guard<std::mutex> lk(c.mu);
  HIP_TRY(hipSetDevice(c.device));
  if (c.nccl_comm) return (set_last_error("plonk_comm_init", "context already has a communicator", __FILE__, __LINE__), PLONK_ERR_STATE);
  if (c.comm_poisoned) return (set_last_error("plonk_comm_init", "context unusable: a collective timed out and its stream never drained", __FILE__, __LINE__), PLONK_ERR_STATE);
  RcclApi* api = rccl_api();
  if (!api) return PLONK_ERR_STATE;
  ncclUniqueId id;
  memcpy(&id, id128, 128);
  ncclComm_t comm = nullptr;
  {
    const ncclResult_t r = api->CommInitRank(&comm, world, id, rank);
    if (r != ncclSuccess) {
      const char* ipc = getenv("HSA_ENABLE_IPC_MODE_LEGACY");
      std::string msg = api->GetErrorString ? api->GetErrorString(r) : "rccl error";
      if (world > 1 && !(ipc && ipc[0] == '0'))
        msg += " (HSA_ENABLE_IPC_MODE_LEGACY=0 is not in the environment: on a dmabuf-only driver the host program must export it before its first HIP call)";
      set_last_error("ncclCommInitRank", msg.c_str(), __FILE__, __LINE__);
      return PLONK_ERR_HIP;
    }
  }
  (void)hipFree(c.comm_send);   // left over when comm_sync aborted the previous communicator
  (void)hipFree(c.comm_recv);
  c.comm_send = c.comm_recv = nullptr;
  (void)hipHostFree(c.comm_send_host);
  (void)hipHostFree(c.comm_recv_host);
  c.comm_send_host = c.comm_recv_host = nullptr;
  hipError_t e = hipMalloc((void**)&c.comm_send, COMM_STAGE);
  if (e == hipSuccess) e = hipMalloc((void**)&c.comm_recv, COMM_STAGE * (size_t)world);
  if (e == hipSuccess) e = hipHostMalloc((void**)&c.comm_send_host, COMM_STAGE, hipHostMallocDefault);
  if (e == hipSuccess) e = hipHostMalloc((void**)&c.comm_recv_host, COMM_STAGE * (size_t)world, hipHostMallocDefault);
  if (e != hipSuccess) {
    (void)api->CommDestroy(comm);
    (void)hipFree(c.comm_send);
    (void)hipFree(c.comm_recv);
    (void)hipHostFree(c.comm_send_host);
    c.comm_send = c.comm_recv = c.comm_send_host = nullptr;
    set_last_error("hipMalloc(comm staging)", hipGetErrorString(e), __FILE__, __LINE__);
    return PLONK_ERR_HIP;
  }
  c.nccl_comm = comm;
  c.comm_rank = rank;
  c.comm_world = world;
  // the communicator can come up without the variable and fail later, inside an IPC exchange nobody
  // annotates.  A call that SUCCEEDS must not leave text in the thread's last-error string (a later failure that sets none
  // would report it): the note is kept per context and read with plonk_comm_warning.
  {
    const char* ipc = getenv("HSA_ENABLE_IPC_MODE_LEGACY");
    c.comm_warning = (world > 1 && !(ipc && ipc[0] == '0'))
        ? "HSA_ENABLE_IPC_MODE_LEGACY=0 is not in the environment; on a dmabuf-only driver device-memory exchange between "
          "processes fails later with hipIpcGetMemHandle: invalid argument — export it before the first HIP call"
        : "";
  }
  return PLONK_OK;
  });
}

int plonk_comm_destroy(plonk_ctx* ctx) {
  const char* const api_fn = __func__;
  return plonk::api_guard(api_fn, [&]() -> int {
  if (!ctx) return PLONK_ERR_ARG;
  Ctx& c = ctx->c;
  std::lock_guard<std::mutex> lk(c.mu);
  if (!c.nccl_comm && !c.comm_send && !c.comm_recv && !c.comm_send_host) return PLONK_OK;
  (void)hipSetDevice(c.device);
  if (ctx_abandon(&c)) {
    // the dead collective still holds the stream: waiting for it, hipFree (device-wide wait) or ncclCommDestroy would hang.
    // The staging buffers are leaked on purpose — the copies queued behind the collective may still target them.
    c.nccl_comm = nullptr;
    c.comm_send = c.comm_recv = c.comm_send_host = c.comm_recv_host = nullptr;
    c.comm_world = 1;
    c.comm_rank = 0;
    set_last_error("plonk_comm_destroy", "context unusable (collective time-out, stream never drained): staging buffers abandoned, not freed", __FILE__, __LINE__);
    return PLONK_ERR_STATE;
  }
  (void)hipStreamSynchronize(c.main_stream);
  if (c.nccl_comm) {   // (already gone after a time-out abort in comm_sync: only the staging buffers are left to free)
    RcclApi* api = rccl_api();
    if (api) (void)api->CommDestroy((ncclComm_t)c.nccl_comm);
  }
  c.nccl_comm = nullptr;
  (void)hipFree(c.comm_send);
  (void)hipFree(c.comm_recv);
  (void)hipHostFree(c.comm_send_host);
  (void)hipHostFree(c.comm_recv_host);
  c.comm_send = c.comm_recv = c.comm_send_host = c.comm_recv_host = nullptr;
  c.comm_world = 1;
  c.comm_rank = 0;
  return PLONK_OK;
  });
}

// The note plonk_comm_init left on this context ("" = none); NUL-terminated, truncated to cap.
int plonk_comm_warning(plonk_ctx* ctx, char* out, uint64_t cap) {
  const char* const api_fn = __func__;
  return plonk::api_guard(api_fn, [&]() -> int {
  if (!ctx || !out || cap == 0) return PLONK_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->c.mu);
  snprintf(out, (size_t)cap, "%s", ctx->c.comm_warning);
  return PLONK_OK;
  });
}

// What the communicator itself reports (ncclCommUserRank / ncclCommCount): bench.py prints it as n_ranks_rccl.
int plonk_comm_info(plonk_ctx* ctx, int* rank, int* world) {
  const char* const api_fn = __func__;
  return plonk::api_guard(api_fn, [&]() -> int {
  if (!ctx) return PLONK_ERR_ARG;
  Ctx& c = ctx->c;
  std::lock_guard<std::mutex> lk(c.mu);
  if (!c.nccl_comm) return (set_last_error("plonk_comm_info", "no communicator", __FILE__, __LINE__), PLONK_ERR_STATE);
  RcclApi* api = rccl_api();
  int r = c.comm_rank, w = c.comm_world;
  if (api && api->CommCount && api->CommUserRank) {
    RCCL_TRY(api, api->CommCount((ncclComm_t)c.nccl_comm, &w));
    RCCL_TRY(api, api->CommUserRank((ncclComm_t)c.nccl_comm, &r));
  }
  if (rank) *rank = r;
  if (world) *world = w;
  return PLONK_OK;
  });
}

// Both collectives once, with a rank-dependent pattern, checked on every rank.
int plonk_comm_selftest(plonk_ctx* ctx) {
  const char* const api_fn = __func__;
  return plonk::api_guard(api_fn, [&]() -> int {
  if (!ctx) return PLONK_ERR_ARG;
  Ctx& c = ctx->c;
  CTX_ENTER(c, api_fn);
  HIP_TRY(hipSetDevice(c.device));
  if (!c.nccl_comm) return (set_last_error("plonk_comm_selftest", "no communicator", __FILE__, __LINE__), PLONK_ERR_STATE);
  CommLink l;
  l.rank = c.comm_rank;
  l.world = c.comm_world;
  const int W = l.world, R = l.rank;
  uint8_t send[64];
  for (int i = 0; i < 64; ++i) send[i] = (uint8_t)(R * 7 + i);
  std::vector<uint8_t> recv(64 * (size_t)W);
  int rc = comm_allgather_host(&c, l, send, recv.data(), 64);
  if (rc) return rc;
  for (int r = 0; r < W; ++r)
    for (int i = 0; i < 64; ++i)
      if (recv[64 * r + i] != (uint8_t)(r * 7 + i)) return (set_last_error("plonk_comm_selftest", "all-gather returned wrong bytes", __FILE__, __LINE__), PLONK_ERR_STATE);
  const size_t per = 4096;
  uint8_t *ds = nullptr, *dr = nullptr;
  HIP_TRY(hipMalloc((void**)&ds, per * W));
  if (hipMalloc((void**)&dr, per * W) != hipSuccess) { (void)hipFree(ds); return PLONK_ERR_HIP; }
  std::vector<uint8_t> hs(per * W), hr(per * W);
  for (int p = 0; p < W; ++p)
    for (size_t i = 0; i < per; ++i) hs[per * p + i] = (uint8_t)(R * 31 + p * 5 + i);
  rc = PLONK_OK;
  if (hipMemcpyAsync(ds, hs.data(), per * W, hipMemcpyHostToDevice, c.main_stream) != hipSuccess) rc = PLONK_ERR_HIP;
  if (!rc) rc = comm_alltoall_dev(&c, l, ds, dr, per);
  if (!rc && hipMemcpyAsync(hr.data(), dr, per * W, hipMemcpyDeviceToHost, c.main_stream) != hipSuccess) rc = PLONK_ERR_HIP;
  { const int rs = comm_sync(&c, c.main_stream); if (rs) rc = rc ? rc : rs; }
  (void)hipFree(ds);
  (void)hipFree(dr);
  if (rc) return rc;
  for (int src = 0; src < W; ++src)
    for (size_t i = 0; i < per; ++i)
      if (hr[per * src + i] != (uint8_t)(src * 31 + R * 5 + i)) return (set_last_error("plonk_comm_selftest", "all-to-all returned wrong bytes", __FILE__, __LINE__), PLONK_ERR_STATE);
  return PLONK_OK;
  });
}

}  // extern "C"
