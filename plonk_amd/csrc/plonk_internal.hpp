// Internal declarations shared by the HIP translation units of libplonk_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <map>
#include <mutex>
#include <set>
#include <vector>

#include "../../include/plonk_hip.h"
#include "api_guard.hpp"
#include "curve.cuh"
#include "msm_recode.cuh"

#define HIP_TRY(expr)                                                    \
  do {                                                                   \
    hipError_t _e = (expr);                                              \
    if (_e != hipSuccess) {                                              \
      plonk::set_last_error(#expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return PLONK_ERR_HIP;                                              \
    }                                                                    \
  } while (0)

namespace plonk {

void set_last_error(const char* what, const char* detail, const char* file, int line);
struct Ctx;
// A context whose collective timed out without its stream ever draining (comm_sync, Ctx::comm_poisoned) has a dead kernel at
// the head of its main stream: anything queued behind it never runs and anything that waits for the stream never returns.
// Every entry point that queues work on a context or waits for it therefore goes in through CTX_ENTER (the mutex + a
// refusal with PLONK_ERR_STATE), and the destroy paths ask ctx_abandon() whether the device side has to be leaked.
int ctx_refuse_poisoned(const char* api_fn);          // capi.hip: sets the last-error text, returns PLONK_ERR_STATE
void finish_pool_release(Ctx* c);                     // prover.hip: joins and deletes the context's host helper threads
bool ctx_abandon(Ctx* c);                             // capi.hip: poisoned AND the streams did not drain within a bounded poll
#define CTX_ENTER(C, FN)                      \
  std::lock_guard<std::mutex> lk((C).mu);     \
  if ((C).comm_poisoned) return plonk::ctx_refuse_poisoned(FN)

struct NttTables {
  Fr* tw_lo = nullptr;         // w_N^e, e < 2^min(L,13)
  Fr* tw_hi = nullptr;         // w_N^(e << 13)
  Fr* tw_lo_scaled = nullptr;  // n^-1 * w_N^e (inverse only)
  Fr* w512 = nullptr;          // w_512^e, e < 256
  Fr* g_lo = nullptr;          // g^i (fwd) or g^-i (inv), i < 1024
  Fr* g_hi = nullptr;          // g^(i << 10)
  // the same as w * 2^261 in 29-bit limbs (48-byte slots, fr29.cuh) for the pass kernels
  void* tw_lo29 = nullptr;
  void* tw_hi29 = nullptr;
  void* tw_lo_scaled29 = nullptr;
  void* w512_29 = nullptr;
  void* g_lo29 = nullptr;
  void* g_hi29 = nullptr;
  // whole inter-pass twiddle tables of the multi-pass plans (ntt.hip: NttPass::tw_direct), built for L <= NTT_DIRECT_MAX_LOG
  // when memory allows: pass A  first * w_N^e, e < N (first = n^-1 for the inverse);  pass B  w_N^(R1 j), j < N / R1
  void* tw_a29 = nullptr;
  void* tw_b29 = nullptr;
  Fr n_inv;
};

// MSM geometry: 2^15 signed-digit buckets shared by every table row.  Window tables (16 rows, 2^(16 w) * P_i): 16-bit
// signed windows; bit-position tables (256 rows, 2^r * P_i, round 3): width-17 NAF digits (msm_recode.cuh).  Either
// way a scalar yields at most MSM_W entries.
//
// The bucket count is a COMPILE-TIME constant of the MSM kernels (msm.hip, msm_sort.hip), and those two files are compiled
// twice: PLONK_MSM_NB_BITS = 15 (namespace nb15: 2^15 buckets, both recodings, every size) and a LARGE count (namespace
// nbl: 2^19 — or 2^18 — buckets for bit-position tables and large MSMs: width-21 / -20 NAF digits, 12.1 / 12.6 additions
// per scalar, one lane per bucket).  Round 4 built and measured a MEDIUM count too (namespace nbm: 2^17 buckets, width-19
// digits, 13.2 additions per scalar; opt-in A/B build): at 2^19 terms it loses to 2^19 buckets and at 2^18 to 2^15.
// msm_batch_device (msm.hip, compiled once) picks the variant per call; Ctx / MsmWork are shared and sized for the larger.
#ifndef PLONK_MSM_NB_BITS
#define PLONK_MSM_NB_BITS 15
#endif
#if PLONK_MSM_NB_BITS == 15
#define PLONK_MSM_NS nb15
#elif PLONK_MSM_NB_BITS == 17
#define PLONK_MSM_NS nbm          /* round 4 A/B build only (PLONK_BUILD_MSM_MEDIUM=1): 2^17 buckets, width-19 NAF digits (13.2 additions per
                                     scalar).  Measured at 2^19 terms against 2^15 / 2^19 buckets: no size where it wins (profiles/r04*) */
#elif PLONK_MSM_NB_BITS >= 18 && PLONK_MSM_NB_BITS <= 19
#define PLONK_MSM_NS nbl          /* the "large" variant; 2^18 / 2^19 buckets are both built for A/B (tools/build_variants.sh) */
#else
#error "PLONK_MSM_NB_BITS must be 15 or 17..19"
#endif
static constexpr int MSM_C = 16;                          // window tables: signed 16-bit windows (2^15 buckets only)
static constexpr int MSM_W = MSM_DIGITS;
static constexpr int MSM_NB_BITS = PLONK_MSM_NB_BITS;
static constexpr uint32_t MSM_NB = 1u << MSM_NB_BITS;     // buckets of THIS translation unit's variant
static constexpr int MSM_NB_BITS_MAX = 19;
static constexpr uint32_t MSM_NB_MAX = 1u << MSM_NB_BITS_MAX;   // what the shared work buffers are sized for
static constexpr uint32_t MSM_NAF_WIDTH = MSM_NB_BITS + 2;      // bit-position digits: odd, |d| < 2^(NB_BITS + 1), bucket = |d| >> 1
static constexpr int MSM_MAX_BATCH = 4;                  // commitments per group launch

struct MsmBatch {   // one commitment group: up to MSM_MAX_BATCH MSMs over the same bases, launched together
  const Fr* scalars[MSM_MAX_BATCH];
  uint64_t m[MSM_MAX_BATCH];
  G1* out[MSM_MAX_BATCH];
  int count;
  uint32_t ksl;   // entries per slice of this launch
  uint64_t cap_m, cap_slices;
  const void* table;   // tables the entries index: the context's commit key, or a prover's Lagrange-basis key
  uint64_t table_n;    // points per row of `table`
  uint32_t rows;       // MSM_ROWS_WINDOW (16), MSM_ROWS_BITPOS (256) or MSM_ROWS_HALFPOS (128): which recoding the entries come from
  uint32_t ordered;    // lanes of the accumulation in order of slice length: the sort also writes full_off / part_list
  uint32_t wide;       // the coarse-partitioned words are 64-bit (rows * table_n above 2^27)
  uint32_t heavy_thresh;   // a bucket with more slices than this is "heavy" (msm_slices_kernel lists it, msm.hip sums it by segments)
  // scalars of commitment k: scalars[k][i] for i < split[k], tail[k][i - split[k]] above (a wire column in place + its
  // blinders elsewhere); split[k] >= m[k] when the scalars are one array
  const Fr* tail[MSM_MAX_BATCH];
  uint64_t split[MSM_MAX_BATCH];
  // a group launched in parts (round 6: wire columns that arrive one after the other over PCIe): this launch covers the
  // `count` commitments [kb0, kb0 + count) of a group of `group_count`; scalars[] / m[] / out[] stay indexed by the GROUP's kb
  int kb0;
  int group_count;
};

struct HeavyItem { uint32_t bucket, seg_base, nseg, pad; };   // a bucket with far more slices than expected, cut into 256-slice segments

struct MsmWork {   // per-context scratch, grown on demand
  bool fixed_ok = false;           // every size-independent buffer below is allocated (msm_reserve: all or nothing)
  uint64_t cap_m = 0;
  uint32_t* tmp_words = nullptr;   // W * m words grouped by coarse bin (msm_sort.hip); room for 64-bit words
  uint32_t* entries = nullptr;     // W * m entries grouped by bucket
  uint32_t* coarse_cnt = nullptr;  // 2048 per commitment
  uint32_t* coarse_off = nullptr;  // 2049
  uint32_t* coarse_cur = nullptr;  // 2048
  uint32_t* big_off = nullptr;     // 2049: chunk prefix of the oversized coarse bins (skewed digits)
  uint32_t* big_cnt = nullptr;     // 2 x NB: bucket counts / run cursors of the oversized bins
  uint32_t* offsets = nullptr;     // NB + 1
  uint32_t* slice_off = nullptr;   // NB + 1
  uint32_t* full_off = nullptr;    // NB + 1: scan of the FULL slices per bucket (PLONK_MSM_ORDER=1: lanes in order of slice length)
  uint32_t* part_list = nullptr;   // NB + 1: buckets with a partial slice, longest first; [NB] = their number
  uint32_t* multi_list = nullptr;  // 2^19-bucket variant: buckets with 2 .. heavy_thresh slices (the only ones msm_bucket_sum visits)
  uint32_t* layout = nullptr;      // 2^19-bucket variant: block totals / remainder-class counts of the multi-workgroup layout pass
  uint64_t cap_slices = 0;
  void* partial = nullptr;         // slices x 256 B (XYZZ over Fp28, msm.hip)
  void* buckets = nullptr;         // NB
  uint32_t* nheavy = nullptr;      // per commitment: number of heavy buckets / of their segments in this launch; [2 KB + k]: listed multi-slice buckets
  void* heavy_list = nullptr;      // NB items per commitment (msm.hip HeavyItem)
  void* seg_sum = nullptr;         // segment sums of the heavy buckets
  uint64_t cap_segs = 0;
  void* chunk = nullptr;           // row / column sums
  int last_rowbits = 8;            // row bit sums written by the last msm_batch_device call (8: 2^15 buckets, 12: 2^19)
  uint8_t* result = nullptr;       // 97 B device
  uint8_t* result_host = nullptr;  // pinned
  Fr* scalars_stage = nullptr;     // H2D staging for the host-pointer API
  uint64_t cap_stage = 0;
};

// Effective configuration of a context: plonk_gpu_config with defaults resolved, then the environment overrides applied —
// ONCE, at plonk_ctx_create[_ex] / plonk_ctx_set_config (capi.hip config_resolve).  No other file of the library reads the
// environment for a behaviour switch (round 4 had 22 getenv sites, most of them latched in function statics at first use).
struct Config {
  // public fields (include/plonk_hip.h)
  uint64_t table_budget = 0;        // bytes all point tables of the context may take together (resolved: never 0)
  int table_mode = 0;               // 0 auto, else MSM_ROWS_WINDOW / _HALFPOS / _BITPOS
  int bucket_bits = 0;              // 0 by the number of terms, 15, 17 (A/B build), 19
  int quotient_domain = 4;          // 4 or 8
  int wire_commit_coeff = 0;        // 1: coefficient-form wire commitments
  int z_commit_coeff = 0;           // PLONK_Z_COMMIT=coeff (A/B): commit to z in coefficient form although the Lagrange table could take its evaluations (1); =evals: from its evaluations at every size (-1)
  int shard_quotient = 0;           // 0 default, 1 on, -1 off
  int shard_z = 0;
  int shard_side = 0;
  int ntt_elog = 0;                 // 0 default, 2, 3
  int comm_timeout_ms = 120000;
  int side_cus = 0;                 // CUs reserved for the side stream (0 = none)
  // kernel-tuning switches of the A/B scripts: environment only (DESIGN.md section 2 lists them)
  int ksl = 0;                      // PLONK_MSM_KSL: forced slice length
  bool prof_fine = false;           // PLONK_PROF_FINE=1: per-phase hipEvent slots
  bool acc_lds = false;             // PLONK_MSM_ACC=lds
  int order = -1;                   // PLONK_MSM_ORDER=0/1
  bool tail_serial = false;         // PLONK_MSM_TAIL=serial
  bool bsum_lane = false;           // PLONK_MSM_BSUM=lane
  bool rc_lane_tree = false;        // PLONK_MSM_RCTREE=lane
  int lps = 0;                      // PLONK_MSM_LPS
  int rcwv = 0;                     // PLONK_MSM_RCWV
  int sort13 = -1;                  // PLONK_MSM_SORT13=0/1: 13-digit staging of the 2^19-bucket partition (round 5)
  int acc_wg = 0;                   // PLONK_MSM_ACC_WG=64: threads per workgroup of the ordered accumulation (A/B)
  bool ntt_direct = true;           // PLONK_NTT_DIRECT=0 switches the whole inter-pass twiddle tables off
  int bi_cfg = -1;                  // PLONK_BI_CFG=0..3: batch-inversion geometry
  int side_defer = -1;              // PLONK_SIDE_DEFER=0/1/2
  int wire_by_column = 0;           // PLONK_WIRE_BY_COLUMN: 0 -> -1 (host wire columns commit as ONE grouped launch after the last copy, round 5), 1 / 2 -> by column at every size (a, b, c + d / one launch each); unset: by column from 2^19 gates on
  int wire_polys_side = -1;         // PLONK_WIRE_POLYS_SIDE=0/1 (A/B): the wire inverse transforms on the side stream under their commitments; unset: up to 2^18 gates
  int side_after_elog = 0;          // PLONK_SIDE_AFTER_ELOG=2/3: pass geometry of side transforms issued after a group's accumulation
  int host_threads = -1;            // PLONK_HOST_THREADS=k: helper threads for the host arithmetic between device phases (finish_pool.hpp); -1: 3 when the process may run on >= 8 CPUs, else 0
};

struct plonk_msm_plan_internal {   // what msm_batch_device chose (mirrors plonk_msm_plan)
  uint32_t table_rows = 0, bucket_bits = 0, digit_width = 0, slice_entries = 0, ordered_lanes = 0, wide_words = 0, flags = 0;
  uint64_t terms = 0;
  const char* kernel = "";
};

struct Ctx {
  int device = 0;
  Config cfg;
  uint64_t srs_table_alloc = 0;        // bytes of srs_table
  uint64_t table_bytes = 0;            // bytes of point tables this context holds (commit key + Lagrange-basis keys of its provers)
  plonk_msm_plan_internal last_plan;   // of the last msm_batch_device call (plonk_ctx_last_msm)
  hipStream_t stream = nullptr;    // every launch goes to this stream (prover.hip swaps it for side work)
  hipStream_t main_stream = nullptr;
  hipStream_t side_stream = nullptr;   // low priority: challenge-independent NTTs overlapped with MSM phases
  hipEvent_t acc_done = nullptr;       // when set: recorded by msm_batch_device right after its msm_accumulate launch (prover.hip gates side work on it)
  int ntt_elog_hint = 0;               // 0: ntt.hip's default; 2 / 3: elements per lane (log2) of the pass kernels launched while it is set (prover.hip SideScope)
  std::mutex mu;         // serialises entry points (reference calls concurrently from rayon)
  std::mutex table_mu;
  std::map<uint32_t, NttTables*> ntt_tables;
  std::set<const void*> smem_opt_in;   // kernels whose dynamic-LDS limit was raised on THIS device (hipFuncSetAttribute is per device)
  // NTT staging for the host-pointer API
  Fr* ntt_buf = nullptr;
  Fr* ntt_buf2 = nullptr;          // second and third transform buffer of plonk_ntt_batch's upload / compute / download pipeline
  Fr* ntt_buf3 = nullptr;
  hipStream_t copy_stream = nullptr;
  hipStream_t down_stream = nullptr;   // device -> host leg of that pipeline
  Fr* ntt_tmp = nullptr;
  uint64_t ntt_cap = 0;
  // SRS
  void* srs_table = nullptr;       // [srs_rows][npoints] 128-B affine entries (Fp28): 2^(16 w) * P_i (16 rows), 2^r * P_i (256 rows) or 4^r * P_i (128 rows)
  uint32_t srs_rows = 0;
  void* table_scratch = nullptr;   // srs_table_kernel's per-window ZZ / ZZZ / running products, alive during a key load
  uint64_t table_scratch_pts = 0;
  uint64_t srs_n = 0;
  uint64_t srs_gen = 0;            // bumped by every (re)load: provers remember the generation they were built on
  MsmWork msm;
  // multi-GPU (comm.hip): RCCL communicator of this rank, staging for small all-gathers
  void* nccl_comm = nullptr;
  uint8_t* comm_send = nullptr;
  uint8_t* comm_recv = nullptr;
  uint8_t* comm_send_host = nullptr;   // pinned twins of the two staging buffers: the host legs of a small all-gather never block
  uint8_t* comm_recv_host = nullptr;   // inside hipMemcpyAsync behind a collective (comm.hip comm_allgather_host)
  int comm_rank = 0, comm_world = 1;
  const char* comm_warning = "";   // static text left by plonk_comm_init (plonk_comm_warning); never the last-error string
  bool comm_poisoned = false;      // comm_sync timed out and the stream never drained: sharded proofs and new communicators are refused
  bool comm_loopback = false;      // measurement only: collectives return the rank's own contribution (plonk_comm_measure_loopback)
  void* finish_pool = nullptr;     // FinishPool* (finish_pool.hpp), created by the first commitment group of a prover (prover.hip), freed by finish_pool_release
  // instrumentation: hipEvent pairs around the dominant kernels
  bool profile = false;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  static constexpr int PROF_SLOTS = 32;   // 0-7: the bench line's slots; 8-10: HOST time of prove() (prover.hip HostGap), 11: helper threads per commitment group (a count); 16-21 / 24-29: MSM phases of groups of >= 3 / <= 2 commitments (PLONK_PROF_FINE=1)
  double acc_ms[PROF_SLOTS] = {0};
  uint64_t acc_n[PROF_SLOTS] = {0};
};

// ntt.hip
struct NttCoset {   // powers of an arbitrary coset shift (two-level tables, both forms) — see ntt_coset_tables
  Fr* g_lo = nullptr;
  Fr* g_hi = nullptr;
  void* g_lo29 = nullptr;
  void* g_hi29 = nullptr;
};
int ntt_coset_tables(Ctx* c, uint32_t L, const Fr& shift, bool inverse, NttCoset* out);
void ntt_coset_free(NttCoset* t);
// shift == nullptr: the multiplicative generator 7 (the reference's coset)
int ntt_device(Ctx* c, const Fr* src, Fr* dst, Fr* tmp, uint32_t L, bool inverse, bool coset, uint64_t in_len,
               const NttCoset* shift = nullptr);
int ntt_tables(Ctx* c, uint32_t L, bool inverse, NttTables** out);
// raise a kernel's dynamic shared-memory limit once per context (= per device)
inline void smem_opt_in(Ctx* c, const void* fn, size_t bytes) {
  if (c->smem_opt_in.insert(fn).second) (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}
void ntt_plan(uint32_t L, int r[3], int* npass);

// comm.hip: who this rank is and, when the context has no RCCL communicator, the host all-gather to use
struct CommLink {
  int rank = 0, world = 1;
  plonk_allgather_fn fn = nullptr;
  void* user = nullptr;
};
int comm_allgather_host(Ctx* c, const CommLink& l, const void* send, void* recv, size_t bytes);
int comm_sync(Ctx* c, hipStream_t st);   // stream wait that cannot hang on a dead peer (polls, aborts the communicator on time-out)
int comm_alltoall_dev(Ctx* c, const CommLink& l, const void* send_dev, void* recv_dev, size_t bytes_per_peer);
int comm_allgather_dev(Ctx* c, const CommLink& l, void* buf_dev, size_t bytes_per_rank);   // in place: rank r's part at buf + r * bytes

// msm.hip
int srs_load_device(Ctx* c, const G1Affine* pts_dev, uint64_t n);
int srs_table_begin(Ctx* c, uint64_t n);
int srs_table_chunk(Ctx* c, const G1Affine* pts_dev, uint64_t n, uint64_t first, uint64_t count, hipStream_t st);
void srs_table_scratch_free(Ctx* c);   // once the stream that built the tables was synchronised
int srs_validate_device(Ctx* c, const G1Affine* pts_dev, uint64_t n, int* flag_dev);
int srs_export_device(Ctx* c, G1Affine* out_dev);   // the context's commit key (srs_n points) as x || y Montgomery limbs
int msm_device(Ctx* c, const Fr* scalars_dev, uint64_t m, G1* out_xyzz_dev);
// bit_sums = false: out[k] = the commitment (one XYZZ point).  true: out[k][0..16) = partial sums the
// host combines with a short doubling chain (msm.hip msm_bits_kernel, prover.hip finish_bit_sums).
// [16] = the sum of ALL buckets: bit-position entries weigh 2 b + 1, so their commitment is 2 W - S (hostg1.hpp).
// Layout with rb = log2(rows) row bits (8 for 2^15 buckets, 12 for 2^19): rows [0, rb), columns [rb, rb + 7), C_128, S.
static constexpr int MSM_BIT_SUMS = 12 + 9;          // slots per commitment (the larger variant's count)
// variant entry points (msm.hip / msm_sort.hip, one set per compiled bucket count)
#define PLONK_MSM_VARIANT_DECLS                                                                                          \
  int msm_order_slices(Ctx* c, const MsmBatch& bt);                                                                      \
  int msm_group_sort(Ctx* c, const MsmBatch& bt, uint64_t mmax);                                                         \
  bool msm_needs_wide_words(uint32_t rows, uint64_t table_n);                                                            \
  int msm_batch_device_v(Ctx* c, MsmBatch& bt, uint64_t mmax, bool bit_sums, int phase);                                 \
  uint32_t msm_ksl(const Ctx* c, uint64_t m);                                                                            \
  bool msm_acc_ordered(const Ctx* c, uint32_t ksl);
namespace nb15 { PLONK_MSM_VARIANT_DECLS }
#ifdef PLONK_MSM_WITH_MEDIUM
namespace nbm { PLONK_MSM_VARIANT_DECLS int msm_buckets_bits(); }
#endif
namespace nbl { PLONK_MSM_VARIANT_DECLS int msm_buckets_bits(); }
// table == nullptr: the context's commit key; otherwise tables built by srs_table_build (same layout, table_rows rows).
// c->msm.last_rowbits tells the caller how the bit sums of THIS call are laid out (finish_bit_sums).
int msm_batch_device(Ctx* c, const Fr* const* scalars_dev, const uint64_t* m, int count, G1* const* out_xyzz_dev,
                     bool bit_sums = false, const void* table = nullptr, uint64_t table_n = 0,
                     const Fr* const* tail_dev = nullptr, const uint64_t* split = nullptr, uint32_t table_rows = 0,
                     int phase = 3, int kb0 = 0, int kcount = -1);
// phase / kb0 / kcount (round 6): a group launched in parts.  phase 1 = bucket sort + accumulation + bucket sums of the
// commitments [kb0, kb0 + kcount) only; phase 2 = the group's reduction tail over all `count`; 3 = the ordinary grouped launch.
// The caller passes the SAME arrays (all `count` entries) to every part and runs the parts on one stream, phase 2 last.
// rows of the tables of an n-point key — 256 (one per bit position), 128 (every second) or the 16 window rows — from the
// context's table budget and what it already holds (msm.hip); Config::table_mode forces one
// last_key: the key is built after everything else the context needs (a prover's Lagrange-basis key) and may take what is
// left of the budget; the commit key (false) leaves room for what follows
uint32_t msm_table_rows(const Ctx* c, uint64_t n, bool last_key);
uint64_t msm_table_bytes(uint32_t rows, uint64_t n);
// what msm_batch_device would choose for `count` sets of at most mmax terms over a key of table_rows rows / table_n points
void msm_plan(const Ctx* c, uint32_t table_rows, uint64_t table_n, uint64_t mmax, int count, bool bit_sums, plonk_msm_plan_internal* out);
void config_resolve(const plonk_gpu_config* user, int device, Config* out);   // capi.hip
// tables for n points given as G1Affine (caller frees *table_out with hipFree); *rows_out = rows chosen
int srs_table_build(Ctx* c, const G1Affine* pts_dev, uint64_t n, void** table_out, uint32_t* rows_out);
void srs_table_release(Ctx* c, void* table, uint32_t rows, uint64_t n);   // frees a srs_table_build table and returns its bytes to the budget
// [L_i(tau)] G for the size-n domain (n = 2^L) from the context's commit key (needs n + 2 points), followed by the
// two blinding points [tau^n] G - G and [tau^(n+1)] G - [tau] G: n + 2 affine points (an EC inverse FFT)
int lagrange_points_device(Ctx* c, uint32_t L, G1Affine* out_dev);
int lagrange_blind_points_device(Ctx* c, uint64_t n, uint32_t k0, uint32_t cnt, G1Affine* out_dev);   // [tau^(n+k)] G - [tau^k] G, k = k0 .. k0 + cnt - 1
int xyzz_to_affine97_device(Ctx* c, const G1* in_dev, uint8_t* out97_dev);
// host-side affine normalisation of an XYZZ result: out = x || y || infinity flag
void xyzz_to_affine97_host(const G1& p, uint8_t out[97]);
int msm_reserve(Ctx* c, uint64_t m);
int msm_sort_reserve_fixed(Ctx* c);

}  // namespace plonk

struct plonk_ctx {
  plonk::Ctx c;
};
