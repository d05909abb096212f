// Device-resident Prover::prove (V3) around the NTT / MSM kernels.
//
// Host driver restating reference src/compiler/prover.rs:415-761 (prove_inner),
// Prover::new (:53-115) and the parts of Compiler::preprocess that derive cached
// proving state from the ProverKey polynomials (src/compiler.rs:310-425).  The host
// only runs the Fiat-Shamir transcript and O(1) scalar algebra; every O(n) step is a
// kernel on arrays that stay in HBM (ntt.hip, msm.hip, poly.hip).  Witness generation
// (Composer::prove, composer.rs:442) and the RNG stay with the caller: the prover takes
// the padded wire columns and the 11 blinding scalars in the reference's draw order.
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "poly.hpp"
#include "transcript.hpp"
#include <chrono>

#include "hostg1.hpp"
#include "finish_pool.hpp"
#include "widgets.hpp"
#include "permutation.hpp"

static_assert(plonk::WQS_RANGE == plonk::QS_RANGE && plonk::WQS_LOGIC == plonk::QS_LOGIC && plonk::WQS_FIXED == plonk::QS_FIXED &&
              plonk::WQS_VAR == plonk::QS_VAR && plonk::WQS_COUNT == plonk::QS_COUNT, "widgets.hpp selector ids");

namespace plonk {

// transcript labels in VerifierKey::seed_transcript order (widget.rs:229-254)
static const int VK_ORDER[15] = {P_QM, P_QL, P_QR, P_QO, P_QC, P_QF, P_QARITH, P_QRANGE, P_QLOGIC,
                                 P_QVAR, P_QFIXED, P_S1, P_S2, P_S3, P_S4};
static const char* VK_LABEL[15] = {"q_m", "q_l", "q_r", "q_o", "q_c", "q_f", "q_arith", "q_range", "q_logic",
                                   "q_variable_group_add", "q_fixed_group_add",
                                   "s_sigma_1", "s_sigma_2", "s_sigma_3", "s_sigma_4"};

struct Prover {
  Ctx* c = nullptr;
  int transcript_version = 3;      // PlonkVersion of the transcript seeding: 3 (Transcript::base_v3) or the legacy 2 (plonk_prover_set_version)
  uint64_t n = 0, n8 = 0, np = 0, constraints = 0;   // n8 = quotient-domain size = qf * n
  uint32_t logn = 0;
  uint32_t qf = 8, lq = 3;         // quotient domain: 8n (the reference's) or 4n + de-aliasing, see quotient_low()
  Fr n_inv, inv32, edwards_d, gq_inv, omega, omega_inv;   // proof-independent host constants (inversions are ~25 us each)
  Fr key_low[P_COUNT][7];          // lowest 7 coefficients of the key polynomials (host copy)
  Fr* low_host = nullptr;          // pinned: lowest 7 coefficients of a, b, c, d, z, pi
  hipEvent_t ev_pi = nullptr;
  std::string label;
  uint8_t vk[15][48];              // compressed commitments in PolyId order
  Fr* polys = nullptr;             // [P_COUNT][np] coefficient form
  uint64_t poly_len[P_COUNT] = {0};
  Fr* evals8 = nullptr;            // [P_COUNT + 2][n8]: coset evals of the 15 polys, linear, l1
  Fr* sigma_n = nullptr;           // [4][n] sigma evaluations over the proving domain
  Fr vinv[8];
  bool has[QS_COUNT];
  // per-proof work buffers
  Fr* wires = nullptr;             // [4][n] evaluations (device copy)
  Fr* wpoly = nullptr;             // [4][np] blinded coefficient polys a, b, c, d
  Fr* zpoly = nullptr;             // [np]
  Fr* pipoly = nullptr;            // [np]
  Fr* cos = nullptr;               // [6][n8] coset evals of z, a, b, c, d, pi
  Fr* tbuf = nullptr;              // [n8 + 16] quotient evals -> coefficients
  Fr* tmp8 = nullptr;              // [n8] NTT scratch (main stream)
  Fr* tmp8b = nullptr;             // [n8] NTT scratch (side stream)
  hipEvent_t ev_ready = nullptr, ev_side = nullptr;
  hipEvent_t ev_acc = nullptr;   // end of a commitment group's msm_accumulate (deferred side work starts there)
  hipEvent_t ev_wire[4] = {nullptr, nullptr, nullptr, nullptr};   // host-wire uploads in flight on the copy stream (plonk_prover_prove)
  bool wires_pending = false;
  uint32_t last_wire_launches = 0;   // of the last proof's wire group: 1 grouped, 3 = a, b, c + d, 4 = one per column (plonk_prover_describe)
  Fr* tparts = nullptr;            // [3][np] t_low, t_mid, t_high
  Fr* agg = nullptr;               // [np] linear combination
  Fr* wit = nullptr;               // [np] opening witness polynomial W_z
  Fr* wit2 = nullptr;              // [np] W_zw
  Fr* scratch = nullptr;           // [2 * np] (perm num / den, ruffini scratch)
  Fr* totals = nullptr;            // scan block totals
  Fr* evpart = nullptr;            // eval partials
  Fr* evout = nullptr;             // 16 evaluations
  uint8_t* res = nullptr;          // 16 x 256 B MSM results, XYZZ (device)
  uint8_t* res_host = nullptr;     // pinned mirror
  bool res_bitpos[16] = {};        // slot holds the sums of bit-position entries (finish_bit_sums: 2 W - S)
  int res_rowbits[16] = {};        // row bit sums in the slot (8: 2^15 buckets, 12: 2^19; msm_batch_device reports it per call)
  uint8_t* gather_host = nullptr;  // world x 16 x 192 B all-gathered partial sums
  // multi-GPU: this rank owns SRS points [shard_lo, shard_lo + c->srs_n) of srs_total
  int rank = 0, world = 1;
  uint64_t srs_total = 0, shard_lo = 0;
  uint64_t srs_gen = 0;            // Ctx::srs_gen at creation: the SRS this prover's degree checks / slices refer to
  CommLink link;                   // rank / world / host all-gather callback (comm.hip picks RCCL when the ctx has a communicator)
  // ---- sharded quotient (multi-GPU, world in {2, 4, 8}; SURVEY §8e ii).  The quotient coset {g w_N^i}, N = Q n,
  // is the union of Q size-n cosets (g w_N^j) H_n ("classes"); rank r owns classes r, r + world, ...: its
  // key / wire evaluations, the point-wise pass and a size-n inverse transform per class are local, one
  // all-to-all + a Q-point inverse DFT per coefficient turn the per-class remainders into the coefficient
  // range [lo, hi) of t this rank's MSM shard needs; rounds 4-5 work on the same coefficient range.
  bool sharded = false;
  uint32_t Q = 0, cpr = 0;         // classes in total (4, or 8 for world == 8) / per rank
  uint32_t cls[8] = {0};           // owned class indices
  NttCoset cs_fwd[8], cs_inv[8];   // coset tables for the class shifts (forward / inverse)
  Fr sigma_j[8];                   // x^n on the class: g^n w_Q^j
  uint64_t qn = 0;                 // elements per evaluation array: cpr * n (sharded) or n8
  uint64_t per = 0, lo = 0, hi = 0, stride = 0;   // coefficient range owned: [lo, hi), hi <= np - 1; message stride
  Fr coef[5][8];                   // inverse-DFT factors, ShardCombineArgs::coef
  Fr* fold = nullptr;              // [2][n] folded polynomials (main / side stream)
  Fr* Fbuf = nullptr;              // [cpr][n] per-class quotient evaluations -> remainders
  Fr* send = nullptr;              // [world][cpr][stride]
  Fr* recv = nullptr;
  // Lagrange-basis commit key of the proving domain (single GPU): the wire commitments are taken from the wire
  // VALUES, [L_i(tau)] G tables + the two blinding points, so small witness values cost few additions (zero
  // digits never reach the accumulation) and the wire iNTTs leave the critical path
  void* lag_table = nullptr;       // window tables over n + 2 points (lagrange_points_device), or over this rank's slice of them
  uint64_t lag_n = 0;              // points in lag_table
  uint32_t lag_rows = 0;           // its rows: 256 (a row per bit position) or 16 (window rows)
  bool lag_on = false;             // wire commitments over the Lagrange key (lag_n may be 0 for a rank without a slice)
  bool lag_z = false;              // single GPU: the table holds a third blinding point ([tau^(n+2)] G - [tau^2] G): z is committed from its evaluations too
  hipEvent_t ev_zlow = nullptr;    // z's lowest coefficients have reached low_host (they come from the side stream when lag_z)
  bool lag_whole = false;          // sharded prover holding the WHOLE Lagrange key: the wire group is split by commitment, not by point range
  Fr* wscal = nullptr;             // [11] the wire blinders (8) and z's (3) on the device: tail scalars of the Lagrange-key MSMs
  Fr* agg2 = nullptr;              // [np] second linear combination (W_zw numerator)
  Fr* scratch2 = nullptr;          // [np + 1]
  plonk_allgather_fn allgather = nullptr;
  void* allgather_user = nullptr;
  Fr* ev_host = nullptr;           // pinned 16 Fr
  unsigned long long* len_dev = nullptr;
  unsigned long long* len_host = nullptr;
  int* flag_dev = nullptr;
  int* flag_host = nullptr;
  uint64_t* pi_idx_dev = nullptr;
  Fr* pi_val_dev = nullptr;
  uint64_t pi_cap = 0;
  uint32_t ev_max_blocks = 0;
  // built by plonk_compile: the witness index behind every wire, so a proof can start from the witness values
  uint32_t* wire_idx = nullptr;    // [4][constraints]
  uint64_t witnesses = 0;
  Fr* wit_vals = nullptr;          // [witnesses] staging of one proof's values
};

// ---- host helpers -------------------------------------------------------------------
static Fr omega_of(uint32_t L) {
  Fr g = fr_root_of_unity();
  for (uint32_t i = L; i < 32; ++i) g = g.sqr();
  return g;
}



#define PTRY(expr) do { int _rc = (expr); if (_rc) return _rc; } while (0)

// Work that does not depend on a Fiat-Shamir challenge (the coset FFTs of the wire, public-input
// and z polynomials) is issued on the context's low-priority side stream so it fills the
// bandwidth-/latency-bound stretches of the MSM pipeline.  While a SideScope is alive every
// launch helper (they all use c->stream) targets the side stream.  These transforms run UNDER the sort, accumulation and
// reduction kernels of the main stream: they use the 8-elements-per-lane pass kernels (two waves per SIMD, ~234 VGPRs), which
// only take the CUs the main stream leaves free — the 4-element kernels (ntt.hip's default) co-reside with the main stream's
// kernels and slow the critical path by more than they gain (profiles/r04/log_r4v.txt).
struct SideScope {
  Ctx* c;
  explicit SideScope(Ctx* ctx, hipEvent_t wait_for) : c(ctx) {
    (void)hipEventRecord(wait_for, c->main_stream);
    (void)hipStreamWaitEvent(c->side_stream, wait_for, 0);
    c->stream = c->side_stream;
    // side stream confined to CUs of its own by a CU mask (plonk_gpu_config.side_stream_cus, round 5): the four-wave kernels
    // can no longer spread over the chip and tax the critical path, so they keep ntt.hip's default geometry
    c->ntt_elog_hint = c->cfg.side_cus > 0 ? 0 : 3;
  }
  ~SideScope() { c->stream = c->main_stream; c->ntt_elog_hint = 0; }
};
// Side work that must not start before an event ALREADY recorded on the main stream (the end of a group's
// msm_accumulate): the transforms then fill the latency-bound tail of the group and the host synchronisation instead
// of competing with the bandwidth-bound sort and the VALU-bound accumulation.
struct SideScopeAfter {
  Ctx* c;
  SideScopeAfter(Ctx* ctx, hipEvent_t recorded) : c(ctx) {
    (void)hipStreamWaitEvent(c->side_stream, recorded, 0);
    c->stream = c->side_stream;
    // the deferred scope starts under the latency-bound TAIL of a group, not under its accumulation; which pass
    // geometry suits it is measured, not assumed (PLONK_SIDE_AFTER_ELOG=2|3 for the A/B; default: ntt.hip's own choice,
    // the state the round-4 numbers were taken in — profiles/r05/SUMMARY.md section 3 has the same-box comparison)
    c->ntt_elog_hint = c->cfg.side_after_elog;
  }
  ~SideScopeAfter() { c->stream = c->main_stream; c->ntt_elog_hint = 0; }
};
struct AccMark {   // msm_batch_device records `ev` after its accumulate launch while this is alive
  Ctx* c;
  AccMark(Ctx* ctx, hipEvent_t ev) : c(ctx) { c->acc_done = ev; }
  ~AccMark() { c->acc_done = nullptr; }
};
struct SideJoin {   // never leave side work in flight when prove() returns (buffers are reused)
  Ctx* c;
  ~SideJoin() { (void)hipStreamSynchronize(c->side_stream); }
};

// Host time of a proof (slots 8-10 of plonk_profile_read, only while profiling is on): what the HOST does while the device
// waits for it.  prove() has five points where the transcript needs device results before the next kernels can be queued
// (four commitment groups and the 15 evaluations): slot 10 = time blocked in those synchronisations (the device is
// working), slot 8 = the host arithmetic that turns bit sums into compressed commitments (finish_bit_sums, the shared
// inversion, compression), slot 9 = from the return of a synchronisation to the next launch on the main stream (slot 8
// included) — the device's main stream is idle for exactly that long, plus the launch latency.
struct HostGap {
  using clock = std::chrono::steady_clock;
  Ctx* c;
  clock::time_point t_sync;
  bool open = false;
  static double ms(clock::time_point a, clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); }
  void synced(clock::time_point before) {   // a synchronisation just returned
    if (!c->profile) return;
    t_sync = clock::now();
    open = true;
    prof_host_add(c, 10, ms(before, t_sync));
  }
  void launching() {                        // the next main-stream launch follows
    if (!c->profile || !open) return;
    open = false;
    prof_host_add(c, 9, ms(t_sync, clock::now()));
  }
};
static thread_local HostGap* tl_gap = nullptr;   // the proof in flight on this thread (fetch_commitments reports through it)
struct GapScope {   // (the last gap of a proof — after the opening commitments — closes at return)
  HostGap* g;
  explicit GapScope(HostGap* x) : g(x) { tl_gap = g; }
  ~GapScope() { g->launching(); tl_gap = nullptr; }
};

static constexpr int RES_STRIDE = MSM_BIT_SUMS * (int)sizeof(G1);   // 17 bit sums per commitment (msm_bits_kernel)

// CommitKey::commit (key.rs:376-388) on the rank's slice of the SRS: points
// [shard_lo, shard_lo + srs_n) against the matching scalars; partial sums are combined in
// fetch_commitments.  With world == 1 this is the whole MSM.
// phase / kb0 / kcount: a group over a prover-owned key launched in parts (msm_batch_device) — the wire columns of
// plonk_prover_prove as they arrive over PCIe
static int msm_group(Prover* p, const Fr* const* scalars, const uint64_t* m, int count, int first_slot,
                     const void* table = nullptr, uint64_t table_n = 0, const Fr* const* tail = nullptr, const uint64_t* split = nullptr,
                     int phase = 3, int kb0 = 0, int kcount = -1) {
  const Fr* sc[MSM_MAX_BATCH];
  uint64_t cnt[MSM_MAX_BATCH];
  G1* out[MSM_MAX_BATCH];
  if (table) {   // a prover-owned key (Lagrange basis, single GPU): no point-range sharding
    for (int k = 0; k < count; ++k) { out[k] = (G1*)(p->res + RES_STRIDE * (first_slot + k)); p->res_bitpos[first_slot + k] = p->lag_rows == MSM_ROWS_BITPOS; }
    const int rc = msm_batch_device(p->c, scalars, m, count, out, true, table, table_n, tail, split, p->lag_rows, phase, kb0, kcount);
    for (int k = 0; k < count; ++k) p->res_rowbits[first_slot + k] = p->c->msm.last_rowbits;
    return rc;
  }
  for (int k = 0; k < count; ++k) {
    if (m[k] > p->srs_total) return PLONK_ERR_DEGREE;   // check_commit_degree_is_within_bounds, key.rs:362-370
    const uint64_t lo = p->shard_lo;
    uint64_t hi = lo + p->c->srs_n;
    if (hi > m[k]) hi = m[k];
    cnt[k] = hi > lo ? hi - lo : 0;
    sc[k] = scalars[k] + (cnt[k] ? lo : 0);
    out[k] = (G1*)(p->res + RES_STRIDE * (first_slot + k));
    p->res_bitpos[first_slot + k] = p->c->srs_rows == MSM_ROWS_BITPOS;
  }
  const int rc = msm_batch_device(p->c, sc, cnt, count, out, true);
  for (int k = 0; k < count; ++k) p->res_rowbits[first_slot + k] = p->c->msm.last_rowbits;
  return rc;
}
static int msm_to(Prover* p, const Fr* scalars, uint64_t m, int slot) { return msm_group(p, &scalars, &m, 1, slot); }
static int fetch_commitments(Prover* p, int first, int count, uint8_t (*out48)[48], uint32_t local_mask = 0);

// A Lagrange-basis key handed in by the caller (plonk_prover_desc.lagrange_xy96, kept from an earlier plonk_lagrange_key)
// must be THE Lagrange form of the context's commit key — a key cached from another SRS would give wire commitments that
// disagree with every other commitment of the proof, and the prover's own identity check would not notice.  Checked the
// way the wire commitments use it: for pseudo-random values r_0 .. r_{n+1},
//     sum_{i<n} r_i [L_i(tau)] G + r_n ([tau^n] G - G) + r_{n+1} ([tau^(n+1)] G - [tau] G)   (an MSM over the supplied key)
// must be the commitment of interpolate(r_0 .. r_{n-1}) + (r_n + r_{n+1} X)(X^n - 1) over the context's key — one inverse
// transform and two MSMs, once per prover.  The r_i are 64-bit values derived from the index AND a per-check seed drawn
// from the operating system's entropy source (std::random_device) — with a fixed public mix (round 3) a crafted key could
// cancel its errors against known coefficients; with the seed a wrong point survives with probability ~2^-64.
__device__ __host__ static inline uint64_t lag_check_mix(uint64_t i, uint64_t seed) {
  uint64_t x = (i + 1) * 0x9e3779b97f4a7c15ull ^ seed;
  x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull; x ^= x >> 27; x *= 0x94d049bb133111ebull; x ^= x >> 31;
  return x | 1ull;
}
__global__ void lag_check_fill_kernel(Fr* r, uint64_t count, uint64_t seed) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) r[i] = Fr::from_u64(lag_check_mix(i, seed));
}
// Sharded provers (round 6): the same identity over the ranks.  Every rank fills the same r (rank 0's seed,
// all-gathered); a rank holding its SLICE of the key sums r over it and the partial sums are added like any sharded
// commitment; a rank holding the WHOLE key (wire group by column) sums all of it and compares its own sum (local_mask).  The
// coefficient-form commitment is sharded by point range as always.  So the union of the slices — or every rank's copy of
// the whole key — is checked against the union of the commit-key slices, which round 5 could not do ("only the points
// themselves can be checked here").
static int check_lagrange_key(Prover* p, uint32_t L) {
  Ctx* c = p->c;
  const uint64_t n = 1ull << L;
  Fr* r = p->agg;        // n + 2 values
  Fr* coef = p->wit;     // n + 2 coefficients
  std::random_device rd;
  uint64_t seed = ((uint64_t)rd() << 32) ^ (uint64_t)rd() ^ ((uint64_t)rd() << 17);
  if (p->world > 1) {
    PTRY(comm_allgather_host(c, p->link, &seed, p->gather_host, sizeof seed));
    memcpy(&seed, p->gather_host, sizeof seed);   // rank 0's
  }
  hipLaunchKernelGGL(lag_check_fill_kernel, dim3((uint32_t)((n + 2 + 255) / 256)), dim3(256), 0, c->stream, r, n + 2, seed);
  HIP_TRY(hipGetLastError());
  const bool slice = p->world > 1 && !p->lag_whole;
  uint64_t m = slice ? p->lag_n : n + 2;
  const Fr* sc = slice ? r + p->shard_lo : r;
  if (m && p->lag_table) {
    PTRY(msm_group(p, &sc, &m, 1, 0, p->lag_table, p->lag_n));
  } else {   // an empty slice: the identity (all-zero bit sums)
    HIP_TRY(hipMemsetAsync(p->res, 0, (size_t)RES_STRIDE, c->stream));
    p->res_bitpos[0] = false;
    p->res_rowbits[0] = 8;
  }
  PTRY(ntt_device(c, r, coef, p->wit2, L, true, false, n));
  BlindArgs ba;
  ba.count = 2;
  ba.b[0] = Fr::from_u64(lag_check_mix(n, seed));
  ba.b[1] = Fr::from_u64(lag_check_mix(n + 1, seed));
  ba.b[2] = Fr::zero();
  PTRY(poly_blind(c, coef, n, ba));     // coef -= (b0 + b1 X), coef[n], coef[n + 1] = b0, b1: + (b0 + b1 X)(X^n - 1)
  PTRY(msm_to(p, coef, n + 2, 1));
  uint8_t out[2][48];
  PTRY(fetch_commitments(p, 0, 2, out, p->lag_whole ? 1u : 0u));
  bool bad = memcmp(out[0], out[1], 48) != 0;
  if (p->world > 1 && p->lag_whole) {   // whole keys are compared per rank: share the verdict, so that no rank walks on into a collective its failed peer never enters
    uint8_t mine = bad ? 1 : 0;
    PTRY(comm_allgather_host(c, p->link, &mine, p->gather_host, 1));
    for (uint32_t r = 0; r < p->world; ++r) bad = bad || p->gather_host[r] != 0;
  }
  if (bad)
    return (plonk::set_last_error("lagrange_xy96", "not the Lagrange-basis form of this context's commit key", __FILE__, __LINE__), PLONK_ERR_DATA);
  return PLONK_OK;
}
// Bring `count` results to the host, all-gather the per-rank partial sums (EC addition is not
// an RCCL reduction, so the "bucket-sum all-reduce" is an all-gather + local add), normalise
// to affine on the host (one Fp inversion each) and compress.
// local_mask: bit i set = slot first + i is NOT summed over the ranks — it holds a sum this rank computed whole (the
// Lagrange-key check of a rank that holds the whole key); the all-gather still runs (every rank calls in the same order).
// The context's host helper threads (finish_pool.hpp): created by the first commitment group that has more than one
// commitment, joined by plonk_ctx_destroy.
static FinishPool* finish_pool_of(Ctx* c) {
  if (!c->finish_pool && c->cfg.host_threads > 0) c->finish_pool = new FinishPool(c->cfg.host_threads);
  return (FinishPool*)c->finish_pool;
}
void finish_pool_release(Ctx* c) {
  delete (FinishPool*)c->finish_pool;
  c->finish_pool = nullptr;
}
struct FinishJob {
  const Prover* p;
  int first;
  G1* sums;
};
static void finish_task(void* arg, int i) {
  const FinishJob* j = (const FinishJob*)arg;
  j->sums[i] = finish_bit_sums(reinterpret_cast<const G1*>(j->p->res_host + RES_STRIDE * (j->first + i)), j->p->res_rowbits[j->first + i], j->p->res_bitpos[j->first + i]);
}
static int fetch_commitments(Prover* p, int first, int count, uint8_t (*out48)[48], uint32_t local_mask) {
  Ctx* c = p->c;
  HIP_TRY(hipMemcpyAsync(p->res_host + RES_STRIDE * first, p->res + RES_STRIDE * first, RES_STRIDE * (size_t)count,
                         hipMemcpyDeviceToHost, c->stream));
  // the Horner chains of the group's commitments are independent: helper threads are woken NOW, while this thread blocks in
  // the synchronisation, and take their share as soon as the bit sums are here (finish_pool.hpp)
  FinishPool* const pool = count >= 2 ? finish_pool_of(c) : nullptr;
  if (tl_gap) prof_host_add(c, 11, pool ? pool->workers() : 0);   // slot 11: helper threads of this group (a count, not a time)
  Armed helpers(pool);
  // a sharded proof queues collectives on this stream (the quotient's all-to-all precedes the t commitments): never a
  // blocking wait behind one — comm_sync polls and aborts the communicator on time-out (a dead peer must not hang the rest)
  const auto t_before = HostGap::clock::now();
  if (p->world > 1) PTRY(comm_sync(c, c->stream));
  else HIP_TRY(hipStreamSynchronize(c->stream));
  if (tl_gap) tl_gap->synced(t_before);
  const auto t_fin = HostGap::clock::now();
  std::vector<G1> sums(count);
  FinishJob job{p, first, sums.data()};
  helpers.run(finish_task, &job, count);
  if (p->world > 1) {
    const size_t bytes = sizeof(G1) * (size_t)count;
    PTRY(comm_allgather_host(c, p->link, sums.data(), p->gather_host, bytes));
    for (int i = 0; i < count; ++i) {
      if (local_mask >> i & 1) continue;
      sums[i] = h1_sum_strided(p->gather_host + sizeof(G1) * i, bytes, (int)p->world);
    }
  }
  uint8_t aff[16][97];
  batch_xyzz_to_affine97(sums.data(), count, aff);
  for (int i = 0; i < count; ++i) g1_compress97(aff[i], out48[i]);
  if (tl_gap) prof_host_add(c, 8, HostGap::ms(t_fin, HostGap::clock::now()));
  return PLONK_OK;
}

static void prover_free(Prover* p) {
  if (!p) return;
  void* bufs[] = {p->polys, p->evals8, p->sigma_n, p->wires, p->wpoly, p->zpoly, p->pipoly, p->cos, p->tbuf, p->tmp8, p->tmp8b,
                  p->tparts, p->agg, p->wit, p->wit2, p->scratch, p->totals, p->evpart, p->evout, p->res, p->len_dev,
                  p->flag_dev, p->pi_idx_dev, p->pi_val_dev};
  for (void* b : bufs) if (b) (void)hipFree(b);
  for (void* b : {(void*)p->fold, (void*)p->Fbuf, (void*)p->send, (void*)p->recv, (void*)p->agg2, (void*)p->scratch2, (void*)p->wscal}) if (b) (void)hipFree(b);
  srs_table_release(p->c, p->lag_table, p->lag_rows, p->lag_n);   // gives the bytes back to the context's table budget
  for (void* b : {(void*)p->wire_idx, (void*)p->wit_vals}) if (b) (void)hipFree(b);
  for (int k = 0; k < 8; ++k) { ntt_coset_free(&p->cs_fwd[k]); ntt_coset_free(&p->cs_inv[k]); }
  if (p->ev_ready) (void)hipEventDestroy(p->ev_ready);
  if (p->ev_side) (void)hipEventDestroy(p->ev_side);
  if (p->ev_acc) (void)hipEventDestroy(p->ev_acc);
  if (p->ev_pi) (void)hipEventDestroy(p->ev_pi);
  if (p->ev_zlow) (void)hipEventDestroy(p->ev_zlow);
  for (int k = 0; k < 4; ++k) if (p->ev_wire[k]) (void)hipEventDestroy(p->ev_wire[k]);
  if (p->low_host) (void)hipHostFree(p->low_host);
  if (p->res_host) (void)hipHostFree(p->res_host);
  if (p->gather_host) free(p->gather_host);
  if (p->ev_host) (void)hipHostFree(p->ev_host);
  if (p->len_host) (void)hipHostFree(p->len_host);
  if (p->flag_host) (void)hipHostFree(p->flag_host);
  delete p;
}

struct BuildGuard {   // every early return of prover_build releases what was allocated so far
  Prover* p;
  ~BuildGuard() { if (p) prover_free(p); }
};

// What Compiler::preprocess starts from (compiler.rs:132-175): the composer's gates as columns
struct CircuitSrc {
  const Fr* selectors[QS_COUNT];   // per-gate selector values, nullptr = identically zero
  const uint32_t* wires[4];        // witness index on wires a, b, c, d of every gate
  uint64_t witnesses;
};

struct DevFree {   // scratch of one build step
  void* p = nullptr;
  ~DevFree() { if (p) (void)hipFree(p); }
};

// Compiler::preprocess steps 1-2 (compiler.rs:139-207) on the device: the selector columns are uploaded as they are
// (zero beyond the last gate) and interpolated in place; the sigma mappings are worked out on the host in one pass over
// the gates (permutation.hpp), turned into K_col * w^row evaluations by a kernel (permutation.rs:141-175) and
// interpolated the same way.  Leaves p->polys / poly_len / key_low as the coefficient-form path does.
static int compile_polynomials(Prover* p, const CircuitSrc* s) {
  Ctx* c = p->c;
  const uint64_t n = p->n, np = p->np, m = p->constraints;
  const uint32_t L = p->logn;
  if (n < 2) return (set_last_error("invalid argument", "plonk_compile: fewer than 2 gates", __FILE__, __LINE__), PLONK_ERR_ARG);
  for (int w = 0; w < 4; ++w)
    if (!s->wires[w]) return (set_last_error("invalid argument", "plonk_compile: wires", __FILE__, __LINE__), PLONK_ERR_ARG);
  for (int k = 0; k < QS_COUNT; ++k)
    if (s->selectors[k]) HIP_TRY(hipMemcpyAsync(p->polys + k * np, s->selectors[k], sizeof(Fr) * m, hipMemcpyHostToDevice, c->stream));
  // w^i, i < n (domain.elements(), permutation.rs:146): the polynomial X evaluated over the domain
  Fr* roots = p->cos;   // per-proof buffer, free until the first proof
  const Fr lin[2] = {Fr::zero(), Fr::one()};
  HIP_TRY(hipMemcpyAsync(p->scratch, lin, sizeof lin, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  PTRY(ntt_device(c, p->scratch, roots, p->tmp8, L, false, false, 2));
  // the selector interpolations run while the host walks the copy constraints
  for (int k = 0; k < QS_COUNT; ++k)
    if (s->selectors[k]) PTRY(ntt_device(c, p->polys + k * np, p->polys + k * np, p->tmp8, L, true, false, n));
  {
    std::vector<uint32_t> map(4 * n);
    if (!sigma_mappings(s->wires, m, n, s->witnesses, map.data()))
      return (set_last_error("invalid argument", "plonk_compile: witness index out of range", __FILE__, __LINE__), PLONK_ERR_ARG);
    DevFree map_dev;
    HIP_TRY(hipMalloc(&map_dev.p, sizeof(uint32_t) * 4 * n));
    HIP_TRY(hipMemcpyAsync(map_dev.p, map.data(), sizeof(uint32_t) * 4 * n, hipMemcpyHostToDevice, c->stream));
    SigmaArgs sa;
    sa.ks[0] = Fr::one(); sa.ks[1] = fr_small(7); sa.ks[2] = fr_small(13); sa.ks[3] = fr_small(17);
    PTRY(poly_sigma_evals(c, (const uint32_t*)map_dev.p, roots, p->polys + P_S1 * np, n, np, sa));
    HIP_TRY(hipStreamSynchronize(c->stream));   // map (host and device copy) is released here
  }
  for (int k = P_S1; k < P_S1 + 4; ++k) PTRY(ntt_device(c, p->polys + k * np, p->polys + k * np, p->tmp8, L, true, false, n));
  // Polynomial::from_coefficients_vec (polynomial.rs:79): degrees and the low coefficients the host keeps
  DevFree lens_dev;
  HIP_TRY(hipMalloc(&lens_dev.p, sizeof(unsigned long long) * P_COUNT));
  unsigned long long lens[P_COUNT];
  for (int k = 0; k < P_COUNT; ++k) PTRY(poly_trimmed_len(c, p->polys + k * np, n, (unsigned long long*)lens_dev.p + k));
  HIP_TRY(hipMemcpyAsync(lens, lens_dev.p, sizeof lens, hipMemcpyDeviceToHost, c->stream));
  for (int k = 0; k < P_COUNT; ++k)
    HIP_TRY(hipMemcpyAsync(p->key_low[k], p->polys + k * np, 7 * sizeof(Fr), hipMemcpyDeviceToHost, c->stream));   // np >= n + 8: zero past the degree
  // the witness index of every wire stays on the device for plonk_prover_prove_witnesses
  HIP_TRY(hipMalloc((void**)&p->wire_idx, sizeof(uint32_t) * 4 * m));
  for (int w = 0; w < 4; ++w)
    HIP_TRY(hipMemcpyAsync(p->wire_idx + w * m, s->wires[w], sizeof(uint32_t) * m, hipMemcpyHostToDevice, c->stream));
  p->witnesses = s->witnesses;
  HIP_TRY(hipMalloc((void**)&p->wit_vals, sizeof(Fr) * (s->witnesses ? s->witnesses : 1)));
  HIP_TRY(hipStreamSynchronize(c->stream));
  for (int k = 0; k < P_COUNT; ++k) p->poly_len[k] = lens[k];
  return PLONK_OK;
}

static int prover_build(Ctx* c, const plonk_prover_desc* d, const CircuitSrc* circ, Prover** out) {
  if (!d || d->constraints == 0) return (plonk::set_last_error("invalid argument", __func__, __FILE__, __LINE__), PLONK_ERR_ARG);
  if (d->label_len && !d->label) return (plonk::set_last_error("invalid argument: label", __func__, __FILE__, __LINE__), PLONK_ERR_ARG);
  for (int k = 0; k < P_COUNT; ++k)
    if (d->poly_len[k] && !d->polys[k]) return (plonk::set_last_error("invalid argument: polys", __func__, __FILE__, __LINE__), PLONK_ERR_ARG);
  Prover* p = new Prover();
  BuildGuard guard{p};
  p->c = c;
  p->constraints = d->constraints;
  uint64_t n = 1;
  uint32_t L = 0;
  while (n < d->constraints) { n <<= 1; ++L; }   // constraints.next_power_of_two() (compiler.rs:141)
  if (L + 3 >= 28) return (plonk::set_last_error("invalid argument", __func__, __FILE__, __LINE__), PLONK_ERR_ARG);
  {
    const bool force8 = c->cfg.quotient_domain == 8;   // plonk_gpu_config.quotient_domain / PLONK_QUOTIENT_DOMAIN
    p->qf = (n >= 8 && !force8) ? 4 : 8;
    p->lq = p->qf == 4 ? 2 : 3;
  }
  p->n = n; p->logn = L; p->n8 = p->qf * n; p->np = n + 8;
  p->label.assign((const char*)d->label, d->label_len);
  p->world = d->shard_world > 1 ? d->shard_world : 1;
  p->rank = p->world > 1 ? d->shard_rank : 0;
  if (p->rank < 0 || p->rank >= p->world) return (plonk::set_last_error("invalid argument", __func__, __FILE__, __LINE__), PLONK_ERR_ARG);
  p->srs_total = p->world > 1 ? d->srs_total : c->srs_n;
  p->srs_gen = c->srs_gen;
  {
    const uint64_t per = (p->srs_total + p->world - 1) / p->world;   // contiguous point ranges
    p->shard_lo = per * (uint64_t)p->rank;
    const uint64_t hi = p->shard_lo + per < p->srs_total ? p->shard_lo + per : p->srs_total;
    const uint64_t want = hi > p->shard_lo ? hi - p->shard_lo : 0;
    if (p->world > 1 && c->srs_n != want) return (plonk::set_last_error("invalid argument", __func__, __FILE__, __LINE__), PLONK_ERR_ARG);   // ctx must hold exactly this rank's slice
  }
  p->allgather = d->allgather;
  p->allgather_user = d->allgather_user;
  p->link.rank = p->rank; p->link.world = p->world; p->link.fn = d->allgather; p->link.user = d->allgather_user;
  if (p->world > 1 && c->nccl_comm && (c->comm_world != p->world || c->comm_rank != p->rank))
    return (plonk::set_last_error("invalid argument", "shard_rank / shard_world disagree with the context's communicator", __FILE__, __LINE__), PLONK_ERR_ARG);
  if (p->world > 1 && !c->nccl_comm && !d->allgather)
    return (plonk::set_last_error("invalid argument", "sharded prover needs plonk_comm_init on the context or an all-gather callback", __FILE__, __LINE__), PLONK_ERR_ARG);
  {
    // plonk_gpu_config.shard_quotient = -1 (PLONK_SHARD_QUOTIENT=0): shard only the MSMs (every rank runs the whole quotient)
    p->sharded = (p->world == 2 || p->world == 4 || p->world == 8) && n >= 64 && c->cfg.shard_quotient >= 0;
    if (p->sharded) {
      p->Q = p->world == 8 ? 8 : 4;
      p->cpr = p->Q / (uint32_t)p->world;
      for (uint32_t k = 0; k < p->cpr; ++k) p->cls[k] = (uint32_t)p->rank + (uint32_t)p->world * k;
      p->qf = p->Q; p->lq = p->Q == 8 ? 3 : 2;
      p->n8 = p->qf * n;
      p->per = (p->srs_total + p->world - 1) / p->world;
      p->stride = p->per + 8;
      p->lo = p->shard_lo < n + 7 ? p->shard_lo : n + 7;
      p->hi = p->shard_lo + p->per < n + 7 ? p->shard_lo + p->per : n + 7;
      if (p->srs_total < n + 7) return (plonk::set_last_error("invalid argument", "sharded prover: srs_total < size + 7", __FILE__, __LINE__), PLONK_ERR_DEGREE);
    }
    p->qn = p->sharded ? (uint64_t)p->cpr * n : p->n8;
  }
  if (p->world > 1) {
    p->gather_host = (uint8_t*)malloc((size_t)p->world * 16 * sizeof(G1));
    if (!p->gather_host) return (plonk::set_last_error("malloc", "gather_host", __FILE__, __LINE__), PLONK_ERR_HIP);
  }
  const uint64_t np = p->np, n8 = p->n8, qn = p->qn;
  const uint64_t wn = p->sharded ? n : n8;   // largest transform run inside prove(): scratch size
#define ALLOC(ptr, count) do { hipError_t _e = hipMalloc((void**)&(ptr), sizeof(*(ptr)) * (size_t)(count)); \
    if (_e != hipSuccess) { set_last_error("hipMalloc " #ptr, hipGetErrorString(_e), __FILE__, __LINE__); return PLONK_ERR_HIP; } } while (0)
  ALLOC(p->polys, P_COUNT * np);
  ALLOC(p->evals8, (P_COUNT + 2) * qn);
  ALLOC(p->sigma_n, 4 * n);
  ALLOC(p->wires, 4 * n);
  ALLOC(p->wpoly, 4 * np);
  ALLOC(p->zpoly, np);
  ALLOC(p->pipoly, np);
  ALLOC(p->cos, 6 * qn);
  ALLOC(p->tbuf, (p->sharded ? np : n8) + 16);   // sharded: t_fourth only, indexed by coefficient
  ALLOC(p->tmp8, wn);
  ALLOC(p->tmp8b, wn);
  ALLOC(p->tparts, 3 * np);
  ALLOC(p->agg, np);
  ALLOC(p->wit, np);
  ALLOC(p->wit2, np);
  ALLOC(p->scratch, 2 * np);
  ALLOC(p->totals, 256 * 64 + 1);   // scan block totals (poly.hip: SCAN_T * SCAN_MAXPER)
  p->ev_max_blocks = (uint32_t)((np + 1023) / 1024);   // poly_eval: 2^10 coefficients per workgroup for small polynomials, 2^12 above
  ALLOC(p->evpart, 16 * (uint64_t)p->ev_max_blocks);
  ALLOC(p->evout, 16);
  ALLOC(p->res, 16 * RES_STRIDE);
  ALLOC(p->len_dev, 1);
  ALLOC(p->flag_dev, 1);
  if (p->sharded) {
    ALLOC(p->fold, 2 * n);
    ALLOC(p->Fbuf, (uint64_t)p->cpr * n);
    ALLOC(p->send, (uint64_t)p->world * p->cpr * p->stride);
    ALLOC(p->recv, (uint64_t)p->world * p->cpr * p->stride);
    ALLOC(p->agg2, np);
    ALLOC(p->scratch2, np + 1);
  }
#undef ALLOC
  HIP_TRY(hipEventCreateWithFlags(&p->ev_ready, hipEventDisableTiming));
  HIP_TRY(hipEventCreateWithFlags(&p->ev_side, hipEventDisableTiming));
  HIP_TRY(hipEventCreateWithFlags(&p->ev_acc, hipEventDisableTiming));
  HIP_TRY(hipEventCreateWithFlags(&p->ev_pi, hipEventDisableTiming));
  HIP_TRY(hipEventCreateWithFlags(&p->ev_zlow, hipEventDisableTiming));
  for (int k = 0; k < 4; ++k) HIP_TRY(hipEventCreateWithFlags(&p->ev_wire[k], hipEventDisableTiming));
  HIP_TRY(hipHostMalloc((void**)&p->low_host, 42 * sizeof(Fr), hipHostMallocDefault));
  HIP_TRY(hipHostMalloc((void**)&p->res_host, 16 * RES_STRIDE, hipHostMallocDefault));
  HIP_TRY(hipHostMalloc((void**)&p->ev_host, 16 * sizeof(Fr), hipHostMallocDefault));
  HIP_TRY(hipHostMalloc((void**)&p->len_host, sizeof(unsigned long long), hipHostMallocDefault));
  HIP_TRY(hipHostMalloc((void**)&p->flag_host, sizeof(int), hipHostMallocDefault));

  // ---- ProverKey polynomials (coefficient form, trailing zeros ignored)
  PTRY(poly_fill_zero(c, p->polys, P_COUNT * np));
  if (circ) {
    PTRY(compile_polynomials(p, circ));
  } else {
    for (int k = 0; k < P_COUNT; ++k) {
      uint64_t len = d->poly_len[k];
      if (len > n) return (plonk::set_last_error("invalid argument", __func__, __FILE__, __LINE__), PLONK_ERR_ARG);
      if (len) HIP_TRY(hipMemcpyAsync(p->polys + k * np, d->polys[k], sizeof(Fr) * len, hipMemcpyHostToDevice, c->stream));
      // Polynomial::from_coefficients_vec trim (polynomial.rs:79): highest non-zero coefficient
      const Fr* hp = (const Fr*)d->polys[k];
      while (len && hp[len - 1].is_zero()) --len;
      p->poly_len[k] = len;
    }
    for (int k = 0; k < P_COUNT; ++k)
      for (int i = 0; i < 7; ++i)
        p->key_low[k][i] = (uint64_t)i < p->poly_len[k] ? ((const Fr*)d->polys[k])[i] : Fr::zero();
  }
  for (int s = 0; s < QS_COUNT; ++s) p->has[s] = p->poly_len[s] != 0;   // PolyId 0..10 == QS_*

  // ---- cached evaluations: 16 coset FFTs on the quotient domain (8n in compiler.rs:312-377) ...
  L1Args l1a;
  {
    // vanishing polynomial over the coset: qf distinct values g^n * w_qf^i - 1 (domain.rs:338-351) and
    // their inverses (prover.rs:78-91); the 8 slots repeat with period qf
    Fr point = fr_generator().pow_u64(n);
    const Fr step = omega_of(L + p->lq).pow_u64(n);   // primitive qf-th root of unity
    for (int i = 0; i < 8; ++i) {
      l1a.vh[i] = point - Fr::one();
      p->vinv[i] = l1a.vh[i].inv();
      p->sigma_j[i] = point;
      point = point * step;
    }
    l1a.n_inv = Fr::from_u64(n).inv();
    p->n_inv = l1a.n_inv;
    p->inv32 = Fr::from_u64(32).inv();
    p->edwards_d = (fr_small(10240) * fr_small(10241).inv()).neg();   // dusk_jubjub::EDWARDS_D
    p->gq_inv = fr_generator().pow_u64(p->n8).inv();                    // g^-(quotient domain size), de-aliasing
    p->omega = omega_of(L);
    p->omega_inv = p->omega.inv();
    if (p->sharded) {   // coef[i1][j] = g^(-n i1) w_Q^(-j i1) / Q
      const Fr gn_inv = fr_generator().pow_u64(n).inv(), w_inv = step.inv(), q_inv = Fr::from_u64(p->Q).inv();
      Fr gi = q_inv;
      for (uint32_t i1 = 0; i1 < 5; ++i1) {
        const Fr wi = w_inv.pow_u64(i1);
        Fr t = gi;
        for (uint32_t j = 0; j < 8; ++j) { p->coef[i1][j] = t; t = t * wi; }
        gi = gi * gn_inv;
      }
    }
  }
  const Fr lin[2] = {Fr::zero(), Fr::one()};
  HIP_TRY(hipMemcpyAsync(p->scratch, lin, sizeof lin, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (!p->sharded) {
    for (int k = 0; k < P_COUNT; ++k)
      PTRY(ntt_device(c, p->polys + k * np, p->evals8 + k * n8, p->tmp8, L + p->lq, false, true, p->poly_len[k]));
    PTRY(ntt_device(c, p->scratch, p->evals8 + P_COUNT * n8, p->tmp8, L + p->lq, false, true, 2));
    // L1 over the coset (quotient_poly.rs:266-284)
    PTRY(poly_l1(c, p->evals8 + P_COUNT * n8, p->evals8 + (P_COUNT + 1) * n8, n8, l1a));
  } else {
    // the same evaluations restricted to the classes this rank owns: size-n coset transforms with
    // shift g w_N^j, arrays laid out [polynomial][owned class][n]
    const Fr wN = omega_of(L + p->lq);
    for (uint32_t k = 0; k < p->cpr; ++k) {
      const uint32_t j = p->cls[k];
      const Fr shift = fr_generator() * wN.pow_u64(j);
      PTRY(ntt_coset_tables(c, L, shift, false, &p->cs_fwd[k]));
      PTRY(ntt_coset_tables(c, L, shift, true, &p->cs_inv[k]));
      for (int id = 0; id < P_COUNT; ++id)
        PTRY(ntt_device(c, p->polys + id * np, p->evals8 + id * qn + k * n, p->tmp8, L, false, true, p->poly_len[id], &p->cs_fwd[k]));
      PTRY(ntt_device(c, p->scratch, p->evals8 + P_COUNT * qn + k * n, p->tmp8, L, false, true, 2, &p->cs_fwd[k]));
      L1Args lj = l1a;
      for (int i = 0; i < 8; ++i) lj.vh[i] = l1a.vh[j];   // X^n - 1 is constant on a class
      PTRY(poly_l1(c, p->evals8 + P_COUNT * qn + k * n, p->evals8 + (P_COUNT + 1) * qn + k * n, n, lj));
    }
  }
  // ... 4 sigma FFTs on n (prover.rs:95-100)
  for (int k = 0; k < 4; ++k)
    PTRY(ntt_device(c, p->polys + (P_S1 + k) * np, p->sigma_n + k * n, p->tmp8, L, false, false, p->poly_len[P_S1 + k]));
  // pre-scaling for the reduced-radix quotient kernel (poly.hip): q_m * 2^10; q_l q_r q_o q_f q_arith, L1 * 2^5
  {
    const Fr s5 = Fr::from_u64(32), s10 = Fr::from_u64(1024);
    PTRY(poly_scale_array(c, p->evals8 + P_QM * qn, qn, s10));
    const int five[] = {P_QL, P_QR, P_QO, P_QF, P_QARITH, P_COUNT + 1};
    for (int id : five) PTRY(poly_scale_array(c, p->evals8 + (uint64_t)id * qn, qn, s5));
  }

  // ---- verifier-key commitments for transcript seeding
  if (d->vk_commitments) {
    memcpy(p->vk, d->vk_commitments, 15 * 48);
  } else {   // Compiler::preprocess commits (compiler.rs:213-232); zero polynomial -> identity
    for (int k0 = 0; k0 < P_COUNT; k0 += MSM_MAX_BATCH) {
      const int cnt = P_COUNT - k0 < MSM_MAX_BATCH ? P_COUNT - k0 : MSM_MAX_BATCH;
      const Fr* sc[MSM_MAX_BATCH];
      uint64_t ms[MSM_MAX_BATCH];
      for (int k = 0; k < cnt; ++k) { sc[k] = p->polys + (k0 + k) * np; ms[k] = p->poly_len[k0 + k]; }
      PTRY(msm_group(p, sc, ms, cnt, k0));
    }
    PTRY(fetch_commitments(p, 0, 15, p->vk));
  }
  {
    // plonk_gpu_config.wire_commit = 1 (PLONK_WIRE_COMMIT=coeff): commit to the coefficient form like the reference (A/B, fallback)
    if (p->world == 1 && !c->cfg.wire_commit_coeff && c->srs_n >= n + 2 && n >= 2) {
      if (d->lagrange_xy96 && d->lagrange_count != n + 2)
        return (plonk::set_last_error("invalid argument", "lagrange_count is not size + 2", __FILE__, __LINE__), PLONK_ERR_ARG);
      // Round 6, second session: a THIRD blinding point, [tau^(n+2)] G - [tau^2] G, behind the n + 2 points of the key — z
      // is blinded with b0 + b1 X + b2 X^2, so with it the commitment of z is an MSM of its EVALUATIONS as well (round 2 of
      // prove()), and z's inverse transform leaves the critical path.  The point comes from the context's commit key either
      // way (plonk_lagrange_key and lagrange_xy96 stay at n + 2 points); a key too short for it keeps the coefficient form.
      const bool third = c->srs_n >= n + 3 && c->cfg.z_commit_coeff <= 0;
      const uint64_t lag_cnt = n + 2 + (third ? 1 : 0);
      G1Affine* lag_pts = nullptr;
      HIP_TRY(hipMalloc((void**)&lag_pts, sizeof(G1Affine) * lag_cnt));
      int rc = PLONK_OK;
      if (d->lagrange_xy96) {   // a key the caller kept from an earlier plonk_lagrange_key: skips the group FFT
        if (hipMemcpyAsync(lag_pts, d->lagrange_xy96, sizeof(G1Affine) * (n + 2), hipMemcpyHostToDevice, c->stream) != hipSuccess) rc = PLONK_ERR_HIP;
      } else {
        rc = lagrange_points_device(c, L, lag_pts);
      }
      if (rc == PLONK_OK && third) rc = lagrange_blind_points_device(c, n, 2, 1, lag_pts + (n + 2));
      if (rc == PLONK_OK && d->lagrange_xy96) rc = srs_validate_device(c, lag_pts, n + 2, p->flag_dev);   // on the curve, in the subgroup
      if (rc == PLONK_OK) rc = srs_table_build(c, lag_pts, lag_cnt, &p->lag_table, &p->lag_rows);
      p->lag_n = lag_cnt;
      p->lag_on = true;
      p->lag_z = third;
      if (rc == PLONK_OK && hipStreamSynchronize(c->stream) != hipSuccess) rc = PLONK_ERR_HIP;
      (void)hipFree(lag_pts);
      if (rc) return rc;
      if (d->lagrange_xy96) {
        int bad = 0;
        HIP_TRY(hipMemcpy(&bad, p->flag_dev, sizeof(int), hipMemcpyDeviceToHost));
        if (bad) return (plonk::set_last_error("lagrange_xy96", "point off the curve or outside the prime-order subgroup", __FILE__, __LINE__), PLONK_ERR_POINT);
        PTRY(check_lagrange_key(p, L));
      }
      HIP_TRY(hipMalloc((void**)&p->wscal, sizeof(Fr) * 11));
    } else if (p->world > 1 && !c->cfg.wire_commit_coeff && d->lagrange_xy96) {
      // multi-GPU: the rank's slice [shard_lo, shard_lo + count) of the (n + 2)-point Lagrange key, computed where the whole
      // commit key was available (plonk_lagrange_key)
      const uint64_t lhi = p->shard_lo + p->per < n + 2 ? p->shard_lo + p->per : n + 2;
      uint64_t want = lhi > p->shard_lo ? lhi - p->shard_lo : 0;
      // Round 5 ("by commitment"): a rank of 2 or 4 handed the WHOLE key (size + 2 points) commits to whole wire
      // columns — rank r of 2 to columns 2r, 2r + 1, rank r of 4 to column r — instead of a quarter / half of every column:
      // the same additions, but one or two commitments' worth of sort / bucket / row-column / bit-sum tails instead of four.
      // Every rank holds all wire values anyway; the price is the whole table on every rank (32 GiB of bit-position rows
      // at 2^20 gates, taken from the context's table budget like any other key).
      p->lag_whole = p->sharded && (p->world == 2 || p->world == 4) && d->lagrange_count == n + 2 && want != n + 2;
      if (p->lag_whole) want = n + 2;
      if (!p->sharded || d->lagrange_count != want)
        return (plonk::set_last_error("invalid argument", "lagrange_count is neither this rank's slice of the size + 2 points nor all of them", __FILE__, __LINE__), PLONK_ERR_ARG);
      if (want) {
        G1Affine* lag_pts = nullptr;
        HIP_TRY(hipMalloc((void**)&lag_pts, sizeof(G1Affine) * want));
        int rc = PLONK_OK;
        if (hipMemcpyAsync(lag_pts, d->lagrange_xy96, sizeof(G1Affine) * want, hipMemcpyHostToDevice, c->stream) != hipSuccess) rc = PLONK_ERR_HIP;
        // a rank holds a SLICE of both keys, so only the points themselves can be checked here (curve, subgroup); the
        // consistency of the whole key with the commit key is the single-GPU check above, run where plonk_lagrange_key ran
        if (rc == PLONK_OK) rc = srs_validate_device(c, lag_pts, want, p->flag_dev);
        if (rc == PLONK_OK) rc = srs_table_build(c, lag_pts, want, &p->lag_table, &p->lag_rows);
        if (rc == PLONK_OK && hipStreamSynchronize(c->stream) != hipSuccess) rc = PLONK_ERR_HIP;
        (void)hipFree(lag_pts);
        if (rc) return rc;
        int bad = 0;
        HIP_TRY(hipMemcpy(&bad, p->flag_dev, sizeof(int), hipMemcpyDeviceToHost));
        if (bad) return (plonk::set_last_error("lagrange_xy96", "point off the curve or outside the prime-order subgroup", __FILE__, __LINE__), PLONK_ERR_POINT);
      }
      p->lag_n = want;
      p->lag_on = true;
      HIP_TRY(hipMalloc((void**)&p->wscal, sizeof(Fr) * 11));
    }
    if (p->world > 1) {
      // every rank derives its mode from ITS OWN descriptor — no key (coefficient-form wire commitments), its slice,
      // or the whole key (wire group by column).  Ranks that disagree would add whole-column commitments to point-range partial
      // sums (or Lagrange-form to coefficient-form ones): silently wrong a / b / c / d commitments that the prover's own identity
      // check cannot see.  One byte per rank settles it before the first proof.
      uint8_t mine = p->lag_whole ? 2 : (p->lag_on ? 1 : 0);
      PTRY(comm_allgather_host(c, p->link, &mine, p->gather_host, 1));
      for (uint32_t r = 0; r < p->world; ++r)
        if (p->gather_host[r] != mine)
          return (plonk::set_last_error("plonk_prover_desc.lagrange_xy96 / lagrange_count",
                                        "the ranks disagree on the wire-commitment mode (no Lagrange-basis key / the rank's slice / the whole key): "
                                        "every rank of a sharded prover must pass the same kind", __FILE__, __LINE__), PLONK_ERR_ARG);
      if (mine && !c->comm_loopback) PTRY(check_lagrange_key(p, L));   // (loop-back measurement: the gathered sums are wrong by construction)
    }
  }
  HIP_TRY(hipStreamSynchronize(c->stream));
  guard.p = nullptr;
  *out = p;
  return PLONK_OK;
}

// ---- pieces shared by the single-GPU and the sharded prove() ---------------------------------------------------
// transcript_for_version(V3) = Transcript::base_v3 (transcript.rs:131-145) + VerifierKey::seed_transcript
// (widget.rs:218-258) + the public inputs (prover.rs:440-442)
static void seed_transcript(Transcript& tr, const Prover* p, const Fr* pi_val, uint64_t pi_count) {
  tr.circuit_domain_sep(p->constraints);
  // PlonkVersion::V2 (prove_with_version, prover.rs:365-413; feature `legacy-proving`): Transcript::base + seed_transcript_legacy
  // (transcript.rs:110-129, widget.rs:224-228,260-265) bind the LABEL s_sigma_4 to the commitment of s_sigma_1; nothing else differs
  for (int k = 0; k < 15; ++k) tr.append_commitment(VK_LABEL[k], p->vk[(p->transcript_version == 2 && k == 14) ? P_S1 : VK_ORDER[k]]);
  tr.circuit_domain_sep(p->constraints);   // vk.n == constraints (compiler.rs:279)
  for (uint64_t i = 0; i < pi_count; ++i) tr.append_scalar("pi", pi_val[i]);
}
// prover.rs:623-632,650-658
static void append_evaluations(Transcript& tr, const Evals& ev) {
  tr.append_scalar("a_eval", ev.a);
  tr.append_scalar("b_eval", ev.b);
  tr.append_scalar("c_eval", ev.c);
  tr.append_scalar("d_eval", ev.d);
  tr.append_scalar("s_sigma_1_eval", ev.s1);
  tr.append_scalar("s_sigma_2_eval", ev.s2);
  tr.append_scalar("s_sigma_3_eval", ev.s3);
  tr.append_scalar("z_eval", ev.z);
  tr.append_scalar("a_w_eval", ev.a_w);
  tr.append_scalar("b_w_eval", ev.b_w);
  tr.append_scalar("d_w_eval", ev.d_w);
  tr.append_scalar("q_arith_eval", ev.q_arith);
  tr.append_scalar("q_c_eval", ev.q_c);
  tr.append_scalar("q_l_eval", ev.q_l);
  tr.append_scalar("q_r_eval", ev.q_r);
}
// Proof::to_bytes (proof.rs:137-162, linearization_poly.rs:98-124)
static void write_proof(uint8_t proof[1008], const uint8_t (*comm)[48], const Evals& ev) {
  memcpy(proof, comm, 11 * 48);
  const Fr* order[15] = {&ev.a, &ev.b, &ev.c, &ev.d, &ev.a_w, &ev.b_w, &ev.d_w, &ev.q_arith, &ev.q_c, &ev.q_l,
                         &ev.q_r, &ev.s1, &ev.s2, &ev.s3, &ev.z};
  for (int k = 0; k < 15; ++k) fr_to_bytes(*order[k], proof + 11 * 48 + 32 * k);
}
// public-input evaluation at z (compute_barycentric_eval, proof.rs:1041-1088): one shared inversion for all
// denominators (Montgomery's trick); a zero denominator means z is a root of unity (probability n/q) and
// contributes nothing, like the reference's skip.  zh = z^n - 1.
static Fr public_input_eval(const Prover* p, const uint64_t* pi_idx, const Fr* pi_val, uint64_t pi_count, const Fr& z_ch, const Fr& zh) {
  if (!pi_count) return Fr::zero();
  const Fr one = Fr::one();
  std::vector<Fr> den(pi_count), pre(pi_count);
  Fr run = one;
  for (uint64_t i = 0; i < pi_count; ++i) {
    den[i] = p->omega_inv.pow_u64(pi_idx[i]) * z_ch - one;
    pre[i] = run;
    if (!pi_val[i].is_zero() && !den[i].is_zero()) run = run * den[i];
  }
  Fr inv = fr_inv_gcd(run), acc = Fr::zero();
  for (uint64_t i = pi_count; i-- > 0;) {
    if (pi_val[i].is_zero() || den[i].is_zero()) continue;
    acc = acc + inv * pre[i] * pi_val[i];
    inv = inv * den[i];
  }
  return acc * (zh * p->n_inv);
}
// num(z) = t(z) Z_H(z)  <=>  r(z) = alpha^2 L1(z) + alpha (a + beta s1 + gamma)(b + beta s2 + gamma)(c + beta s3 + gamma)
// (d + gamma) z(omega z)  (the constant the verifier calls pi(z) - r_0, proof.rs:290-310); the W_z numerator adds
// sum_i v^i eval_i to r.  A mismatch means the witness does not satisfy the circuit (Error::CircuitUnsatisfied).
static bool quotient_identity_holds(const Fr& numerator_at_z, const Evals& ev, const Fr& alpha, const Fr& beta, const Fr& gamma,
                                    const Fr& l1_z, const Fr* vp /* v^0 .. v^11 */) {
  Fr expect = alpha.sqr() * l1_z + (ev.a + beta * ev.s1 + gamma) * (ev.b + beta * ev.s2 + gamma) *
                                       (ev.c + beta * ev.s3 + gamma) * (ev.d + gamma) * ev.z * alpha;
  const Fr* evs[11] = {&ev.a, &ev.b, &ev.c, &ev.d, &ev.s1, &ev.s2, &ev.s3, &ev.q_arith, &ev.q_c, &ev.q_l, &ev.q_r};
  for (int k = 0; k < 11; ++k) expect = expect + vp[k + 1] * *evs[k];
  return numerator_at_z == expect;
}
// the sparse public inputs -> dense evaluation vector -> PI(X) in coefficient form, on the current stream
static int public_input_polynomial(Prover* p, const uint64_t* pi_idx, const Fr* pi_val, uint64_t pi_count, Fr* ntt_tmp) {
  Ctx* c = p->c;
  const uint64_t n = p->n;
  if (pi_count > p->pi_cap) {
    if (p->pi_idx_dev) { HIP_TRY(hipFree(p->pi_idx_dev)); HIP_TRY(hipFree(p->pi_val_dev)); p->pi_idx_dev = nullptr; p->pi_val_dev = nullptr; p->pi_cap = 0; }
    HIP_TRY(hipMalloc((void**)&p->pi_idx_dev, sizeof(uint64_t) * pi_count));
    HIP_TRY(hipMalloc((void**)&p->pi_val_dev, sizeof(Fr) * pi_count));
    p->pi_cap = pi_count;
  }
  for (uint64_t i = 0; i < pi_count; ++i) if (pi_idx[i] >= n) return (plonk::set_last_error("invalid argument", "public input row beyond the domain", __FILE__, __LINE__), PLONK_ERR_ARG);
  HIP_TRY(hipMemcpyAsync(p->pi_idx_dev, pi_idx, sizeof(uint64_t) * pi_count, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(p->pi_val_dev, pi_val, sizeof(Fr) * pi_count, hipMemcpyHostToDevice, c->stream));
  PTRY(poly_scatter_pi(c, p->pipoly, p->pi_idx_dev, p->pi_val_dev, pi_count));
  return ntt_device(c, p->pipoly, p->pipoly, ntt_tmp, p->logn, true, false, n);
}

static int prover_prove_sharded(Prover* p, const Fr* wires_dev, const uint64_t* pi_idx, const Fr* pi_val, uint64_t pi_count,
                                const Fr* bl, uint8_t proof[1008]);

static int prover_prove(Prover* p, const Fr* wires_dev, const uint64_t* pi_idx, const Fr* pi_val, uint64_t pi_count,
                        const Fr* bl, uint8_t proof[1008]) {
  if (p->c->comm_poisoned) return (set_last_error("context unusable", "a collective timed out and its stream never drained (comm_sync): destroy the context", __FILE__, __LINE__), PLONK_ERR_STATE);
  if (p->sharded) return prover_prove_sharded(p, wires_dev, pi_idx, pi_val, pi_count, bl, proof);
  Ctx* c = p->c;
  // the commit key of the context was replaced after this prover was built (plonk_srs_load /
  // plonk_prover_from_bytes): its degree bounds and shard ranges no longer describe the tables
  if (p->srs_gen != c->srs_gen) return (set_last_error("prover is bound to an SRS that was replaced on its context", __func__, __FILE__, __LINE__), PLONK_ERR_STATE);
  const uint64_t n = p->n, n8 = p->n8, np = p->np;
  const uint32_t L = p->logn;
  NttTables* tbn;
  PTRY(ntt_tables(c, L, false, &tbn));
  const Fr omega = omega_of(L);

  // transcript_for_version(V3) = Transcript::base_v3 (transcript.rs:131-145, widget.rs:218-258)
  Transcript tr((const uint8_t*)p->label.data(), p->label.size());
  seed_transcript(tr, p, pi_val, pi_count);

  uint8_t comm[11][48];
  HostGap gap{c};
  GapScope gap_scope(&gap);
  // ---- round 1 (prover.rs:444-479)
  const bool lag = p->lag_on;
  // iNTT + blinding of the four columns and their lowest coefficients (quotient_low), on the CURRENT stream
  auto wire_polynomials = [&](Fr* ntt_tmp, int k0 = 0, int k1 = 4) -> int {
    prof_begin(c, 4);   // slot 4: the polynomial work of rounds 1-2 (replicated on every rank of a multi-GPU run)
    for (int k = k0; k < k1; ++k) {
      Fr* wp = p->wpoly + k * np;
      if (p->wires_pending) HIP_TRY(hipStreamWaitEvent(c->stream, p->ev_wire[k], 0));   // column k has arrived
      PTRY(ntt_device(c, wires_dev + k * n, wp, ntt_tmp, L, true, false, n));
      BlindArgs ba;
      ba.count = 2;
      ba.b[0] = bl[2 * k];
      ba.b[1] = bl[2 * k + 1];
      PTRY(poly_fill_zero(c, wp + n, np - n));
      PTRY(poly_blind(c, wp, n, ba));
    }
    prof_end(c, 4);
    for (int k = k0; k < k1; ++k)
      HIP_TRY(hipMemcpyAsync(p->low_host + 7 * k, p->wpoly + k * np, 7 * sizeof(Fr), hipMemcpyDeviceToHost, c->stream));
    return PLONK_OK;
  };
  // Up to 2^19 gates the wire polynomials (needed from round 3 on) ride on the side stream under the commitment pipeline,
  // started after the group's accumulation (side_defer below): they fill its latency-bound reduction tail.  At 2^20 the
  // accumulation owns the VALU and its register file (two waves of 256 VGPRs per SIMD: no transform wave fits beside
  // them), the tails of the four groups are too short for the 6 ms of side transforms, and the inverse transforms keep
  // their place in front of the commitment (same-box A/B, profiles/r06b/polys_side_*.jsonl: 2^19 18.1-18.3 -> 17.4-17.7 ms
  // with both moved, 2^20 32.17 -> 32.07-32.44).  PLONK_WIRE_POLYS_SIDE=0/1 forces it.
  // plonk_gpu_config has no field for it; PLONK_WIRE_BY_COLUMN: 0 = never, 1 / 2 = at EVERY size (a, b, c + d / one launch per
  // column: what the variant tests use to run the phased launches on small circuits), unset = from 2^19 gates on
  const int bc_cfg = c->cfg.wire_by_column;
  const bool by_column = lag && p->wires_pending && p->world == 1 && bc_cfg >= 0 && (L > 18 || bc_cfg > 0);
  const bool polys_on_side = lag && !by_column && (c->cfg.wire_polys_side >= 0 ? c->cfg.wire_polys_side == 1 : L <= 19);
  // Host wire columns (plonk_prover_prove, round 6): column k lands over PCIe 32 n bytes after column k - 1 (0.65 ms apart at
  // 2^20 gates), and until round 5 the commitment group waited for all four before its grouped bucket sort — 2.1 ms of every
  // such proof with nothing but the four inverse transforms to hide in.  Now each column's transform is followed at once by
  // ITS bucket sort, accumulation and bucket sums (msm_batch_device phase 1 on column k: the throughput-bound 90 % of a
  // commitment), under the next columns' copies; the latency-bound reduction tail still runs once for the group (phase 2).
  // Same additions, same bucket sums, same bytes.  Resident columns (plonk_prover_prove_dev, what `value` times) keep the
  // grouped launch: four launches have four ramp-downs.
  if (!polys_on_side && !by_column) PTRY(wire_polynomials(p->tmp8));
  if (lag && by_column) {
    HIP_TRY(hipMemcpyAsync(p->wscal, bl, 8 * sizeof(Fr), hipMemcpyHostToDevice, c->stream));
  } else if (lag) {
    // Lagrange-basis key: a(X) = sum_i w_i L_i(X) + b0 (X^n - 1) + b1 (X^(n+1) - X)  (blind_poly, prover.rs:139-152), so the
    // commitment is an MSM of the wire VALUES and the two blinders over [L_i(tau)] G, [tau^n] G - G, [tau^(n+1)] G - [tau] G
    // (the values are read in place; only the eight blinders travel: bl[0..8) = a0 a1 b0 b1 c0 c1 d0 d1)
    for (int k = 0; k < 4; ++k)
      if (p->wires_pending) HIP_TRY(hipStreamWaitEvent(c->stream, p->ev_wire[k], 0));
    HIP_TRY(hipMemcpyAsync(p->wscal, bl, 8 * sizeof(Fr), hipMemcpyHostToDevice, c->stream));
  }
  SideJoin side_join{c};
  uint64_t pi_len = 0;
  // Side transforms either start with the group (they then compete with its bandwidth-bound sort) or wait for the end of
  // its accumulation and fill the latency-bound tail.  Same-box A/B (r02e): waiting wins up to 2^18 gates (2^16: 5.19 vs
  // 5.33 ms) and on the widget workload (38.1 vs 38.5 ms) but loses on the dense 2^20 headline (37.2-37.4 vs 36.6-36.8 ms:
  // z's transform no longer fits between its commitment and the quotient), so it follows the size; round 6: 2^19 gates
  // wait too, together with the wire inverse transforms (above).  PLONK_SIDE_DEFER=0/1 forces it.
  const int side_defer_env = c->cfg.side_defer;
  const bool defer_size = L <= 18 || (L == 19 && !by_column);   // (host wire columns at 2^19 gates keep the phased launches' order)
  const bool side_defer = side_defer_env >= 0 ? side_defer_env >= 1 : defer_size;       // round 1: a, b, c, d (+ PI)
  const bool side_defer_z = side_defer_env >= 0 ? side_defer_env == 1 : defer_size;      // round 2: z ("2" = round 1 only)
  const bool z_from_evals = lag && p->lag_z && (L <= 17 || c->cfg.z_commit_coeff < 0);                                   // z committed from its evaluations (round 2)
  auto side_round1 = [&]() -> int {
    // quotient_poly.rs:139-157,177: coset FFTs of a, b, c, d and of the public-input polynomial
    // (prover.rs:520-521) need no challenge -> side stream, overlapped with the commitments; with the
    // Lagrange key the wire polynomials themselves are only needed from round 3 on and move there too
    if (polys_on_side) PTRY(wire_polynomials(p->tmp8b));
    for (int k = 0; k < 4; ++k)
      PTRY(ntt_device(c, p->wpoly + k * np, p->cos + (1 + k) * n8, p->tmp8b, L + p->lq, false, true, n + 2));
    PTRY(poly_fill_zero(c, p->pipoly, np));
    if (pi_count) {
      PTRY(public_input_polynomial(p, pi_idx, pi_val, pi_count, p->tmp8b));
      pi_len = n;
    }
    HIP_TRY(hipMemcpyAsync(p->low_host + 35, p->pipoly, 7 * sizeof(Fr), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipEventRecord(p->ev_pi, c->stream));
    if (pi_len) PTRY(ntt_device(c, p->pipoly, p->cos + 5 * n8, p->tmp8b, L + p->lq, false, true, pi_len));   // no public inputs: PI(X) = 0, nothing to transform or read
    return PLONK_OK;
  };
  if (!side_defer && !by_column) {
    SideScope side(c, p->ev_ready);
    PTRY(side_round1());
  }
  if (by_column) {
    const uint64_t ms[4] = {n + 2, n + 2, n + 2, n + 2};
    const Fr* sc[4] = {wires_dev, wires_dev + n, wires_dev + 2 * n, wires_dev + 3 * n};
    const Fr* tl[4] = {p->wscal, p->wscal + 2, p->wscal + 4, p->wscal + 6};
    const uint64_t sp[4] = {n, n, n, n};
    // Columns a and b each get a launch of their own as they land; c and d — both have arrived by the time b's accumulation
    // ends (0.65 ms per column over PCIe against ~1.5 ms of commitment work per column at 2^20 gates) — share one: a sort
    // launch set and an accumulation ramp-down less.  Measured and NOT adopted (profiles/r06/host_wires_ab.jsonl): column k's
    // work on a stream of its own (k & 1), gated only by its copy — no gain at 2^20 (gap 1.09-1.27 against 1.12 ms) and a
    // loss at 2^19 (0.38-0.54 against 0.24): the accumulations are VALU-bound, so overlapping them only interleaves them.
    // PLONK_WIRE_BY_COLUMN=2: one launch per column (A/B)
    const int parts[3][2] = {{0, 1}, {1, 1}, {2, 2}};
    const int nparts = c->cfg.wire_by_column == 2 ? 4 : 3;
    p->last_wire_launches = (uint32_t)nparts;
    for (int q = 0; q < nparts; ++q) {
      const int k0 = nparts == 4 ? q : parts[q][0], kc = nparts == 4 ? 1 : parts[q][1];
      PTRY(wire_polynomials(p->tmp8, k0, k0 + kc));   // waits for the columns' copies, then their inverse transforms + blinding
      if (k0 + kc == 4 && !side_defer) {              // all four polynomials exist: their coset transforms go to the side stream
        SideScope side(c, p->ev_ready);
        PTRY(side_round1());
      }
      AccMark mark(c, (side_defer && k0 + kc == 4) ? p->ev_acc : nullptr);
      PTRY(msm_group(p, sc, ms, 4, 0, p->lag_table, p->lag_n, tl, sp, 1, k0, kc));
    }
    PTRY(msm_group(p, sc, ms, 4, 0, p->lag_table, p->lag_n, tl, sp, 2));
  } else {
    p->last_wire_launches = 1;
    AccMark mark(c, side_defer ? p->ev_acc : nullptr);
    const uint64_t ms[4] = {n + 2, n + 2, n + 2, n + 2};
    if (lag) {
      const Fr* sc[4] = {wires_dev, wires_dev + n, wires_dev + 2 * n, wires_dev + 3 * n};
      const Fr* tl[4] = {p->wscal, p->wscal + 2, p->wscal + 4, p->wscal + 6};
      const uint64_t sp[4] = {n, n, n, n};
      PTRY(msm_group(p, sc, ms, 4, 0, p->lag_table, p->lag_n, tl, sp));
    } else {
      const Fr* sc[4] = {p->wpoly, p->wpoly + np, p->wpoly + 2 * np, p->wpoly + 3 * np};
      PTRY(msm_group(p, sc, ms, 4, 0));   // commit_polynomials (prover.rs:187-210) as one group launch
    }
  }
  if (side_defer) {
    SideScopeAfter side(c, p->ev_acc);
    PTRY(side_round1());
  }
  PTRY(fetch_commitments(p, 0, 4, comm));
  tr.append_commitment("a_comm", comm[0]);
  tr.append_commitment("b_comm", comm[1]);
  tr.append_commitment("c_comm", comm[2]);
  tr.append_commitment("d_comm", comm[3]);

  // ---- round 2 (prover.rs:481-505)
  const Fr beta = tr.challenge_scalar("beta");
  tr.append_scalar("beta", beta);
  const Fr gamma = tr.challenge_scalar("gamma");
  {
    prof_begin(c, 4);
    PermArgs pa;
    pa.n = n;
    for (int k = 0; k < 4; ++k) { pa.wires[k] = wires_dev + k * n; pa.sigma[k] = p->sigma_n + k * n; }
    pa.beta = beta; pa.gamma = gamma;
    pa.ks[0] = Fr::one(); pa.ks[1] = fr_small(7); pa.ks[2] = fr_small(13); pa.ks[3] = fr_small(17);
    pa.tw_lo29 = tbn->tw_lo29; pa.tw_hi29 = tbn->tw_hi29; pa.lobits = L < 13 ? L : 13; pa.use_hi = L > 13;
    pa.num = p->scratch; pa.den = p->scratch + np;
    gap.launching();
    HIP_TRY(hipMemsetAsync(p->flag_dev, 0, sizeof(int), c->stream));
    PTRY(poly_perm_terms(c, pa));
    // numerators / denominators stay in twiddle form (x * 2^261) from perm_terms to the end of the scan
    PTRY(poly_batch_inverse(c, pa.den, n, true));
    PTRY(poly_mul_arrays(c, pa.num, pa.den, n, p->flag_dev));
    PTRY(scan_prefix_product(c, pa.num, n, p->totals));
    if (!z_from_evals) {
      PTRY(ntt_device(c, pa.num, p->zpoly, p->tmp8, L, true, false, n));
      BlindArgs ba;
      ba.count = 3;
      ba.b[0] = bl[8]; ba.b[1] = bl[9]; ba.b[2] = bl[10];
      PTRY(poly_fill_zero(c, p->zpoly + n, np - n));
      PTRY(poly_blind(c, p->zpoly, n, ba));
    }
    prof_end(c, 4);
  }
  // z(X) in coefficient form (needed from round 3 on), its lowest coefficients and its coset transform, on the CURRENT stream
  auto z_polynomial = [&](Fr* ntt_tmp) -> int {
    PTRY(ntt_device(c, p->scratch, p->zpoly, ntt_tmp, L, true, false, n));
    BlindArgs ba;
    ba.count = 3;
    ba.b[0] = bl[8]; ba.b[1] = bl[9]; ba.b[2] = bl[10];
    PTRY(poly_fill_zero(c, p->zpoly + n, np - n));
    PTRY(poly_blind(c, p->zpoly, n, ba));
    return PLONK_OK;
  };
  if (z_from_evals) {
    // Round 6, second session: z(X) = sum_i z_i L_i(X) + (b0 + b1 X + b2 X^2)(X^n - 1), so — like the wires — its commitment is
    // an MSM of the n EVALUATIONS the scan just wrote and the three blinders over the Lagrange-basis key and its blinding
    // points: the inverse transform, the blinding and the copy of the low coefficients leave the critical path (the side
    // stream runs them before z's coset transform; `scratch` keeps the evaluations until round 5 reuses it).  Up to 2^17 gates
    // only: same-box A/B 2^12 2.04 -> 2.01 ms, 2^16 4.09 -> 4.03, 2^17 5.88 -> 5.86 (even in a later
    // five-repetition pair); at 2^18 the coefficient form is 1 % ahead (9.88 against 10.00 ms, mid_sizes_ab.jsonl), at 2^19 too, and
    // at 2^20 the transform that left the main stream competes with the commitment's sort on the side stream instead (32.2 against
    // 32.3), so larger circuits keep the coefficient form (profiles/r06b/zcommit_ab.jsonl).
    HIP_TRY(hipMemcpyAsync(p->wscal + 8, bl + 8, 3 * sizeof(Fr), hipMemcpyHostToDevice, c->stream));
    if (!side_defer_z) {
      SideScope side(c, p->ev_ready);
      PTRY(z_polynomial(p->tmp8b));
      HIP_TRY(hipMemcpyAsync(p->low_host + 28, p->zpoly, 7 * sizeof(Fr), hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(hipEventRecord(p->ev_zlow, c->stream));
      PTRY(ntt_device(c, p->zpoly, p->cos, p->tmp8b, L + p->lq, false, true, n + 3));
      HIP_TRY(hipEventRecord(p->ev_side, c->side_stream));
    }
    {
      AccMark mark(c, side_defer_z ? p->ev_acc : nullptr);
      const Fr* sc[1] = {p->scratch};
      const Fr* tl[1] = {p->wscal + 8};
      const uint64_t ms[1] = {n + 3}, sp[1] = {n};
      PTRY(msm_group(p, sc, ms, 1, 4, p->lag_table, p->lag_n, tl, sp));
    }
    if (side_defer_z) {
      SideScopeAfter side(c, p->ev_acc);
      PTRY(z_polynomial(p->tmp8b));
      HIP_TRY(hipMemcpyAsync(p->low_host + 28, p->zpoly, 7 * sizeof(Fr), hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(hipEventRecord(p->ev_zlow, c->stream));
      PTRY(ntt_device(c, p->zpoly, p->cos, p->tmp8b, L + p->lq, false, true, n + 3));
      HIP_TRY(hipEventRecord(p->ev_side, c->side_stream));
    }
  } else {
  HIP_TRY(hipMemcpyAsync(p->low_host + 28, p->zpoly, 7 * sizeof(Fr), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipEventRecord(p->ev_zlow, c->stream));
  if (!side_defer_z) {
    SideScope side(c, p->ev_ready);   // z's coset FFT only needs z(X): overlap with its commitment
    PTRY(ntt_device(c, p->zpoly, p->cos, p->tmp8b, L + p->lq, false, true, n + 3));
    HIP_TRY(hipEventRecord(p->ev_side, c->side_stream));
  }
  {
    AccMark mark(c, side_defer_z ? p->ev_acc : nullptr);
    PTRY(msm_to(p, p->zpoly, n + 3, 4));
  }
  if (side_defer_z) {
    SideScopeAfter side(c, p->ev_acc);
    PTRY(ntt_device(c, p->zpoly, p->cos, p->tmp8b, L + p->lq, false, true, n + 3));
    HIP_TRY(hipEventRecord(p->ev_side, c->side_stream));
  }
  }
  PTRY(fetch_commitments(p, 4, 1, comm + 4));
  tr.append_commitment("z_comm", comm[4]);

  // ---- round 3 (prover.rs:507-589)
  const Fr alpha = tr.challenge_scalar("alpha");
  const Fr range_ch = tr.challenge_scalar("range separation challenge");
  const Fr logic_ch = tr.challenge_scalar("logic separation challenge");
  const Fr fixed_ch = tr.challenge_scalar("fixed base separation challenge");
  const Fr var_ch = tr.challenge_scalar("variable base separation challenge");
  const Fr edwards_d = p->edwards_d;
  // quotient (quotient_poly.rs:20-137): the 6 coset FFTs on 8n were issued on the side stream in
  // rounds 1-2; point-wise pass, then coset iFFT
  HIP_TRY(hipStreamWaitEvent(c->main_stream, p->ev_side, 0));
  {
    QuotientArgs q;
    q.n8 = n8;
    q.rot = p->qf;
    q.z = p->cos; q.a = p->cos + n8; q.b = p->cos + 2 * n8; q.c = p->cos + 3 * n8; q.d = p->cos + 4 * n8; q.pi = pi_len ? p->cos + 5 * n8 : nullptr;
    const Fr* e = p->evals8;
    q.q_m = e + P_QM * n8; q.q_l = e + P_QL * n8; q.q_r = e + P_QR * n8; q.q_o = e + P_QO * n8; q.q_f = e + P_QF * n8;
    q.q_c = e + P_QC * n8; q.q_arith = e + P_QARITH * n8; q.q_range = e + P_QRANGE * n8; q.q_logic = e + P_QLOGIC * n8;
    q.q_fixed = e + P_QFIXED * n8; q.q_var = e + P_QVAR * n8;
    q.s1 = e + P_S1 * n8; q.s2 = e + P_S2 * n8; q.s3 = e + P_S3 * n8; q.s4 = e + P_S4 * n8;
    q.linear = e + P_COUNT * n8; q.l1 = e + (P_COUNT + 1) * n8;
    for (int s = 0; s < QS_COUNT; ++s) q.has[s] = p->has[s];
    q.range_ch = range_ch; q.logic_ch = logic_ch; q.fixed_ch = fixed_ch; q.var_ch = var_ch;
    q.edwards_d = edwards_d;
    q.inv32 = p->inv32;
    quotient_data(gamma, q.k.gamma);
    quotient_data(Fr::one(), q.k.one);
    const Fr ks[4] = {Fr::one(), fr_small(7), fr_small(13), fr_small(17)};
    for (int k = 0; k < 4; ++k) quotient_const(beta * ks[k], 0, q.k.beta_k[k]);
    quotient_const(alpha, 20, q.k.alpha_pos);
    quotient_const(alpha.neg(), 20, q.k.alpha_neg);
    quotient_const(alpha.sqr(), 0, q.k.alpha_sq);
    for (int i = 0; i < 8; ++i) quotient_const(p->vinv[i], 0, q.k.vinv[i]);
    q.out = p->tbuf;
    gap.launching();
    PTRY(poly_quotient(c, q));
    PTRY(ntt_device(c, p->tbuf, p->tbuf, p->tmp8, L + p->lq, true, true, n8));
  }
  if (p->qf == 4) {   // de-alias the 4n interpolation with the host-computed low coefficients (see quotient_low)
    HIP_TRY(hipEventSynchronize(p->ev_pi));   // wire lows were complete at the round-1 synchronisation
    HIP_TRY(hipEventSynchronize(p->ev_zlow)); // z's come from the side stream when z is committed from its evaluations
    QuotientLowIn qi;
    qi.low = p->low_host;
    qi.alpha = alpha; qi.beta = beta; qi.gamma = gamma;
    qi.range_ch = range_ch; qi.logic_ch = logic_ch; qi.fixed_ch = fixed_ch; qi.var_ch = var_ch;
    qi.edwards_d = edwards_d; qi.omega = omega; qi.n_inv = p->n_inv;
    Fr tlow[7];
    quotient_low(p->key_low, p->has, qi, tlow);
    PTRY(poly_dealias(c, p->tbuf, n8, tlow, p->gq_inv));
  }
  HIP_TRY(hipMemcpyAsync(p->flag_host, p->flag_dev, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  uint64_t len4;
  if (p->qf == 4) {
    // t has 4n + 7 coefficients by construction (the top one is a product of blinding scalars);
    // explicit zeros at the top would not change the commitment, so no length scan and no
    // synchronisation here — the permutation flag is checked at the next one
    len4 = n + 7;
  } else {
    PTRY(poly_trimmed_len(c, p->tbuf, n8, p->len_dev));
    HIP_TRY(hipMemcpyAsync(p->len_host, p->len_dev, sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    const uint64_t tlen = *p->len_host;
    if (tlen > 7 * n) return PLONK_ERR_UNSAT;                // quotient_poly.rs:132
    len4 = tlen > 3 * n ? tlen - 3 * n : 0;
  }
  // honest quotient: degree <= 4n + 6 (quotient_poly.rs:106-111) => t_fourth has <= n + 7 coefficients;
  // anything longer cannot be committed with the trimmed key (key.rs:362-370)
  if (len4 > n + 7 || len4 > p->srs_total) return PLONK_ERR_DEGREE;
  {
    SplitArgs sa;
    sa.b[0] = bl[11]; sa.b[1] = bl[12]; sa.b[2] = bl[13];
    sa.len4 = len4;
    PTRY(poly_split_t(c, p->tbuf, n, np, p->tparts, sa));
  }
  Fr* t4 = p->tbuf + 3 * n;
  const uint64_t t4_len = len4 ? len4 : 1;                 // t_fourth[0] -= b14 even when the tail is empty
  {
    const Fr* sc[4] = {p->tparts, p->tparts + np, p->tparts + 2 * np, t4};
    const uint64_t ms[4] = {n + 1, n + 1, n + 1, t4_len};
    PTRY(msm_group(p, sc, ms, 4, 5));
  }
  PTRY(fetch_commitments(p, 5, 4, comm + 5));
  if (*p->flag_host) return (plonk::set_last_error("invalid argument", __func__, __FILE__, __LINE__), PLONK_ERR_ARG);   // "permutation denominator must be nonzero" (permutation.rs:231-234)
  tr.append_commitment("t_low_comm", comm[5]);
  tr.append_commitment("t_mid_comm", comm[6]);
  tr.append_commitment("t_high_comm", comm[7]);
  tr.append_commitment("t_fourth_comm", comm[8]);

  // ---- round 4 (prover.rs:591-676): 15 evaluations
  const Fr z_ch = tr.challenge_scalar("z_challenge");
  const Fr zw = z_ch * omega;
  Evals ev;
  {
    EvalArgs ea;
    ea.partial = p->evpart;
    ea.max_blocks = p->ev_max_blocks;
    const Fr* P = p->polys;
    const Fr* pol[15] = {p->wpoly, p->wpoly + np, p->wpoly + 2 * np, p->wpoly + 3 * np,
                         p->wpoly, p->wpoly + np, p->wpoly + 3 * np,
                         P + P_QARITH * np, P + P_QC * np, P + P_QL * np, P + P_QR * np,
                         P + P_S1 * np, P + P_S2 * np, P + P_S3 * np, p->zpoly};
    const uint64_t len[15] = {n + 2, n + 2, n + 2, n + 2, n + 2, n + 2, n + 2,
                              p->poly_len[P_QARITH], p->poly_len[P_QC], p->poly_len[P_QL], p->poly_len[P_QR],
                              p->poly_len[P_S1], p->poly_len[P_S2], p->poly_len[P_S3], n + 3};
    for (int k = 0; k < 15; ++k) {
      ea.items[k].poly = pol[k];
      ea.items[k].len = len[k];
      ea.items[k].x = (k >= 4 && k <= 6) || k == 14 ? zw : z_ch;
    }
    gap.launching();
    PTRY(poly_eval(c, ea, 15, n + 3, p->evout));
    HIP_TRY(hipMemcpyAsync(p->ev_host, p->evout, 15 * sizeof(Fr), hipMemcpyDeviceToHost, c->stream));
    const auto t_before = HostGap::clock::now();
    HIP_TRY(hipStreamSynchronize(c->stream));
    gap.synced(t_before);
    const Fr* h = p->ev_host;
    ev.a = h[0]; ev.b = h[1]; ev.c = h[2]; ev.d = h[3]; ev.a_w = h[4]; ev.b_w = h[5]; ev.d_w = h[6];
    ev.q_arith = h[7]; ev.q_c = h[8]; ev.q_l = h[9]; ev.q_r = h[10]; ev.s1 = h[11]; ev.s2 = h[12]; ev.s3 = h[13]; ev.z = h[14];
  }
  append_evaluations(tr, ev);

  // ---- round 5 (prover.rs:678-739)
  const Fr v = tr.challenge_scalar("v_challenge");
  const Fr one = Fr::one();
  const Fr z_n = z_ch.pow_u64(n);
  const Fr zh = z_n - one;                                           // evaluate_vanishing_polynomial
  const Fr n_inv = p->n_inv;
  // 1/z, 1/(z - 1) and 1/(z omega) from one shared inversion
  const bool z_zero = z_ch.is_zero();
  Fr inv_z, inv_zm1;
  {
    const Fr zm1 = z_ch - one;
    const Fr a = z_zero ? one : z_ch, b = zm1.is_zero() ? one : zm1;
    const Fr iab = fr_inv_gcd(a * b);
    inv_z = iab * b;
    inv_zm1 = iab * a;
  }
  if (z_zero || zw.is_zero()) return PLONK_ERR_STATE;   // probability 2^-255; (X - 0) division is a shift — not worth a code path
  // W_z's commitment is not absorbed before v_w is drawn (prover.rs:727-730): both opening witnesses are committed as one
  // group, and the second one — W_zw = (z + v_w a + v_w^2 b + v_w^3 d) / (X - z w), which needs nothing but z and v_w — is
  // built on the SIDE stream (round 6, second session) while the host is still computing the linearisation scalars of the
  // first and the main stream builds that: below 2^18 gates the two chains of ~6 latency-bound kernels ran back to back.
  // Buffers: the transform scratch (no transform runs after round 3), the upper half of `scratch`, the upper half of `totals`.
  const Fr v_w = tr.challenge_scalar("v_w_challenge");
  {
    gap.launching();
    SideScope side(c, p->ev_ready);
    LinCombArgs la;
    la.t[0].p = p->zpoly; la.t[0].len = n + 3; la.t[0].s = one;
    la.t[1].p = p->wpoly; la.t[1].len = n + 2; la.t[1].s = v_w;
    la.t[2].p = p->wpoly + np; la.t[2].len = n + 2; la.t[2].s = v_w.sqr();
    la.t[3].p = p->wpoly + 3 * np; la.t[3].len = n + 2; la.t[3].s = v_w.sqr() * v_w;
    la.count = 4;
    la.len = np - 1;
    la.constant = Fr::zero();
    la.out = p->tmp8;
    PTRY(poly_lincomb(c, la));
    PTRY(poly_ruffini(c, p->tmp8, p->wit2, np - 1, zw, inv_z * p->omega_inv, p->scratch + np, p->totals + 8192));
    HIP_TRY(hipEventRecord(p->ev_side, c->side_stream));
  }
  const Fr pi_eval = public_input_eval(p, pi_idx, pi_val, pi_count, z_ch, zh);
  // permutation linearisation scalars (permutation/proverkey.rs:127-269)
  const Fr bz = beta * z_ch;
  const Fr lin_a = (ev.a + bz + gamma) * (ev.b + fr_small(7) * bz + gamma) * (ev.c + fr_small(13) * bz + gamma) *
                   (ev.d + fr_small(17) * bz + gamma) * alpha;
  const Fr lin_b = (ev.a + beta * ev.s1 + gamma) * (ev.b + beta * ev.s2 + gamma) * (ev.c + beta * ev.s3 + gamma) *
                   (beta * ev.z) * alpha;
  Fr l1_z;                                                            // evaluate_all_lagrange_coefficients(z)[0]
  if (z_n == one) l1_z = (z_ch == one) ? one : Fr::zero();
  else l1_z = zh * n_inv * inv_zm1;
  const Fr c_range = range_identity(range_ch, ev) * range_ch;
  const Fr c_logic = logic_identity(logic_ch, ev) * logic_ch;
  const Fr c_fixed = fixed_identity(fixed_ch, ev, edwards_d) * fixed_ch;
  const Fr c_var = var_identity(var_ch, ev, edwards_d) * var_ch;
  const Fr nzh = zh.neg();                                            // z_h_eval = -(z^n - 1)
  Fr vp[12];
  vp[0] = one;
  for (int k = 1; k < 12; ++k) vp[k] = vp[k - 1] * v;
  // W_z numerator = r + v a + v^2 b + v^3 c + v^4 d + v^5 s1 + v^6 s2 + v^7 s3 + v^8 q_arith + v^9 q_c
  //                 + v^10 q_l + v^11 q_r  (prover.rs:706-726) with r expanded into its terms
  {
    LinCombArgs la;
    int k = 0;
    const Fr* P = p->polys;
    auto term = [&](const Fr* ptr, uint64_t len, const Fr& s) { la.t[k].p = ptr; la.t[k].len = len; la.t[k].s = s; ++k; };
    term(P + P_QM * np, p->poly_len[P_QM], ev.q_arith * ev.a * ev.b);
    term(P + P_QL * np, p->poly_len[P_QL], ev.q_arith * ev.a + vp[10]);
    term(P + P_QR * np, p->poly_len[P_QR], ev.q_arith * ev.b + vp[11]);
    term(P + P_QO * np, p->poly_len[P_QO], ev.q_arith * ev.c);
    term(P + P_QF * np, p->poly_len[P_QF], ev.q_arith * ev.d);
    term(P + P_QC * np, p->poly_len[P_QC], ev.q_arith + vp[9]);
    term(P + P_QARITH * np, p->poly_len[P_QARITH], vp[8]);
    term(P + P_QRANGE * np, p->poly_len[P_QRANGE], c_range);
    term(P + P_QLOGIC * np, p->poly_len[P_QLOGIC], c_logic);
    term(P + P_QFIXED * np, p->poly_len[P_QFIXED], c_fixed);
    term(P + P_QVAR * np, p->poly_len[P_QVAR], c_var);
    term(P + P_S1 * np, p->poly_len[P_S1], vp[5]);
    term(P + P_S2 * np, p->poly_len[P_S2], vp[6]);
    term(P + P_S3 * np, p->poly_len[P_S3], vp[7]);
    term(P + P_S4 * np, p->poly_len[P_S4], lin_b.neg());
    term(p->zpoly, n + 3, lin_a + l1_z * alpha.sqr());
    term(p->wpoly, n + 2, vp[1]);
    term(p->wpoly + np, n + 2, vp[2]);
    term(p->wpoly + 2 * np, n + 2, vp[3]);
    term(p->wpoly + 3 * np, n + 2, vp[4]);
    term(p->tparts, n + 1, nzh);
    term(p->tparts + np, n + 1, nzh * z_n);
    term(p->tparts + 2 * np, n + 1, nzh * z_n.sqr());
    term(t4, t4_len, nzh * z_n.sqr() * z_n);
    la.count = k;
    la.len = np - 1;
    la.constant = pi_eval;
    la.out = p->agg;
    gap.launching();
    PTRY(poly_lincomb(c, la));
  }
  PTRY(poly_ruffini(c, p->agg, p->wit, np - 1, z_ch, inv_z, p->scratch, p->totals));
  // ruffini's suffix scan leaves sum_j c_j z^j = (W_z numerator)(z) in scratch[0]: keep it for the
  // quotient-identity check below
  HIP_TRY(hipMemcpyAsync(p->evout + 15, p->scratch, sizeof(Fr), hipMemcpyDeviceToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(p->ev_host + 15, p->evout + 15, sizeof(Fr), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamWaitEvent(c->main_stream, p->ev_side, 0));   // W_zw (side stream, above)
  {
    const Fr* sc[2] = {p->wit, p->wit2};
    const uint64_t ms[2] = {np - 2, np - 2};
    PTRY(msm_group(p, sc, ms, 2, 9));
  }
  PTRY(fetch_commitments(p, 9, 2, comm + 9));
  if (!quotient_identity_holds(p->ev_host[15], ev, alpha, beta, gamma, l1_z, vp)) return PLONK_ERR_UNSAT;

  // ---- Proof::to_bytes (proof.rs:137-162, linearization_poly.rs:98-124)
  write_proof(proof, comm, ev);
  return PLONK_OK;
}

// ---- multi-GPU prove(): the same transcript on every rank, the work split as described at
// Prover::sharded.  Replicated: the wire / permutation polynomials (rounds 1-2: 5 size-n inverse
// transforms and the grand product).  Sharded: every MSM (SRS point range), the quotient (residue
// classes of the coset), evaluations, linearisation and opening quotients (coefficient range).
// Exchanges per proof: 4 all-gathers of MSM partial sums, 1 all-to-all of the class remainders,
// 1 all-gather of 15 partial evaluations, 1 all-gather of two suffix-sum carries.
static int prover_prove_sharded(Prover* p, const Fr* wires_dev, const uint64_t* pi_idx, const Fr* pi_val, uint64_t pi_count,
                                const Fr* bl, uint8_t proof[1008]) {
  Ctx* c = p->c;
  if (p->srs_gen != c->srs_gen) return (set_last_error("prover is bound to an SRS that was replaced on its context", __func__, __FILE__, __LINE__), PLONK_ERR_STATE);
  const uint64_t n = p->n, np = p->np, qn = p->qn;
  const uint32_t L = p->logn, W = (uint32_t)p->world, cpr = p->cpr, Q = p->Q;
  const uint64_t lo = p->lo, hi = p->hi;                       // owned coefficient indices, hi <= n + 7
  NttTables* tbn;
  PTRY(ntt_tables(c, L, false, &tbn));
  const Fr omega = p->omega, one = Fr::one();
  auto own_len = [&](uint64_t len) -> uint64_t {               // coefficients of a length-`len` polynomial inside [lo, hi)
    const uint64_t e = len < hi ? len : hi;
    return e > lo ? e - lo : 0;
  };

  Transcript tr((const uint8_t*)p->label.data(), p->label.size());
  seed_transcript(tr, p, pi_val, pi_count);
  HostGap gap{c};   // host time between the device phases of this rank (slots 8-10 of plonk_profile_read; the sharded proof has up to seven synchronisations)
  GapScope gap_scope(&gap);

  uint8_t comm[11][48];
  // ---- round 1 (replicated polynomials, sharded commitments)
  const bool lag = p->lag_on;
  // iNTT + blinding of the four columns and their lowest coefficients (quotient_low), on the CURRENT stream
  auto wire_polynomials = [&](Fr* ntt_tmp) -> int {
    prof_begin(c, 4);   // slot 4: replicated polynomial work (rounds 1-2)
    for (int k = 0; k < 4; ++k) {
      Fr* wp = p->wpoly + k * np;
      if (p->wires_pending) HIP_TRY(hipStreamWaitEvent(c->stream, p->ev_wire[k], 0));   // column k has arrived
      PTRY(ntt_device(c, wires_dev + k * n, wp, ntt_tmp, L, true, false, n));
      BlindArgs ba;
      ba.count = 2;
      ba.b[0] = bl[2 * k];
      ba.b[1] = bl[2 * k + 1];
      PTRY(poly_fill_zero(c, wp + n, np - n));
      PTRY(poly_blind(c, wp, n, ba));
    }
    prof_end(c, 4);
    for (int k = 0; k < 4; ++k)
      HIP_TRY(hipMemcpyAsync(p->low_host + 7 * k, p->wpoly + k * np, 7 * sizeof(Fr), hipMemcpyDeviceToHost, c->stream));
    return PLONK_OK;
  };
  // Round 4 (r04): the schedule of the single-GPU path for a RANK.  With the Lagrange-basis slice the wire commitments read
  // the wire VALUES, so the replicated wire polynomials (needed from round 3 on) ride on the side stream under the
  // commitment pipeline — a rank's MSM over 1/W of the points leaves most of the chip idle in its latency-bound sort and
  // reduction tail (W = 8 at 2^20: 1.4 of the rank's 7.7 ms were these transforms in front of the commitment).  The side
  // work starts after the group's accumulation (side_defer) while the rank's share is small; PLONK_SHARD_SIDE=0 restores
  // the round-3 order.
  const bool shard_side_env = c->cfg.shard_side >= 0;   // plonk_gpu_config.shard_side_stream = -1 / PLONK_SHARD_SIDE=0 restores the round-3 order
  const uint64_t rank_pts = hi - lo;
  const bool polys_on_side = lag && shard_side_env && rank_pts <= (1ull << 18) + 64;   // (2^19 points per rank: 19.2 -> 19.4 ms, the accumulation owns the VALU)
  const bool side_defer = shard_side_env && rank_pts <= (1ull << 18) + 64;
  if (!polys_on_side) PTRY(wire_polynomials(p->tmp8));
  SideJoin side_join{c};
  uint64_t pi_len = 0;
  Fr* fold_side = p->fold + n;
  auto side_round1 = [&]() -> int {
    // class evaluations of a, b, c, d, PI on the side stream: fold mod (X^n - x^n|class), size-n coset transform
    if (polys_on_side) PTRY(wire_polynomials(p->tmp8b));
    for (uint32_t k = 0; k < cpr; ++k)
      for (int w = 0; w < 4; ++w) {
        PTRY(poly_fold(c, p->wpoly + w * np, fold_side, n, 2, p->sigma_j[p->cls[k]]));
        PTRY(ntt_device(c, fold_side, p->cos + (1 + w) * qn + k * n, p->tmp8b, L, false, true, n, &p->cs_fwd[k]));
      }
    PTRY(poly_fill_zero(c, p->pipoly, np));
    if (pi_count) {
      PTRY(public_input_polynomial(p, pi_idx, pi_val, pi_count, p->tmp8b));
      pi_len = n;
    }
    HIP_TRY(hipMemcpyAsync(p->low_host + 35, p->pipoly, 7 * sizeof(Fr), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipEventRecord(p->ev_pi, c->stream));
    if (pi_len)
      for (uint32_t k = 0; k < cpr; ++k)
        PTRY(ntt_device(c, p->pipoly, p->cos + 5 * qn + k * n, p->tmp8b, L, false, true, n, &p->cs_fwd[k]));
    return PLONK_OK;
  };
  if (!side_defer) {
    SideScope side(c, p->ev_ready);
    PTRY(side_round1());
  }
  {
    AccMark mark(c, side_defer ? p->ev_acc : nullptr);
    if (lag && p->lag_whole) {
      // by commitment (round 5): this rank's whole columns over the whole Lagrange-basis key, exactly the single-GPU launch
      // for those columns; the slots of the columns it does not own hold the identity (all-zero bit sums), so the all-gather
      // + add of fetch_commitments needs no change
      const int per_rank = 4 / (int)W, first = per_rank * p->rank;
      for (int k = 0; k < 4; ++k)
        if (p->wires_pending) HIP_TRY(hipStreamWaitEvent(c->stream, p->ev_wire[k], 0));
      HIP_TRY(hipMemcpyAsync(p->wscal, bl, 8 * sizeof(Fr), hipMemcpyHostToDevice, c->stream));
      HIP_TRY(hipMemsetAsync(p->res, 0, (size_t)RES_STRIDE * 4, c->stream));
      for (int k = 0; k < 4; ++k) { p->res_bitpos[k] = false; p->res_rowbits[k] = 8; }
      const Fr* sc[2];
      const Fr* tl[2];
      uint64_t ms[2], sp[2];
      for (int j = 0; j < per_rank; ++j) {
        sc[j] = wires_dev + (uint64_t)(first + j) * n;
        tl[j] = p->wscal + 2 * (first + j);
        ms[j] = n + 2;
        sp[j] = n;
      }
      PTRY(msm_group(p, sc, ms, per_rank, first, p->lag_table, p->lag_n, tl, sp));
    } else if (lag) {
      // wire commitments from the wire VALUES over this rank's slice of the Lagrange-basis key (see prover_prove): values
      // [lo_L, hi_L) of each column in place, the column's two blinders for the slice that reaches indices n, n + 1
      const uint64_t lo_l = p->shard_lo, hi_l = p->shard_lo + p->lag_n;       // hi_l <= n + 2
      for (int k = 0; k < 4; ++k)
        if (p->wires_pending) HIP_TRY(hipStreamWaitEvent(c->stream, p->ev_wire[k], 0));
      HIP_TRY(hipMemcpyAsync(p->wscal, bl, 8 * sizeof(Fr), hipMemcpyHostToDevice, c->stream));
      const Fr* sc[4];
      const Fr* tl[4];
      uint64_t ms[4], sp[4];
      for (int k = 0; k < 4; ++k) {
        sc[k] = wires_dev + (uint64_t)k * n + (lo_l < n ? lo_l : 0);
        sp[k] = lo_l < n ? n - lo_l : 0;                                      // scalars of the slice below index n
        if (sp[k] > p->lag_n) sp[k] = p->lag_n;
        tl[k] = p->wscal + 2 * k + (lo_l > n ? lo_l - n : 0);                 // blinder b_(index - n)
        ms[k] = hi_l - lo_l;
      }
      PTRY(msm_group(p, sc, ms, 4, 0, p->lag_table, p->lag_n, tl, sp));
    } else {
      const Fr* sc[4] = {p->wpoly, p->wpoly + np, p->wpoly + 2 * np, p->wpoly + 3 * np};
      const uint64_t ms[4] = {n + 2, n + 2, n + 2, n + 2};
      PTRY(msm_group(p, sc, ms, 4, 0));
    }
  }
  if (side_defer) {
    SideScopeAfter side(c, p->ev_acc);
    PTRY(side_round1());
  }
  PTRY(fetch_commitments(p, 0, 4, comm));
  tr.append_commitment("a_comm", comm[0]);
  tr.append_commitment("b_comm", comm[1]);
  tr.append_commitment("c_comm", comm[2]);
  tr.append_commitment("d_comm", comm[3]);

  // ---- round 2 (replicated grand product)
  const Fr beta = tr.challenge_scalar("beta");
  tr.append_scalar("beta", beta);
  const Fr gamma = tr.challenge_scalar("gamma");
  {
    prof_begin(c, 4);
    PermArgs pa;
    pa.n = n;
    for (int k = 0; k < 4; ++k) { pa.wires[k] = wires_dev + k * n; pa.sigma[k] = p->sigma_n + k * n; }
    pa.beta = beta; pa.gamma = gamma;
    pa.ks[0] = Fr::one(); pa.ks[1] = fr_small(7); pa.ks[2] = fr_small(13); pa.ks[3] = fr_small(17);
    pa.tw_lo29 = tbn->tw_lo29; pa.tw_hi29 = tbn->tw_hi29; pa.lobits = L < 13 ? L : 13; pa.use_hi = L > 13;
    pa.num = p->scratch; pa.den = p->scratch + np;
    gap.launching();
    HIP_TRY(hipMemsetAsync(p->flag_dev, 0, sizeof(int), c->stream));
    // Round 4: the grand product (permutation.rs:213-294) split over the ranks from 2^19 gates and four ranks on.  Rank r forms the n / W
    // numerator / denominator terms of ITS evaluation indices, inverts, multiplies and scans them locally; the ranks exchange
    // their range products (36 bytes each: the product in twiddle form + the zero-denominator flag), every rank scales its
    // range by the product of the ranges before it, and the z evaluations are all-gathered in place (32 n / W bytes per
    // rank; 0.5 MB per peer link at 2^20 gates and W = 8) for the one replicated step left, z's inverse transform.
    // PLONK_SHARD_Z=0 / 1 forces either (the multi-rank tests run small circuits through both).
    const int shard_z_env = c->cfg.shard_z > 0 ? 1 : (c->cfg.shard_z < 0 ? 0 : -1);   // plonk_gpu_config.shard_grand_product / PLONK_SHARD_Z
    // rank alone, loop-back collectives (profiles/r04): W = 8 at 2^20 gates 7.41 -> 7.01 ms, at 2^22 21.9 -> 20.1; W = 2 at 2^20
    // 19.05 -> 19.27 (half of 0.5 ms of work against two more host round trips): from four ranks on
    const bool shard_z = (shard_z_env >= 0 ? shard_z_env == 1 : (L >= 19 && W >= 4)) && n % W == 0 && n / W >= 2;
    if (!shard_z) {
      PTRY(poly_perm_terms(c, pa));
      PTRY(poly_batch_inverse(c, pa.den, n, true));
      PTRY(poly_mul_arrays(c, pa.num, pa.den, n, p->flag_dev));
      PTRY(scan_prefix_product(c, pa.num, n, p->totals));
    } else {
      const uint64_t cnt = n / W, first = cnt * (uint64_t)p->rank;
      pa.first = first; pa.count = cnt;
      PTRY(poly_perm_terms(c, pa));
      PTRY(poly_batch_inverse(c, pa.den + first, cnt, true));
      PTRY(poly_mul_arrays(c, pa.num + first, pa.den + first, cnt, p->flag_dev));
      PTRY(scan_prefix_product_local(c, pa.num + first, cnt, p->totals));
      struct { Fr total; int flag; int pad[7]; } mine, *all;
      static_assert(sizeof(mine) == 64, "exchange record");
      memset(&mine, 0, sizeof mine);
      HIP_TRY(hipMemcpyAsync(p->ev_host, p->totals + (scan_prefix_blocks(cnt) - 1), sizeof(Fr), hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(hipMemcpyAsync(p->flag_host, p->flag_dev, sizeof(int), hipMemcpyDeviceToHost, c->stream));
      const auto t_before = HostGap::clock::now();
      PTRY(comm_sync(c, c->stream));
      gap.synced(t_before);
      mine.total = p->ev_host[0];
      mine.flag = *p->flag_host;
      std::vector<uint8_t> gathered(sizeof(mine) * (size_t)W);
      PTRY(comm_allgather_host(c, p->link, &mine, gathered.data(), sizeof(mine)));
      all = reinterpret_cast<decltype(all)>(gathered.data());
      // products in twiddle form (raw integer x * 2^261 = the Montgomery form of 32 x): a * b * (1 / 32) stays in that form
      Fr carry = Fr::from_u64(32);
      int any_zero = 0;
      for (uint32_t r = 0; r < W; ++r) {
        any_zero |= all[r].flag;
        if (r < (uint32_t)p->rank) carry = carry * all[r].total * p->inv32;
      }
      if (any_zero) return (plonk::set_last_error("invalid argument", "zero denominator in the permutation grand product", __FILE__, __LINE__), PLONK_ERR_ARG);
      gap.launching();
      PTRY(scan_prefix_product_apply(c, pa.num + first, cnt, p->totals, carry));
      PTRY(comm_allgather_dev(c, p->link, pa.num, sizeof(Fr) * cnt));
    }
    PTRY(ntt_device(c, pa.num, p->zpoly, p->tmp8, L, true, false, n));
    BlindArgs ba;
    ba.count = 3;
    ba.b[0] = bl[8]; ba.b[1] = bl[9]; ba.b[2] = bl[10];
    PTRY(poly_fill_zero(c, p->zpoly + n, np - n));
    PTRY(poly_blind(c, p->zpoly, n, ba));
    prof_end(c, 4);
  }
  HIP_TRY(hipMemcpyAsync(p->low_host + 28, p->zpoly, 7 * sizeof(Fr), hipMemcpyDeviceToHost, c->stream));
  auto side_round2 = [&]() -> int {
    for (uint32_t k = 0; k < cpr; ++k) {
      PTRY(poly_fold(c, p->zpoly, fold_side, n, 3, p->sigma_j[p->cls[k]]));
      PTRY(ntt_device(c, fold_side, p->cos + k * n, p->tmp8b, L, false, true, n, &p->cs_fwd[k]));
    }
    HIP_TRY(hipEventRecord(p->ev_side, c->side_stream));
    return PLONK_OK;
  };
  if (!side_defer) {
    SideScope side(c, p->ev_ready);
    PTRY(side_round2());
  }
  {
    AccMark mark(c, side_defer ? p->ev_acc : nullptr);
    PTRY(msm_to(p, p->zpoly, n + 3, 4));
  }
  if (side_defer) {
    SideScopeAfter side(c, p->ev_acc);
    PTRY(side_round2());
  }
  PTRY(fetch_commitments(p, 4, 1, comm + 4));
  tr.append_commitment("z_comm", comm[4]);

  // ---- round 3: quotient on the owned classes
  const Fr alpha = tr.challenge_scalar("alpha");
  const Fr range_ch = tr.challenge_scalar("range separation challenge");
  const Fr logic_ch = tr.challenge_scalar("logic separation challenge");
  const Fr fixed_ch = tr.challenge_scalar("fixed base separation challenge");
  const Fr var_ch = tr.challenge_scalar("variable base separation challenge");
  const Fr edwards_d = p->edwards_d;
  gap.launching();
  HIP_TRY(hipStreamWaitEvent(c->main_stream, p->ev_side, 0));
  for (uint32_t k = 0; k < cpr; ++k) {
    QuotientArgs q;
    q.n8 = n;
    q.rot = 1;                                                 // X -> omega X is the next point of the class
    const Fr* co = p->cos + k * n;
    q.z = co; q.a = co + qn; q.b = co + 2 * qn; q.c = co + 3 * qn; q.d = co + 4 * qn; q.pi = pi_len ? co + 5 * qn : nullptr;
    const Fr* e = p->evals8 + k * n;
    q.q_m = e + P_QM * qn; q.q_l = e + P_QL * qn; q.q_r = e + P_QR * qn; q.q_o = e + P_QO * qn; q.q_f = e + P_QF * qn;
    q.q_c = e + P_QC * qn; q.q_arith = e + P_QARITH * qn; q.q_range = e + P_QRANGE * qn; q.q_logic = e + P_QLOGIC * qn;
    q.q_fixed = e + P_QFIXED * qn; q.q_var = e + P_QVAR * qn;
    q.s1 = e + P_S1 * qn; q.s2 = e + P_S2 * qn; q.s3 = e + P_S3 * qn; q.s4 = e + P_S4 * qn;
    q.linear = e + P_COUNT * qn; q.l1 = e + (P_COUNT + 1) * qn;
    for (int s = 0; s < QS_COUNT; ++s) q.has[s] = p->has[s];
    q.range_ch = range_ch; q.logic_ch = logic_ch; q.fixed_ch = fixed_ch; q.var_ch = var_ch;
    q.edwards_d = edwards_d;
    q.inv32 = p->inv32;
    quotient_data(gamma, q.k.gamma);
    quotient_data(Fr::one(), q.k.one);
    const Fr ks[4] = {Fr::one(), fr_small(7), fr_small(13), fr_small(17)};
    for (int i = 0; i < 4; ++i) quotient_const(beta * ks[i], 0, q.k.beta_k[i]);
    quotient_const(alpha, 20, q.k.alpha_pos);
    quotient_const(alpha.neg(), 20, q.k.alpha_neg);
    quotient_const(alpha.sqr(), 0, q.k.alpha_sq);
    for (int i = 0; i < 8; ++i) quotient_const(p->vinv[p->cls[k]], 0, q.k.vinv[i]);   // 1 / (x^n - 1): constant on a class
    q.out = p->Fbuf + (uint64_t)k * n;
    PTRY(poly_quotient(c, q));
    // t mod (X^n - x^n|class): size-n inverse coset transform with the class shift
    PTRY(ntt_device(c, q.out, q.out, p->tmp8, L, true, true, n, &p->cs_inv[k]));
  }
  Fr* t4 = p->tbuf;                                            // t_fourth, indexed by coefficient
  {
    ShardPackArgs pk{};
    for (uint32_t k = 0; k < cpr; ++k) pk.F[k] = p->Fbuf + (uint64_t)k * n;
    pk.send = p->send; pk.n = n; pk.per = p->per; pk.stride = p->stride; pk.cpr = cpr;
    PTRY(poly_shard_pack(c, pk, W));
    PTRY(comm_alltoall_dev(c, p->link, p->send, p->recv, sizeof(Fr) * (size_t)cpr * p->stride));
    ShardCombineArgs cb{};
    cb.recv = p->recv; cb.stride = p->stride; cb.per = p->per; cb.lo = lo; cb.hi = hi; cb.n = n;
    cb.cnt = (hi < n ? hi : n) > lo ? (hi < n ? hi : n) - lo : 0;
    cb.W = W; cb.cpr = cpr; cb.Q = Q;
    for (int i1 = 0; i1 < 5; ++i1) for (int j = 0; j < 8; ++j) cb.coef[i1][j] = p->coef[i1][j];
    if (Q == 4) {   // de-alias with the host-computed low coefficients, as on one GPU (quotient_low)
      HIP_TRY(hipEventSynchronize(p->ev_pi));
      QuotientLowIn qi;
      qi.low = p->low_host;
      qi.alpha = alpha; qi.beta = beta; qi.gamma = gamma;
      qi.range_ch = range_ch; qi.logic_ch = logic_ch; qi.fixed_ch = fixed_ch; qi.var_ch = var_ch;
      qi.edwards_d = edwards_d; qi.omega = omega; qi.n_inv = p->n_inv;
      quotient_low(p->key_low, p->has, qi, cb.low);
    }
    cb.g4n_inv = p->gq_inv;
    cb.parts[0] = p->tparts; cb.parts[1] = p->tparts + np; cb.parts[2] = p->tparts + 2 * np; cb.parts[3] = t4;
    PTRY(poly_shard_combine(c, cb));
    ShardSplitFix sf{};
    for (int i = 0; i < 4; ++i) sf.parts[i] = cb.parts[i];
    sf.lo = lo; sf.hi = hi; sf.n = n;
    sf.b[0] = bl[11]; sf.b[1] = bl[12]; sf.b[2] = bl[13];
    PTRY(poly_shard_split_fix(c, sf));
  }
  HIP_TRY(hipMemcpyAsync(p->flag_host, p->flag_dev, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  const uint64_t t4_len = n + 7;   // t has 4n + 7 coefficients by construction (prover.hip, 4n path); explicit zeros do not change a commitment
  {
    const Fr* sc[4] = {p->tparts, p->tparts + np, p->tparts + 2 * np, t4};
    const uint64_t ms[4] = {n + 1, n + 1, n + 1, t4_len};
    PTRY(msm_group(p, sc, ms, 4, 5));
  }
  PTRY(fetch_commitments(p, 5, 4, comm + 5));
  if (*p->flag_host) return (plonk::set_last_error("invalid argument", __func__, __FILE__, __LINE__), PLONK_ERR_ARG);
  tr.append_commitment("t_low_comm", comm[5]);
  tr.append_commitment("t_mid_comm", comm[6]);
  tr.append_commitment("t_high_comm", comm[7]);
  tr.append_commitment("t_fourth_comm", comm[8]);

  // ---- round 4: every rank evaluates its coefficient range, partial sums are all-gathered
  const Fr z_ch = tr.challenge_scalar("z_challenge");
  const Fr zw = z_ch * omega;
  Evals ev;
  {
    EvalArgs ea;
    ea.partial = p->evpart;
    ea.max_blocks = p->ev_max_blocks;
    const Fr* P = p->polys;
    const Fr* pol[15] = {p->wpoly, p->wpoly + np, p->wpoly + 2 * np, p->wpoly + 3 * np,
                         p->wpoly, p->wpoly + np, p->wpoly + 3 * np,
                         P + P_QARITH * np, P + P_QC * np, P + P_QL * np, P + P_QR * np,
                         P + P_S1 * np, P + P_S2 * np, P + P_S3 * np, p->zpoly};
    const uint64_t len[15] = {n + 2, n + 2, n + 2, n + 2, n + 2, n + 2, n + 2,
                              p->poly_len[P_QARITH], p->poly_len[P_QC], p->poly_len[P_QL], p->poly_len[P_QR],
                              p->poly_len[P_S1], p->poly_len[P_S2], p->poly_len[P_S3], n + 3};
    uint64_t max_len = 1;
    for (int k = 0; k < 15; ++k) {
      ea.items[k].poly = pol[k] + lo;
      ea.items[k].len = own_len(len[k]);
      ea.items[k].x = (k >= 4 && k <= 6) || k == 14 ? zw : z_ch;
      if (ea.items[k].len > max_len) max_len = ea.items[k].len;
    }
    gap.launching();
    PTRY(poly_eval(c, ea, 15, max_len, p->evout));
    HIP_TRY(hipMemcpyAsync(p->ev_host, p->evout, 15 * sizeof(Fr), hipMemcpyDeviceToHost, c->stream));
    const auto t_before = HostGap::clock::now();
    PTRY(comm_sync(c, c->stream));
    gap.synced(t_before);   // the all-to-all of the quotient precedes this on the stream: never hang on a dead peer
    std::vector<Fr> all(15 * (size_t)W);
    PTRY(comm_allgather_host(c, p->link, p->ev_host, all.data(), 15 * sizeof(Fr)));
    Fr h[15];
    for (int k = 0; k < 15; ++k) h[k] = Fr::zero();
    // sum_r x^(lo_r) * partial_r.  The powers step by x^per from rank to rank (one product each); only a clamped range start
    // takes an exponentiation of its own — the host sits between two device phases here (2 W square-and-multiply chains of
    // 64 steps were ~0.15 ms of every 8-rank proof)
    const Fr step_z = z_ch.pow_u64(p->per), step_zw = zw.pow_u64(p->per);
    Fr pz = Fr::one(), pzw = Fr::one();
    for (uint32_t r = 0; r < W; ++r) {
      const uint64_t lo_r = p->per * r;
      if (lo_r > n + 7) { pz = z_ch.pow_u64(n + 7); pzw = zw.pow_u64(n + 7); }
      for (int k = 0; k < 15; ++k) h[k] = h[k] + all[15 * r + k] * (((k >= 4 && k <= 6) || k == 14) ? pzw : pz);
      pz = pz * step_z;
      pzw = pzw * step_zw;
    }
    ev.a = h[0]; ev.b = h[1]; ev.c = h[2]; ev.d = h[3]; ev.a_w = h[4]; ev.b_w = h[5]; ev.d_w = h[6];
    ev.q_arith = h[7]; ev.q_c = h[8]; ev.q_l = h[9]; ev.q_r = h[10]; ev.s1 = h[11]; ev.s2 = h[12]; ev.s3 = h[13]; ev.z = h[14];
  }
  append_evaluations(tr, ev);

  // ---- round 5: linearisation + both opening quotients on the owned coefficient range
  const Fr v = tr.challenge_scalar("v_challenge");
  const Fr v_w = tr.challenge_scalar("v_w_challenge");        // W_z's commitment is not absorbed before it (prover.rs:727-730)
  const Fr z_n = z_ch.pow_u64(n);
  const Fr zh = z_n - one;
  const Fr n_inv = p->n_inv;
  if (z_ch.is_zero() || zw.is_zero()) return PLONK_ERR_STATE;   // probability 2^-255
  const Fr zm1 = z_ch - one;
  const Fr iab = fr_inv_gcd(z_ch * (zm1.is_zero() ? one : zm1));
  const Fr inv_z = iab * (zm1.is_zero() ? one : zm1), inv_zm1 = iab * z_ch;
  const Fr pi_eval = public_input_eval(p, pi_idx, pi_val, pi_count, z_ch, zh);
  const Fr bz = beta * z_ch;
  const Fr lin_a = (ev.a + bz + gamma) * (ev.b + fr_small(7) * bz + gamma) * (ev.c + fr_small(13) * bz + gamma) *
                   (ev.d + fr_small(17) * bz + gamma) * alpha;
  const Fr lin_b = (ev.a + beta * ev.s1 + gamma) * (ev.b + beta * ev.s2 + gamma) * (ev.c + beta * ev.s3 + gamma) *
                   (beta * ev.z) * alpha;
  Fr l1_z;
  if (z_n == one) l1_z = (z_ch == one) ? one : Fr::zero();
  else l1_z = zh * n_inv * inv_zm1;
  const Fr c_range = range_identity(range_ch, ev) * range_ch;
  const Fr c_logic = logic_identity(logic_ch, ev) * logic_ch;
  const Fr c_fixed = fixed_identity(fixed_ch, ev, edwards_d) * fixed_ch;
  const Fr c_var = var_identity(var_ch, ev, edwards_d) * var_ch;
  const Fr nzh = zh.neg();
  Fr vp[12];
  vp[0] = one;
  for (int k = 1; k < 12; ++k) vp[k] = vp[k - 1] * v;
  const uint64_t rlen = hi - lo;                               // owned part of the n + 7 numerator coefficients
  gap.launching();
  {
    LinCombArgs la;
    int k = 0;
    const Fr* P = p->polys;
    auto term = [&](const Fr* ptr, uint64_t len, const Fr& s) { la.t[k].p = ptr + lo; la.t[k].len = own_len(len); la.t[k].s = s; ++k; };
    term(P + P_QM * np, p->poly_len[P_QM], ev.q_arith * ev.a * ev.b);
    term(P + P_QL * np, p->poly_len[P_QL], ev.q_arith * ev.a + vp[10]);
    term(P + P_QR * np, p->poly_len[P_QR], ev.q_arith * ev.b + vp[11]);
    term(P + P_QO * np, p->poly_len[P_QO], ev.q_arith * ev.c);
    term(P + P_QF * np, p->poly_len[P_QF], ev.q_arith * ev.d);
    term(P + P_QC * np, p->poly_len[P_QC], ev.q_arith + vp[9]);
    term(P + P_QARITH * np, p->poly_len[P_QARITH], vp[8]);
    term(P + P_QRANGE * np, p->poly_len[P_QRANGE], c_range);
    term(P + P_QLOGIC * np, p->poly_len[P_QLOGIC], c_logic);
    term(P + P_QFIXED * np, p->poly_len[P_QFIXED], c_fixed);
    term(P + P_QVAR * np, p->poly_len[P_QVAR], c_var);
    term(P + P_S1 * np, p->poly_len[P_S1], vp[5]);
    term(P + P_S2 * np, p->poly_len[P_S2], vp[6]);
    term(P + P_S3 * np, p->poly_len[P_S3], vp[7]);
    term(P + P_S4 * np, p->poly_len[P_S4], lin_b.neg());
    term(p->zpoly, n + 3, lin_a + l1_z * alpha.sqr());
    term(p->wpoly, n + 2, vp[1]);
    term(p->wpoly + np, n + 2, vp[2]);
    term(p->wpoly + 2 * np, n + 2, vp[3]);
    term(p->wpoly + 3 * np, n + 2, vp[4]);
    term(p->tparts, n + 1, nzh);
    term(p->tparts + np, n + 1, nzh * z_n);
    term(p->tparts + 2 * np, n + 1, nzh * z_n.sqr());
    term(t4, t4_len, nzh * z_n.sqr() * z_n);
    la.count = k;
    la.len = rlen;
    la.constant = lo == 0 ? pi_eval : Fr::zero();
    la.out = p->agg + lo;
    if (rlen) PTRY(poly_lincomb(c, la));
  }
  PTRY(poly_ruffini_local(c, p->agg + lo, lo, rlen, z_ch, p->scratch, p->totals));
  {
    LinCombArgs la;
    la.t[0].p = p->zpoly + lo; la.t[0].len = own_len(n + 3); la.t[0].s = one;
    la.t[1].p = p->wpoly + lo; la.t[1].len = own_len(n + 2); la.t[1].s = v_w;
    la.t[2].p = p->wpoly + np + lo; la.t[2].len = own_len(n + 2); la.t[2].s = v_w.sqr();
    la.t[3].p = p->wpoly + 3 * np + lo; la.t[3].len = own_len(n + 2); la.t[3].s = v_w.sqr() * v_w;
    la.count = 4;
    la.len = rlen;
    la.constant = Fr::zero();
    la.out = p->agg2 + lo;
    if (rlen) PTRY(poly_lincomb(c, la));
  }
  PTRY(poly_ruffini_local(c, p->agg2 + lo, lo, rlen, zw, p->scratch2, p->totals));
  // the suffix sums of the ranges above this one: all-gather of (total_z, total_zw) per rank
  HIP_TRY(hipMemcpyAsync(p->ev_host, p->scratch, sizeof(Fr), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipMemcpyAsync(p->ev_host + 1, p->scratch2, sizeof(Fr), hipMemcpyDeviceToHost, c->stream));
  const auto t_before_tot = HostGap::clock::now();
  PTRY(comm_sync(c, c->stream));
  gap.synced(t_before_tot);
  std::vector<Fr> tot(2 * (size_t)W);
  PTRY(comm_allgather_host(c, p->link, p->ev_host, tot.data(), 2 * sizeof(Fr)));
  Fr carry_z = Fr::zero(), carry_zw = Fr::zero(), num_at_z = Fr::zero();
  for (uint32_t r = 0; r < W; ++r) {
    num_at_z = num_at_z + tot[2 * r];
    if (r > (uint32_t)p->rank) { carry_z = carry_z + tot[2 * r]; carry_zw = carry_zw + tot[2 * r + 1]; }
  }
  gap.launching();
  PTRY(poly_ruffini_finish(c, p->scratch, p->wit, lo, rlen, inv_z, carry_z, n + 6));
  PTRY(poly_ruffini_finish(c, p->scratch2, p->wit2, lo, rlen, inv_z * p->omega_inv, carry_zw, n + 6));
  {
    const Fr* sc[2] = {p->wit, p->wit2};
    const uint64_t ms[2] = {np - 2, np - 2};
    PTRY(msm_group(p, sc, ms, 2, 9));
  }
  PTRY(fetch_commitments(p, 9, 2, comm + 9));
  if (!quotient_identity_holds(num_at_z, ev, alpha, beta, gamma, l1_z, vp)) return PLONK_ERR_UNSAT;
  write_proof(proof, comm, ev);
  return PLONK_OK;
}

}  // namespace plonk

using namespace plonk;

struct plonk_prover {
  plonk::Prover* p;
  plonk_ctx* ctx;
};

extern "C" {

int plonk_prover_create(plonk_ctx* ctx, const plonk_prover_desc* desc, plonk_prover** out) {
  const char* const api_fn = __func__;
  return plonk::api_guard(api_fn, [&]() -> int {
  if (!ctx || !desc || !out) return (plonk::set_last_error("invalid argument", api_fn, __FILE__, __LINE__), PLONK_ERR_ARG);
  CTX_ENTER(ctx->c, api_fn);
  HIP_TRY(hipSetDevice(ctx->c.device));
  if (!ctx->c.srs_table && desc->shard_world <= 1) return PLONK_ERR_NO_SRS;
  plonk::Prover* p = nullptr;
  int rc = prover_build(&ctx->c, desc, nullptr, &p);
  if (rc) return rc;
  plonk_prover* h = new (std::nothrow) plonk_prover{p, ctx};
  if (!h) { prover_free(p); return (plonk::set_last_error(api_fn, "out of host memory", __FILE__, __LINE__), PLONK_ERR_NOMEM); }
  *out = h;
  return PLONK_OK;
  });
}

int plonk_compile(plonk_ctx* ctx, const plonk_circuit_desc* circuit, plonk_prover** out) {
  const char* const api_fn = __func__;
  return plonk::api_guard(api_fn, [&]() -> int {
  if (!ctx || !circuit || !out) return (plonk::set_last_error("invalid argument", api_fn, __FILE__, __LINE__), PLONK_ERR_ARG);
  CTX_ENTER(ctx->c, api_fn);
  HIP_TRY(hipSetDevice(ctx->c.device));
  if (!ctx->c.srs_table && circuit->shard_world <= 1) return PLONK_ERR_NO_SRS;
  plonk_prover_desc d{};
  d.constraints = circuit->constraints;
  d.label = circuit->label;
  d.label_len = circuit->label_len;
  d.shard_rank = circuit->shard_rank;
  d.shard_world = circuit->shard_world;
  d.srs_total = circuit->srs_total;
  d.allgather = circuit->allgather;
  d.allgather_user = circuit->allgather_user;
  d.lagrange_xy96 = circuit->lagrange_xy96;
  d.lagrange_count = circuit->lagrange_count;
  CircuitSrc src;
  for (int k = 0; k < QS_COUNT; ++k) src.selectors[k] = (const Fr*)circuit->selectors[k];
  for (int w = 0; w < 4; ++w) src.wires[w] = circuit->wires[w];
  src.witnesses = circuit->witnesses;
  plonk::Prover* p = nullptr;
  int rc = prover_build(&ctx->c, &d, &src, &p);
  if (rc) return rc;
  plonk_prover* h = new (std::nothrow) plonk_prover{p, ctx};
  if (!h) { prover_free(p); return (plonk::set_last_error(api_fn, "out of host memory", __FILE__, __LINE__), PLONK_ERR_NOMEM); }
  *out = h;
  return PLONK_OK;
  });
}

void plonk_prover_destroy(plonk_prover* pr) {
  if (!pr) return;
  {
    std::lock_guard<std::mutex> lk(pr->ctx->c.mu);
    (void)hipSetDevice(pr->ctx->c.device);
    if (ctx_abandon(&pr->ctx->c)) {   // poisoned context, streams still busy: hipFree / hipStreamSynchronize would hang — leak the device buffers
      delete pr->p;
    } else {
      (void)hipStreamSynchronize(pr->ctx->c.stream);
      prover_free(pr->p);
    }
  }
  delete pr;
}

int plonk_prover_set_version(plonk_prover* pr, int version) {
  const char* const api_fn = __func__;
  return plonk::api_guard(api_fn, [&]() -> int {
  if (!pr || !pr->p) return PLONK_ERR_ARG;
  if (version != 2 && version != 3) return (plonk::set_last_error("invalid argument", "PlonkVersion: 2 (legacy) or 3; V1 is Error::UnsupportedProvingVersion in the reference too", __FILE__, __LINE__), PLONK_ERR_ARG);
  std::lock_guard<std::mutex> lk(pr->ctx->c.mu);
  pr->p->transcript_version = version;
  return PLONK_OK;
  });
}

int plonk_prover_vk(plonk_prover* pr, uint8_t out[15 * 48]) {
  const char* const api_fn = __func__;
  return plonk::api_guard(api_fn, [&]() -> int {
  if (!pr || !out) return (plonk::set_last_error("invalid argument", api_fn, __FILE__, __LINE__), PLONK_ERR_ARG);
  memcpy(out, pr->p->vk, 15 * 48);
  return PLONK_OK;
  });
}

uint64_t plonk_prover_size(plonk_prover* pr) { return pr ? pr->p->n : 0; }

int plonk_prover_describe(plonk_prover* pr, plonk_prover_info* out) {
  const char* const api_fn = __func__;
  return plonk::api_guard(api_fn, [&]() -> int {
  if (!pr || !out) return PLONK_ERR_ARG;
  std::lock_guard<std::mutex> lk(pr->ctx->c.mu);
  const plonk::Prover* p = pr->p;
  memset(out, 0, sizeof(*out));
  out->size = p->n;
  out->quotient_domain = p->qf;
  out->wire_commit_values = p->lag_table ? 1u : 0u;
  out->lagrange_table_rows = p->lag_table ? p->lag_rows : 0u;
  out->lagrange_points = p->lag_table ? p->lag_n : 0u;
  out->shard_world = (uint32_t)(p->world > 1 ? p->world : 1);
  out->shard_rank = (uint32_t)p->rank;
  out->sharded_quotient = p->sharded ? 1u : 0u;
  out->quotient_classes = p->sharded ? p->Q : 0u;
  out->wire_group_launches = p->last_wire_launches;
  return PLONK_OK;
  });
}

// Test/diagnostic hook: copy `count` Fr starting at `offset` of an internal device array.
int plonk_prover_peek(plonk_prover* pr, int which, uint64_t offset, uint64_t count, uint64_t* out) {
  const char* const api_fn = __func__;
  return plonk::api_guard(api_fn, [&]() -> int {
  if (!pr || !out) return (plonk::set_last_error("invalid argument", api_fn, __FILE__, __LINE__), PLONK_ERR_ARG);
  plonk::Prover* p = pr->p;
  CTX_ENTER(pr->ctx->c, api_fn);
  const Fr* base[] = {p->wpoly, p->zpoly, p->pipoly, p->cos, p->tbuf, p->tparts, p->agg, p->wit,
                      p->evals8, p->sigma_n, p->scratch, p->evout, p->polys};
  const uint64_t cap[] = {4 * p->np, p->np, p->np, 6 * p->qn, p->sharded ? p->np : p->n8, 3 * p->np, p->np, p->np,
                          (uint64_t)(P_COUNT + 2) * p->qn, 4 * p->n, 2 * p->np, 16, (uint64_t)P_COUNT * p->np};
  if (which < 0 || which >= (int)(sizeof(base) / sizeof(base[0])) || offset > cap[which] || count > cap[which] - offset) return (plonk::set_last_error("invalid argument", api_fn, __FILE__, __LINE__), PLONK_ERR_ARG);
  HIP_TRY(hipSetDevice(pr->ctx->c.device));
  HIP_TRY(hipMemcpyAsync(out, base[which] + offset, sizeof(Fr) * count, hipMemcpyDeviceToHost, p->c->stream));
  HIP_TRY(hipStreamSynchronize(p->c->stream));
  return PLONK_OK;
  });
}

int plonk_prover_prove_dev(plonk_prover* pr, const void* wires_dev, const uint64_t* pi_idx, const uint64_t* pi_val,
                           uint64_t pi_count, const uint64_t* blinders, uint8_t proof[1008]) {
  const char* const api_fn = __func__;
  return plonk::api_guard(api_fn, [&]() -> int {
  if (!pr || !wires_dev || !blinders || !proof || (pi_count && (!pi_idx || !pi_val))) return (plonk::set_last_error("invalid argument", api_fn, __FILE__, __LINE__), PLONK_ERR_ARG);
  CTX_ENTER(pr->ctx->c, api_fn);
  HIP_TRY(hipSetDevice(pr->ctx->c.device));
  return prover_prove(pr->p, (const Fr*)wires_dev, pi_idx, (const Fr*)pi_val, pi_count, (const Fr*)blinders, proof);
  });
}

// ---- Prover::to_bytes / Verifier::to_bytes ------------------------------------------------------------------------
namespace {
inline void put_le64(uint8_t* p, uint64_t v) { for (int i = 0; i < 8; ++i) p[i] = (uint8_t)(v >> (8 * i)); }
inline void put_be64(uint8_t* p, uint64_t v) { for (int i = 0; i < 8; ++i) p[i] = (uint8_t)(v >> (8 * (7 - i))); }
inline void put_scalar(uint8_t* p, const Fr& x) {   // BlsScalar::to_bytes: canonical, little-endian
  const Fr c = x.from_mont();
  memcpy(p, c.l, 32);
}
// ProverKey::to_var_bytes / VerifierKey::to_bytes order (widget.rs:347-447, :84-111) against PolyId order:
// q_range and q_logic trade places (the permutation is its own inverse)
constexpr int SER_ORDER[15] = {P_QM, P_QL, P_QR, P_QO, P_QF, P_QC, P_QARITH, P_QLOGIC, P_QRANGE, P_QFIXED, P_QVAR, P_S1, P_S2, P_S3, P_S4};
constexpr uint64_t SER_DOMAIN = 8 + 4 + 5 * 32, SER_VK = 20 * 48 + 8, SER_RAW_POINT = 97;
// EvaluationDomain::to_bytes (domain.rs:59-79) of the domain of size 2^log
void put_domain(uint8_t* p, uint32_t log) {
  const uint64_t size = 1ull << log;
  const Fr gen = omega_of(log), size_fe = Fr::from_u64(size);
  put_le64(p, size);
  for (int i = 0; i < 4; ++i) p[8 + i] = (uint8_t)(log >> (8 * i));
  const Fr vals[5] = {size_fe, size_fe.inv(), gen, gen.inv(), fr_generator().inv()};
  for (int k = 0; k < 5; ++k) put_scalar(p + 12 + 32 * k, vals[k]);
}
void put_vk(uint8_t* p, const Prover* pv) {   // VerifierKey::to_bytes: u64 n, 15 commitments, zero padding to 968 bytes
  memset(p, 0, SER_VK);
  put_le64(p, pv->constraints);
  for (int j = 0; j < 15; ++j) memcpy(p + 8 + 48 * j, pv->vk[SER_ORDER[j]], 48);
}
}  // namespace

int plonk_prover_to_bytes(plonk_prover* pr, uint8_t* out, uint64_t cap, uint64_t* len) {
  const char* const api_fn = __func__;
  return plonk::api_guard(api_fn, [&]() -> int {
  if (!pr || !len) return (plonk::set_last_error("invalid argument", api_fn, __FILE__, __LINE__), PLONK_ERR_ARG);
  CTX_ENTER(pr->ctx->c, api_fn);
  HIP_TRY(hipSetDevice(pr->ctx->c.device));
  plonk::Prover* p = pr->p;
  Ctx* c = p->c;
  if (p->world > 1) return (plonk::set_last_error("a sharded prover holds one slice of the commit key; serialise where the whole key is", api_fn, __FILE__, __LINE__), PLONK_ERR_STATE);
  if (p->srs_gen != c->srs_gen) return (plonk::set_last_error("prover is bound to an SRS that was replaced on its context", api_fn, __FILE__, __LINE__), PLONK_ERR_STATE);
  const uint64_t n = p->n, n8 = 8 * n, np = p->np;
  const uint32_t L = p->logn;
  const uint64_t eval_size = SER_DOMAIN + 32 * n8;
  uint64_t pk_len = 16 + 17 * eval_size;
  for (int k = 0; k < P_COUNT; ++k) pk_len += 8 + 32 * p->poly_len[k];
  const uint64_t ck_len = 8 + SER_RAW_POINT * c->srs_n;
  const uint64_t total = 48 + p->label.size() + pk_len + ck_len + SER_VK;
  *len = total;
  if (!out) return PLONK_OK;                      // size query
  if (cap < total) return (plonk::set_last_error("invalid argument", "output buffer too small", __FILE__, __LINE__), PLONK_ERR_ARG);
  uint8_t* w = out;
  // six big-endian u64 (prover.rs:247-252), then the four sections
  const uint64_t head[6] = {p->label.size(), pk_len, ck_len, SER_VK, n, p->constraints};
  for (int k = 0; k < 6; ++k) { put_be64(w, head[k]); w += 8; }
  memcpy(w, p->label.data(), p->label.size());
  w += p->label.size();
  // ---- ProverKey::to_var_bytes: n, evaluation size, 15 x (length, coefficients, 8n coset evaluations), linear, v_h
  put_le64(w, n); put_le64(w + 8, eval_size);
  w += 16;
  DevFree ev, tmp;
  HIP_TRY(hipMalloc(&ev.p, sizeof(Fr) * n8));
  HIP_TRY(hipMalloc(&tmp.p, sizeof(Fr) * n8));
  uint8_t dom[SER_DOMAIN];
  put_domain(dom, L + 3);
  auto put_evals = [&](const Fr* poly, uint64_t plen) -> int {   // Evaluations::to_var_bytes of coset_fft(poly) on 8n
    memcpy(w, dom, SER_DOMAIN);
    w += SER_DOMAIN;
    if (plen) {
      PTRY(ntt_device(c, poly, (Fr*)ev.p, (Fr*)tmp.p, L + 3, false, true, plen));
      PTRY(poly_from_mont(c, (const Fr*)ev.p, (Fr*)ev.p, n8));
      HIP_TRY(hipMemcpyAsync(w, ev.p, 32 * n8, hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(hipStreamSynchronize(c->stream));
    } else {
      memset(w, 0, 32 * n8);
    }
    w += 32 * n8;
    return PLONK_OK;
  };
  for (int j = 0; j < P_COUNT; ++j) {
    const int k = SER_ORDER[j];
    const uint64_t plen = p->poly_len[k];
    put_le64(w, plen);
    w += 8;
    if (plen) {   // Polynomial::to_var_bytes: the degree + 1 coefficients
      PTRY(poly_from_mont(c, p->polys + k * np, (Fr*)tmp.p, plen));
      HIP_TRY(hipMemcpyAsync(w, tmp.p, 32 * plen, hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(hipStreamSynchronize(c->stream));
      w += 32 * plen;
    }
    PTRY(put_evals(p->polys + k * np, plen));
  }
  {   // permutation.linear_evaluations: X over the coset (compiler.rs:379-382)
    const Fr lin[2] = {Fr::zero(), Fr::one()};
    HIP_TRY(hipMemcpyAsync(p->scratch, lin, sizeof lin, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    PTRY(put_evals(p->scratch, 2));
  }
  {   // v_h_coset_8n: X^n - 1 over the coset takes 8 values (domain.rs:338-351)
    memcpy(w, dom, SER_DOMAIN);
    w += SER_DOMAIN;
    uint8_t vh[8][32];
    Fr point = fr_generator().pow_u64(n);
    const Fr step = omega_of(L + 3).pow_u64(n);
    for (int i = 0; i < 8; ++i) { put_scalar(vh[i], point - Fr::one()); point = point * step; }
    for (uint64_t i = 0; i < n8; ++i) memcpy(w + 32 * i, vh[i & 7], 32);
    w += 32 * n8;
  }
  // ---- CommitKey::to_raw_var_bytes (key.rs:215-229): count, then x || y || infinity flag per point
  {
    DevFree pts;
    HIP_TRY(hipMalloc(&pts.p, sizeof(G1Affine) * (c->srs_n ? c->srs_n : 1)));
    PTRY(srs_export_device(c, (G1Affine*)pts.p));
    std::vector<uint8_t> xy(96 * (size_t)c->srs_n);
    HIP_TRY(hipMemcpyAsync(xy.data(), pts.p, xy.size(), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    put_le64(w, c->srs_n);
    w += 8;
    for (uint64_t i = 0; i < c->srs_n; ++i) {
      memcpy(w, &xy[96 * i], 96);
      w[96] = 0;
      w += SER_RAW_POINT;
    }
  }
  put_vk(w, p);
  w += SER_VK;
  if ((uint64_t)(w - out) != total) return (plonk::set_last_error("plonk_prover_to_bytes", "length accounting", __FILE__, __LINE__), PLONK_ERR_STATE);
  return PLONK_OK;
  });
}

int plonk_verifier_to_bytes(plonk_prover* pr, const uint8_t* opening_key, uint64_t opening_key_len, const uint64_t* pi_idx,
                            uint64_t pi_count, uint8_t* out, uint64_t cap, uint64_t* len) {
  const char* const api_fn = __func__;
  return plonk::api_guard(api_fn, [&]() -> int {
  if (!pr || !len || (opening_key_len && !opening_key) || (pi_count && !pi_idx)) return (plonk::set_last_error("invalid argument", api_fn, __FILE__, __LINE__), PLONK_ERR_ARG);
  plonk::Prover* p = pr->p;
  const uint64_t total = 48 + p->label.size() + SER_VK + opening_key_len + 8 * pi_count;
  *len = total;
  if (!out) return PLONK_OK;
  if (cap < total) return (plonk::set_last_error("invalid argument", "output buffer too small", __FILE__, __LINE__), PLONK_ERR_ARG);
  uint8_t* w = out;
  // Verifier::to_bytes (verifier.rs:88-117): six big-endian u64, label, VerifierKey, OpeningKey, big-endian indexes
  const uint64_t head[6] = {p->label.size(), SER_VK, opening_key_len, pi_count, p->n, p->constraints};
  for (int k = 0; k < 6; ++k) { put_be64(w, head[k]); w += 8; }
  memcpy(w, p->label.data(), p->label.size());
  w += p->label.size();
  put_vk(w, p);
  w += SER_VK;
  if (opening_key_len) memcpy(w, opening_key, opening_key_len);
  w += opening_key_len;
  for (uint64_t i = 0; i < pi_count; ++i) { put_be64(w, pi_idx[i]); w += 8; }
  return PLONK_OK;
  });
}

int plonk_prover_prove_witnesses(plonk_prover* pr, const uint64_t* witnesses, uint64_t count, const uint64_t* pi_idx,
                                 const uint64_t* pi_val, uint64_t pi_count, const uint64_t* blinders, uint8_t proof[1008]) {
  const char* const api_fn = __func__;
  return plonk::api_guard(api_fn, [&]() -> int {
  if (!pr || !blinders || !proof || (count && !witnesses) || (pi_count && (!pi_idx || !pi_val))) return (plonk::set_last_error("invalid argument", api_fn, __FILE__, __LINE__), PLONK_ERR_ARG);
  CTX_ENTER(pr->ctx->c, api_fn);
  HIP_TRY(hipSetDevice(pr->ctx->c.device));
  plonk::Prover* p = pr->p;
  Ctx* c = p->c;
  if (!p->wire_idx) return (plonk::set_last_error("prover was not built by plonk_compile: it has no wire -> witness table", api_fn, __FILE__, __LINE__), PLONK_ERR_STATE);
  if (count != p->witnesses) return (plonk::set_last_error("invalid argument", "witness count differs from the compiled circuit's", __FILE__, __LINE__), PLONK_ERR_ARG);
  // a_scalars .. d_scalars of prove_inner (prover.rs:446-460), gathered in HBM
  if (count) HIP_TRY(hipMemcpyAsync(p->wit_vals, witnesses, sizeof(Fr) * count, hipMemcpyHostToDevice, c->stream));
  PTRY(poly_gather_wires(c, p->wire_idx, p->wit_vals, p->wires, p->constraints, p->n));
  return prover_prove(p, p->wires, pi_idx, (const Fr*)pi_val, pi_count, (const Fr*)blinders, proof);
  });
}

int plonk_prover_prove(plonk_prover* pr, const uint64_t* const wires[4], const uint64_t* pi_idx, const uint64_t* pi_val,
                       uint64_t pi_count, const uint64_t* blinders, uint8_t proof[1008]) {
  const char* const api_fn = __func__;
  return plonk::api_guard(api_fn, [&]() -> int {
  if (!pr || !wires || !blinders || !proof || (pi_count && (!pi_idx || !pi_val))) return (plonk::set_last_error("invalid argument", api_fn, __FILE__, __LINE__), PLONK_ERR_ARG);
  CTX_ENTER(pr->ctx->c, api_fn);
  HIP_TRY(hipSetDevice(pr->ctx->c.device));
  plonk::Prover* p = pr->p;
  Ctx* c = p->c;
  for (int k = 0; k < 4; ++k) if (!wires[k]) return (plonk::set_last_error("invalid argument", api_fn, __FILE__, __LINE__), PLONK_ERR_ARG);
  // the four columns travel on the copy stream while round 1 already transforms the ones that have arrived
  if (!c->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
  HIP_TRY(hipStreamSynchronize(c->stream));   // the previous proof has finished reading p->wires
  for (int k = 0; k < 4; ++k) {
    HIP_TRY(hipMemcpyAsync(p->wires + k * p->n, wires[k], sizeof(Fr) * p->n, hipMemcpyHostToDevice, c->copy_stream));
    HIP_TRY(hipEventRecord(p->ev_wire[k], c->copy_stream));
  }
  p->wires_pending = true;
  const int rc = prover_prove(p, p->wires, pi_idx, (const Fr*)pi_val, pi_count, (const Fr*)blinders, proof);
  p->wires_pending = false;
  (void)hipStreamSynchronize(c->copy_stream);
  return rc;
  });
}

}  // extern "C"
