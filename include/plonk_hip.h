/* libplonk_hip.so — C-ABI of the MI355X (gfx950) backend for dusk-plonk's prover hot path.
 *
 * The reference has no FFI seam; the path sits behind two crate-private leaf
 * call families.  Each entry point below names the reference
 * interface it replaces (paths relative to the dusk-network/plonk tree):
 *
 *   plonk_ntt / plonk_ntt_batch   bodies of EvaluationDomain::{fft,ifft,coset_fft,
 *                                 coset_ifft}_in_place, src/fft/domain.rs:173-232
 *                                 (i.e. best_fft :383-422 + n^-1 scale :195 +
 *                                 distribute_powers :198-204)
 *   plonk_srs_load                CommitKey { powers_of_g } upload,
 *                                 src/commitment_scheme/kzg10/key.rs:37-41
 *   plonk_msm / plonk_msm_batch   the msm_variable_base call inside
 *                                 CommitKey::commit, key.rs:376-388, and the 4-way
 *                                 rayon::join fan-out of Prover::commit_polynomials,
 *                                 src/compiler/prover.rs:187-210
 *
 * Data conventions (bit-identical to the reference's in-memory types):
 *   Fr  = BlsScalar.0 : 4 x uint64 little-endian limbs, Montgomery form, R = 2^256.
 *   G1 base           : 96 bytes = x || y, each 6 x uint64 LE limbs, Montgomery,
 *                       R = 2^384 (first 96 bytes of G1Affine::to_raw_bytes,
 *                       key.rs:215-229).  An SRS never contains the identity.
 *   MSM result        : 97 bytes = x || y (as above) || infinity flag (0/1); the Rust
 *                       shim rebuilds G1Affine / Commitment from it.
 *
 * Ownership: the caller owns every host buffer; nothing is retained after return
 * except the SRS copy made by plonk_srs_load.  The context owns all device memory.
 * Errors: 0 = ok, < 0 = error code below; nothing throws or aborts across the ABI (every
 * entry point catches C++ exceptions: PLONK_ERR_NOMEM for std::bad_alloc, else PLONK_ERR_STATE).
 * The shim maps a non-zero NTT code to a panic (the reference asserts,
 * domain.rs:394,449) and PLONK_ERR_DEGREE to Error::PolynomialDegreeTooLarge
 * (key.rs:362-370).  Threading: every entry point takes a per-context mutex, so
 * rayon workers may call concurrently (prover.rs:174-177,194-197).
 */
#ifndef PLONK_HIP_H
#define PLONK_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct plonk_ctx plonk_ctx;

enum {
  PLONK_OK = 0,
  PLONK_ERR_ARG = -1,     /* null pointer, log_n out of range, length mismatch          */
  PLONK_ERR_HIP = -2,     /* a HIP runtime call failed; see plonk_last_error()          */
  PLONK_ERR_DEGREE = -3,  /* more scalars than SRS points (PolynomialDegreeTooLarge)    */
  PLONK_ERR_NO_SRS = -4,  /* plonk_msm before plonk_srs_load                            */
  PLONK_ERR_NO_GPU = -5,  /* no gfx950 device visible                                   */
  PLONK_ERR_UNSAT = -6,   /* prover: quotient degree check failed (CircuitUnsatisfied)  */
  PLONK_ERR_STATE = -7,   /* prover: called out of order / missing key                  */
  PLONK_ERR_BYTES = -8,   /* serialized input too short (Error::NotEnoughBytes)         */
  PLONK_ERR_DATA = -9,    /* serialized input malformed (dusk_bytes::Error::InvalidData) */
  PLONK_ERR_POINT = -10,  /* commit-key point off curve / not in the subgroup (Error::PointMalformed) */
  PLONK_ERR_NOMEM = -11   /* a host allocation failed inside the library (std::bad_alloc caught at the ABI)  */
};

/* `devices`: HIP device ordinals; ndev must be 1 (one process per GPU — multi-GPU runs
 * use one context per rank).  devices == NULL -> device 0. */
int plonk_ctx_create(plonk_ctx** out, const int* devices, int ndev);
/* Frees every device buffer, stream and event the context owns.  Provers built on the context hold a pointer to it:
 * destroy them FIRST (plonk_prover_destroy locks and synchronises the context it was created on). */
void plonk_ctx_destroy(plonk_ctx* ctx);
const char* plonk_last_error(void);

/* ---- configuration ("a small plonk_gpu_config struct passed at context creation") ----------------
 * Everything that changes what a context BUILDS or RUNS is a field here; a zeroed struct with struct_size set means "all
 * defaults", and plonk_ctx_create(out, devices, ndev) == plonk_ctx_create_ex(out, device, NULL).  The environment names of
 * earlier rounds (PLONK_MSM_TABLE, PLONK_MSM_BUCKETS, PLONK_QUOTIENT_DOMAIN, PLONK_WIRE_COMMIT, PLONK_SHARD_QUOTIENT /
 * _Z / _SIDE, PLONK_NTT_ELOG, PLONK_COMM_TIMEOUT_MS, PLONK_TABLE_BUDGET_MB, PLONK_SIDE_CUS and the kernel-tuning names
 * listed in DESIGN.md section 2) remain as OVERRIDES FOR A/B RUNS ONLY: they are read once, when a context is created or
 * reconfigured, on top of the struct — never latched inside the library at first use.
 *
 * table_budget_bytes: HBM that ALL precomputed point tables of this context may take together (its commit key and the
 *   Lagrange-basis keys of the provers built on it).  0 = 80 % of the device's TOTAL memory.  The layout of a key is the
 *   densest that fits: the commit key takes a row per bit position (32 KiB per point) when that is at most 60 % of the
 *   budget, a row per second bit position (16 KiB) when at most 30 %, else the 16 window rows (2 KiB); a prover's
 *   Lagrange-basis key, built last, takes the densest layout that fits in what is left.  The rule reads the budget and what
 *   THIS context already holds — not the memory that happens to be free — so two contexts given the same budget choose the
 *   same layouts whatever their neighbours do (until round 4: "half of what hipMemGetInfo reports free right now").
 *   Keys of at most 2^18 + 64 points always take window rows (measured faster), table_mode forces a layout.
 * plonk_ctx_get_config returns the EFFECTIVE values (defaults and overrides resolved); plonk_ctx_set_config replaces
 *   them for the key loads, provers and MSMs that follow (e.g. a key that is loaded only to derive another one from it:
 *   table_mode = PLONK_TABLE_WINDOW for that load).  Existing tables and provers keep what they were built with. */
enum { PLONK_TABLE_AUTO = 0, PLONK_TABLE_WINDOW = 16, PLONK_TABLE_HALFPOS = 128, PLONK_TABLE_BITPOS = 256 };
typedef struct plonk_gpu_config {
  uint32_t struct_size;          /* sizeof(plonk_gpu_config) of the caller (shorter = older header: missing fields default) */
  uint32_t reserved;             /* 0 */
  uint64_t table_budget_bytes;   /* 0 = 80 % of the device's total memory */
  int32_t table_mode;            /* PLONK_TABLE_AUTO | _WINDOW | _HALFPOS | _BITPOS */
  int32_t msm_bucket_bits;       /* 0 = by the number of terms (2^19 buckets above 2^18 terms over bit-position rows), 15, 19 */
  int32_t quotient_domain;       /* 0 = 4: quotient on the 4n coset + de-aliasing; 8: the reference's 8n evaluation */
  int32_t wire_commit;           /* 0 = from the wire VALUES over the Lagrange-basis key; 1 = coefficient form like the reference */
  int32_t shard_quotient;        /* multi-GPU: 0 = default (by residue class for world 2 / 4 / 8), 1 = on, -1 = off (MSMs only) */
  int32_t shard_grand_product;   /* multi-GPU: 0 = default (from 2^19 gates and 4 ranks), 1 = on, -1 = off (replicated) */
  int32_t shard_side_stream;     /* multi-GPU: 0 / 1 = replicated transforms on the side stream under the commitments, -1 = off */
  int32_t ntt_elements_log2;     /* 0 = default (2: four elements per lane; side-stream transforms under a busy MSM: 3), 2, 3 */
  int32_t comm_timeout_ms;       /* 0 = 120000: how long a wait behind a collective polls before the communicator is aborted */
  int32_t side_stream_cus;       /* 0 = none; k > 0: the side stream (challenge-independent transforms under the commitments) is confined
                                    to k compute units by a CU mask and uses the four-wave NTT kernels there; the main stream keeps all.
                                    The masked stream (hipExtStreamCreateWithCUMask) has DEFAULT priority and blocking semantics
                                    against the NULL stream — not the low priority / non-blocking flags of the stream it replaces */
} plonk_gpu_config;
int plonk_ctx_create_ex(plonk_ctx** out, int device, const plonk_gpu_config* config /* NULL = defaults */);
int plonk_ctx_get_config(plonk_ctx* ctx, plonk_gpu_config* out /* struct_size set by the caller */);
int plonk_ctx_set_config(plonk_ctx* ctx, const plonk_gpu_config* config);

/* What an MSM of `count` scalar sets of at most m terms WOULD run as on this context right now — over the commit key
 * (table_rows = 0) or over a key of table_rows rows / table_points points (a prover's Lagrange-basis key); bit_sum_tail = 1:
 * as plonk_msm, plonk_msm_batch and every commitment group of prove() run (the host finishes the bit sums: the only tail the
 * 2^19-bucket kernels have), 0: as plonk_msm_dev (final sum on the device, 2^15 buckets) — and what the LAST one did run as
 * (plonk_ctx_last_msm).  The variant tests assert these, so a switch that is silently
 * ignored fails a test. */
enum { PLONK_PLAN_TAIL_SERIAL = 1, PLONK_PLAN_BUCKET_SUM_LANE = 2, PLONK_PLAN_ACCUMULATE_LDS = 4, PLONK_PLAN_SORT13 = 8 };
typedef struct plonk_msm_plan {
  uint32_t table_rows;           /* 16 window rows / 128 / 256 bit-position rows of the key */
  uint32_t bucket_bits;          /* 15 or 19 (17: opt-in A/B build) */
  uint32_t digit_width;          /* bits of a signed digit: 16 (windows), or bucket_bits + 2 (NAF over bit positions; + 1 for half density) */
  uint32_t slice_entries;        /* entries a lane accumulates serially */
  uint32_t ordered_lanes;        /* 1: msm_accumulate_ordered_kernel (lanes in order of slice length) */
  uint32_t wide_words;           /* 1: 64-bit sort words (rows x points above 2^27) */
  uint32_t flags;                /* PLONK_PLAN_*: which opt-in kernel variants of the A/B switches are in effect */
  uint32_t reserved;
  uint64_t terms;                /* m the plan was made for */
  char accumulate_kernel[64];    /* e.g. "nbl::msm_accumulate_ordered_kernel" */
} plonk_msm_plan;
int plonk_ctx_describe_msm(plonk_ctx* ctx, uint64_t m, int count, int bit_sum_tail, uint32_t table_rows, uint64_t table_points,
                           plonk_msm_plan* out);
int plonk_ctx_last_msm(plonk_ctx* ctx, plonk_msm_plan* out);
/* bytes of precomputed tables the context holds right now (commit key + Lagrange-basis keys) and its budget */
int plonk_ctx_table_bytes(plonk_ctx* ctx, uint64_t* in_use, uint64_t* budget);

/* In-place transform of a[0 .. 1<<log_n) (host memory).
 *   inverse = 0: forward with w;  1: inverse with w^-1 and the n^-1 scale.
 *   coset   = 1: forward pre-scales coefficient i by 7^i for i < in_len;
 *                inverse post-scales output i by 7^-i.
 *   in_len  : number of valid leading coefficients; the rest are treated as zero
 *             (the zero-padding of Vec::resize, domain.rs:174).  Use 1<<log_n for
 *             inverse transforms. */
int plonk_ntt(plonk_ctx* ctx, uint64_t* a, uint32_t log_n, int inverse, int coset, uint64_t in_len);
int plonk_ntt_batch(plonk_ctx* ctx, uint64_t* const* a, int count, uint32_t log_n, int inverse,
                    int coset, const uint64_t* in_len);

/* Load the commit key (npoints x 96 B, npoints <= 2^23) and build the per-window tables 2^(16 w) * P_i in HBM
 * (16 x 128 B per point).  The key is STREAMED from host memory in 2^18-point chunks through two device
 * staging buffers, the upload of chunk k + 1 overlapping the table build of chunk k; pass memory from
 * plonk_host_alloc (pinned) for asynchronous uploads.  A multi-GPU rank passes only its point range. */
int plonk_srs_load(plonk_ctx* ctx, const uint8_t* xy96, uint64_t npoints);
int plonk_host_alloc(uint64_t bytes, void** out);   /* pinned host memory (hipHostMalloc) */
int plonk_host_free(void* p);

/* sum_i scalars[i] * P_i for i < m.  m == 0 -> identity. */
int plonk_msm(plonk_ctx* ctx, const uint64_t* scalars, uint64_t m, uint8_t out_xy_inf[97]);
int plonk_msm_batch(plonk_ctx* ctx, const uint64_t* const* scalars, const uint64_t* m, int count,
                    uint8_t* out /* count x 97 */);

/* ---- device-resident variants (buffers stay in HBM between calls) -------------
 * Pointers are device pointers on the context's GPU (e.g. torch.Tensor.data_ptr()).
 * dst may equal src.  tmp must hold 1<<log_n elements when log_n > 10. */
int plonk_ntt_dev(plonk_ctx* ctx, const void* src, void* dst, void* tmp, uint32_t log_n,
                  int inverse, int coset, uint64_t in_len);
int plonk_msm_dev(plonk_ctx* ctx, const void* scalars, uint64_t m, void* out97_dev);
int plonk_srs_load_dev(plonk_ctx* ctx, const void* xy96_dev, uint64_t npoints);
/* Synthetic "random SRS" [tau^i] G for i < npoints generated on the GPU
 * (PublicParameters::setup semantics, src/commitment_scheme/kzg10/srs.rs:61-100);
 * tau and g_scalar are Fr (Montgomery).  Writes npoints x 96 B to out_dev. */
int plonk_srs_generate_dev(plonk_ctx* ctx, const uint64_t tau[4], const uint64_t g_scalar[4],
                           uint64_t npoints, void* out_dev);
/* The Lagrange-basis form of the context's commit key for the domain of size n = 1 << log_n (the key must hold n + 2
 * points): out[i] = [L_i(tau)] G for i < n, then [tau^n] G - G and [tau^(n+1)] G - [tau] G — (n + 2) x 96 bytes, host
 * memory.  An inverse FFT over the group on the GPU.  A single-GPU prover computes this itself; a multi-GPU run calls
 * it once where the whole key is available and hands every rank its slice (plonk_prover_desc.lagrange_xy96). */
int plonk_lagrange_key(plonk_ctx* ctx, uint32_t log_n, uint8_t* out_xy96);

/* plain device memory helpers so callers need no HIP bindings of their own */
int plonk_dev_alloc(plonk_ctx* ctx, uint64_t bytes, void** out);
int plonk_dev_free(plonk_ctx* ctx, void* p);
int plonk_dev_h2d(plonk_ctx* ctx, void* dst_dev, const void* src_host, uint64_t bytes);
int plonk_dev_d2h(plonk_ctx* ctx, void* dst_host, const void* src_dev, uint64_t bytes);
int plonk_dev_sync(plonk_ctx* ctx);
void* plonk_ctx_stream(plonk_ctx* ctx); /* the hipStream_t of the context's main stream (prove() also uses a private side stream) */
/* Rows of the commit-key tables the context holds: 256 = one row per bit position (2^r * P_i: NAF digits over 2^19 buckets,
 * ~12.1 additions per scalar), 128 = a row for every second position (12.8), 16 = window rows (2^(16 w) * P_i, 16 additions
 * per scalar), 0 = no key loaded.  Chosen from the context's table budget (plonk_gpu_config above) for keys of more than
 * 2^18 + 64 points; same results whichever; plonk_gpu_config.table_mode / .msm_bucket_bits force a layout / bucket count. */
int plonk_ctx_table_rows(plonk_ctx* ctx);

/* ---- device-resident Prover::prove (V3) -------------------------------------------
 * Replaces prove_inner (src/compiler/prover.rs:415-761) minus witness generation and the
 * RNG, which stay with the caller (Composer::prove, src/composer.rs:442; BlsScalar::random).
 * The context must hold the commit key (plonk_srs_load: >= size + 7 points).  A prover is bound to
 * the key its context held at plonk_prover_create: after a later plonk_srs_load /
 * plonk_prover_from_bytes on the same context, plonk_prover_prove on the older prover returns
 * PLONK_ERR_STATE (its degree bounds and shard ranges describe a key that is gone) — rebuild it.
 *
 * desc.polys: the 15 ProverKey polynomials in coefficient form (Fr Montgomery limbs), order
 *   q_m q_l q_r q_o q_f q_c q_arith q_range q_logic q_fixed_group_add q_variable_group_add
 *   s_sigma_1 s_sigma_2 s_sigma_3 s_sigma_4   (src/proof_system/widget.rs:284-313)
 * desc.vk_commitments: the 15 VerifierKey commitments (48-byte compressed, same order) that
 *   seed the transcript (widget.rs:218-258); NULL = commit to the polynomials on the GPU as
 *   Compiler::preprocess does (src/compiler.rs:213-232).
 * The 8n coset evaluations, sigma evaluations and vanishing inverses (compiler.rs:310-425,
 * prover.rs:78-100) are rebuilt on the device.  On one GPU the prover also derives the Lagrange-basis form of the
 * commit key ([L_i(tau)] G, an inverse FFT over the group — needs size + 2 key points) and takes the four wire
 * commitments from the wire VALUES: the same group elements as CommitKey::commit of the blinded coefficient forms
 * (prover.rs:139-152,187-210), cheaper the smaller the witness values are.  PLONK_WIRE_COMMIT=coeff disables it. */
typedef struct plonk_prover plonk_prover;
/* Multi-GPU (one process per GPU): rank r loads only SRS points [r*S, (r+1)*S), S = ceil(srs_total / world),
 * into its context (srs_total >= size + 7) and owns the same range of polynomial COEFFICIENTS:
 *   - every MSM runs on the rank's point range; the per-rank partial sums (192-byte XYZZ points) are
 *     all-gathered and added locally (EC addition is not an RCCL reduce op);
 *   - world in {2, 4, 8}: the quotient is computed per residue class of the quotient coset (rank r owns
 *     the size-n cosets g w^j H_n with j = r mod world), one all-to-all turns the per-class remainders into
 *     the coefficient range of t the rank commits to, and evaluations / linearisation / opening quotients
 *     work on that range with two more small all-gathers;
 *   - other world sizes (or PLONK_SHARD_QUOTIENT=0): only the MSMs are sharded.
 * Transport: RCCL on the library's stream when the context has a communicator (plonk_comm_init below);
 * otherwise this host callback — an all-gather, `recv` laid out rank-major — which is how the tests run
 * several ranks on one device.  Return 0 on success. */
typedef int (*plonk_allgather_fn)(void* user, const void* send, void* recv, uint64_t bytes_per_rank);
typedef struct {
  uint64_t constraints;          /* gate count; domain size = next power of two        */
  const uint8_t* label;          /* transcript label (Compiler::compile `label`)        */
  uint64_t label_len;
  const uint64_t* polys[15];
  uint64_t poly_len[15];         /* coefficients per polynomial, <= size                */
  const uint8_t* vk_commitments; /* 15 x 48 bytes or NULL                               */
  int shard_rank;                /* 0 when shard_world <= 1                             */
  int shard_world;               /* <= 1: single GPU, the context holds the whole SRS   */
  uint64_t srs_total;            /* global SRS length (only read when shard_world > 1)  */
  plonk_allgather_fn allgather;
  void* allgather_user;
  /* The Lagrange-basis key of plonk_lagrange_key, or this rank's part of it.
   * shard_world <= 1: all size + 2 points (lagrange_count == size + 2) of a key computed earlier for the same
   *   commit key and size — the prover then skips the group FFT (the bulk of its build time); NULL = derive it.
   * shard_world > 1: the slice [shard_rank * S, shard_rank * S + lagrange_count) of those points, S as above;
   *   NULL / 0 = commit to the wire polynomials in coefficient form.
   * A supplied key is validated: every point on the curve and in the prime-order subgroup (PLONK_ERR_POINT) and, on one
   * GPU, the whole key against the context's commit key by a random linear combination — one inverse transform and two
   * MSMs (PLONK_ERR_DATA for the key of another setup, a permuted or otherwise wrong key).  Sharded provers (round 6): every
   * rank must pass the SAME kind — NULL, its slice (a non-NULL pointer even when the slice is empty) or the whole key; one
   * mode byte per rank is all-gathered at creation and a disagreement is PLONK_ERR_ARG on every rank (mixed modes would add
   * whole-column commitments to point-range partial sums: silently wrong wire commitments).  The same random-combination
   * identity then runs over the ranks — slices summed like any sharded commitment, whole keys compared per rank with the
   * verdict shared — so a wrong key on any rank is PLONK_ERR_DATA on all of them. */
  const uint8_t* lagrange_xy96;
  uint64_t lagrange_count;
} plonk_prover_desc;
int plonk_prover_create(plonk_ctx* ctx, const plonk_prover_desc* desc, plonk_prover** out);
void plonk_prover_destroy(plonk_prover* p);
/* Prover::prove_with_version (prover.rs:365-413): version 3 = PlonkVersion::V3, the default of every prover (Transcript::base_v3,
 * transcript.rs:131-145); version 2 = the legacy seeding the reference keeps behind its `legacy-proving` feature
 * (Transcript::base + VerifierKey::seed_transcript_legacy, transcript.rs:110-129, widget.rs:224-228,260-265: the label s_sigma_4
 * carries the commitment of s_sigma_1).  Nothing else of a proof depends on the version.  Other values: PLONK_ERR_ARG (V1 is
 * Error::UnsupportedProvingVersion in the reference).  Applies to the proofs made after the call. */
int plonk_prover_set_version(plonk_prover* p, int version);
int plonk_prover_vk(plonk_prover* p, uint8_t out[15 * 48]);
/* What the prover was BUILT as — the configuration of its context at plonk_prover_create / plonk_compile, resolved: tests assert
 * it so that a configuration switch the library ignores cannot pass for the variant it names. */
typedef struct plonk_prover_info {
  uint64_t size;                 /* domain size n */
  uint32_t quotient_domain;      /* 4: quotient interpolated on the 4n coset and de-aliased; 8: the reference's 8n evaluation */
  uint32_t wire_commit_values;   /* 1: wire commitments from the wire VALUES over a Lagrange-basis key; 0: coefficient form */
  uint32_t lagrange_table_rows;  /* rows of that key's tables (16 / 128 / 256), 0 without one */
  uint32_t shard_world, shard_rank;
  uint32_t sharded_quotient;     /* 1: quotient by residue class, rounds 4-5 by coefficient range; 0: only the MSMs are sharded */
  uint32_t quotient_classes;     /* Q: 4, or 8 for eight ranks (0 when the quotient is not sharded) */
  uint32_t wire_group_launches;  /* of the LAST proof's wire commitments: 1 = one grouped launch (resident columns); 3 = columns a, b, then c + d
                                    as they arrive from the host (plonk_prover_prove from 2^19 gates on); 4 = one launch per column; 0 before the first proof */
  uint64_t lagrange_points;      /* points of the Lagrange-basis key (or of this rank's slice) */
} plonk_prover_info;
int plonk_prover_describe(plonk_prover* p, plonk_prover_info* out);
uint64_t plonk_prover_size(plonk_prover* p);
/* diagnostic: read `count` Fr at `offset` of internal array `which` (0 wire polys, 1 z poly,
 * 2 pi poly, 3 coset evals z|a|b|c|d|pi, 4 quotient, 5 t_low|t_mid|t_high, 6 lin. comb., 7 opening
 * witness, 8 key coset evals, 9 sigma evals, 10 scratch, 11 evaluations, 12 key polynomials) */
int plonk_prover_peek(plonk_prover* p, int which, uint64_t offset, uint64_t count, uint64_t* out);
/* wires: a, b, c, d columns padded to `size` (prover.rs:446-460), Fr Montgomery.
 * pi_idx/pi_val: sparse public inputs (gate row, value), rows ascending (composer.rs:465-485).
 * blinders: 14 Fr in the reference's RNG draw order: a0 a1 b0 b1 c0 c1 d0 d1 (prover.rs:154-161),
 *           z0 z1 z2 (:133-135,503), b12 b13 b14 (:553-555).
 * proof: 1008 bytes = Proof::to_bytes (src/proof_system/proof.rs:137-162).
 * PLONK_ERR_UNSAT mirrors Error::CircuitUnsatisfied (quotient_poly.rs:132).  By default the quotient is
 * interpolated on the 4n coset and de-aliased — the same t(X), hence the same proof
 * bytes — and an unsatisfied witness is recognised by the quotient identity failing at the evaluation
 * challenge (probability of missing it <= 5n/q); PLONK_QUOTIENT_DOMAIN=8 in the environment selects the
 * reference's 8n evaluation with its exact degree test.
 * One deviation: when the evaluation challenge z or z * omega is ZERO (probability 2^-254) the opening quotients are
 * computed with 1 / z, and the call returns PLONK_ERR_STATE where the reference would go on to emit a proof.
 * Cost of the host columns: they cross PCIe inside the call, one after the other on a copy stream; from 2^19 gates on the
 * library commits to column a, then b, then c + d as each lands (plonk_prover_info.wire_group_launches = 3), so only the
 * FIRST column's copy is exposed — a proof from PINNED columns (plonk_host_alloc) is 0.8-1.15 ms slower than
 * plonk_prover_prove_dev at 2^20 gates (bench.py: prove_ms_host_wires_pinned, host_wires.pcie_floor_ms).  Pageable memory
 * works but the copies then stage synchronously and overlap nothing. */
int plonk_prover_prove(plonk_prover* p, const uint64_t* const wires[4], const uint64_t* pi_idx,
                       const uint64_t* pi_val, uint64_t pi_count, const uint64_t* blinders,
                       uint8_t proof[1008]);
/* same with the 4 x size wire columns already resident in HBM (contiguous a|b|c|d) */
int plonk_prover_prove_dev(plonk_prover* p, const void* wires_dev, const uint64_t* pi_idx,
                           const uint64_t* pi_val, uint64_t pi_count, const uint64_t* blinders,
                           uint8_t proof[1008]);

/* ---- Compiler::preprocess on the device -------------------------------------------------
 * plonk_compile replaces Compiler::preprocess (src/compiler.rs:132-461) for a circuit the caller has
 * already laid out as gates: what Composer holds after Circuit::circuit ran (src/composer.rs:119-167 —
 * per gate the 11 selector values and the witness index on each of its four wires).  On the device:
 * the 11 selector interpolations (compiler.rs:177-187), the four sigma polynomials
 * (Permutation::compute_sigma_polynomials, src/composer/permutation.rs:106-211), the 15 VerifierKey
 * commitments (compiler.rs:213-232) and the evaluation arrays of the ProverKey (compiler.rs:310-425; on the
 * quotient domain this library uses, see plonk_prover_create).  The result is a prover like the one
 * plonk_prover_create builds from coefficient forms; plonk_prover_vk returns the commitments.
 *
 * selectors[k]: `constraints` Fr values (Montgomery limbs) of selector k in plonk_prover_desc.polys order
 *   (q_m q_l q_r q_o q_f q_c q_arith q_range q_logic q_fixed_group_add q_variable_group_add); NULL = zero.
 * wires[w]: `constraints` witness indices (Witness::index) on wires a, b, c, d; every index < witnesses.
 * The shard_* / allgather / lagrange_* fields mean what they mean in plonk_prover_desc.
 * Errors: PLONK_ERR_ARG (NULL wires, index out of range, fewer than 2 gates), PLONK_ERR_DEGREE (commit key
 * shorter than the domain), PLONK_ERR_NO_SRS. */
typedef struct {
  uint64_t constraints;
  const uint8_t* label;
  uint64_t label_len;
  const uint64_t* selectors[11];
  const uint32_t* wires[4];
  uint64_t witnesses;            /* number of witnesses the composer allocated                */
  int shard_rank;
  int shard_world;
  uint64_t srs_total;
  plonk_allgather_fn allgather;
  void* allgather_user;
  const uint8_t* lagrange_xy96;
  uint64_t lagrange_count;
} plonk_circuit_desc;
int plonk_compile(plonk_ctx* ctx, const plonk_circuit_desc* circuit, plonk_prover** out);
/* Prove from the witness VALUES (count x Fr, count == circuit.witnesses) on a prover built by plonk_compile:
 * the four wire columns of prove_inner (prover.rs:446-460) are gathered in HBM from the witness table, so
 * only the distinct values cross PCIe.  Everything else as plonk_prover_prove.  PLONK_ERR_STATE on a prover
 * that was not compiled from a circuit. */
int plonk_prover_prove_witnesses(plonk_prover* p, const uint64_t* witnesses, uint64_t count, const uint64_t* pi_idx,
                                 const uint64_t* pi_val, uint64_t pi_count, const uint64_t* blinders,
                                 uint8_t proof[1008]);

/* ---- multi-GPU: one process per GPU, RCCL over xGMI ---------------------------------------
 * Rank 0 calls plonk_comm_unique_id and hands the 128 bytes (an ncclUniqueId) to the other ranks by
 * whatever out-of-band channel the host program has; every rank then calls plonk_comm_init on its
 * context (collective: it returns when all `world` ranks have joined).  A context with a communicator
 * runs every exchange of a sharded prover (plonk_prover_desc.shard_world > 1) as ncclAllGather /
 * ncclAllToAll on its own stream — no host callback is involved and desc.allgather may be NULL.
 * RCCL is loaded with dlopen at the first of these calls, so the library itself does not depend on it.
 * plonk_comm_selftest runs both collectives once with a rank-dependent pattern and checks the result.
 * What is exchanged and why EC partial sums are all-gathered rather than all-reduced: DESIGN.md §5.
 * plonk_comm_info reports the rank / size the communicator itself holds (ncclCommUserRank / ncclCommCount).
 *
 * Environment: this platform's driver only supports dmabuf IPC, so RCCL between processes needs
 * HSA_ENABLE_IPC_MODE_LEGACY=0 in the environment before the HSA runtime starts (the first HIP call of the process).
 * The HOST PROGRAM exports it before its first HIP call (bench.py and the Python binding do); the library does not touch the
 * process environment.  plonk_comm_init names the variable in its error text when the communicator cannot be created
 * without it.
 *
 * Failure of a peer: a sharded proof is a sequence of collectives, and a rank that fails (or never arrives) would
 * leave the others waiting inside one.  Every wait that follows a collective polls the stream instead of blocking; after
 * PLONK_COMM_TIMEOUT_MS (default 120000) the communicator is aborted (ncclCommAbort) and the call returns
 * PLONK_ERR_STATE on every surviving rank — the context then has no communicator and must be given a new one
 * (plonk_comm_init) before the next sharded proof.  The host program should still tear the job down when one rank
 * reports an error (bench.py does: the launcher kills the remaining ranks).
 *
 * Measurement aid, never for production: after plonk_comm_measure_loopback(ctx, 1) every collective of a sharded prover on
 * THIS context returns data of the rank's own in its peers' places (local copies, no transport: the all-to-all the blocks the
 * rank addressed to each peer, the device all-gather rotated copies of its slice — W different blocks, so that what follows
 * them works on scalars as dense as a real run's), so that one rank of a W-rank job can be timed alone on one GPU
 * (tools/rank_alone.py).  Proofs made that way are wrong by construction:
 * plonk_prover_prove* returns PLONK_ERR_UNSAT from its final identity check.  Nothing in the environment switches it on.
 *
 * Transport library: RCCL is looked up as librccl.so(.1) on the loader path, then under /opt/rocm/lib.
 * plonk_comm_set_library(path) names another shared object exporting the nccl* entry points BEFORE the first
 * communicator call of the process (PLONK_ERR_STATE once a library is loaded): a deployment whose RCCL lives elsewhere, or
 * the test suite's stand-in (tests/fake_rccl: device-pointer collectives over hipIpc between ranks that SHARE one GPU —
 * RCCL refuses two ranks per device — so that the W > 1 branch of every exchange runs on a 1-GPU box).
 * plonk_comm_library writes the path of what was actually loaded (NUL-terminated, truncated to cap); a host program that
 * reports which collective a measurement used must take it from here (bench.py prints "rccl" only for a librccl file).
 * After a time-out whose abort did not drain the stream (or a transport without ncclCommAbort) the context is UNUSABLE:
 * every entry point that queues work on the context or waits for it (transforms, commitments, key loads, copies, provers,
 * plonk_comm_init) returns PLONK_ERR_STATE at once instead of queueing behind the dead collective; destroy it.  The destroy
 * calls (plonk_prover_destroy, plonk_comm_destroy, plonk_ctx_destroy) poll the streams for at most 2 s and, if they are still
 * busy, ABANDON the context's device memory, streams and pinned buffers instead of freeing them — hipFree and
 * hipStreamSynchronize would wait for the dead kernel for ever; plonk_comm_destroy then returns PLONK_ERR_STATE.  A host that
 * wants the memory back resets the device (hipDeviceReset) or exits.
 * plonk_comm_warning: the note plonk_comm_init left on this context when it SUCCEEDED with something to say (today: the
 * dmabuf IPC variable missing from the environment), "" otherwise — a successful call never writes plonk_last_error(). */
int plonk_comm_set_library(const char* path);
int plonk_comm_library(char* out, uint64_t cap);
int plonk_comm_measure_loopback(plonk_ctx* ctx, int on);
int plonk_comm_unique_id(uint8_t out[128]);
int plonk_comm_init(plonk_ctx* ctx, const uint8_t unique_id[128], int rank, int world);
int plonk_comm_info(plonk_ctx* ctx, int* rank, int* world);
int plonk_comm_warning(plonk_ctx* ctx, char* out, uint64_t cap);
int plonk_comm_selftest(plonk_ctx* ctx);
int plonk_comm_destroy(plonk_ctx* ctx);

/* ---- serialized Prover ---------------------------------------------------------------
 * plonk_prover_from_bytes replaces Prover::try_from_bytes (src/compiler/prover.rs:266-345): `blob`
 * is exactly what the reference's Prover::to_bytes() writes (prover.rs:238-263 — six big-endian u64
 * lengths, label, ProverKey::to_var_bytes widget.rs:347-447, CommitKey::to_raw_var_bytes
 * key.rs:215-229, VerifierKey::to_bytes widget.rs:84-111).  It validates the blob as the
 * reference does (errors: PLONK_ERR_BYTES / PLONK_ERR_DATA / PLONK_ERR_POINT), loads the commit
 * key into `ctx` (replacing any SRS it held) and builds the device prover.  The 8n evaluation
 * arrays in the blob are validated structurally and then rebuilt on the device from the
 * coefficient forms.
 * plonk_prover_blob_check is the host-only decoder/validator (no GPU, no context): offsets of the
 * pieces inside `blob`; polynomials in the plonk_prover_desc order.  It checks the curve
 * equation of the commit-key points; the subgroup check needs the GPU (plonk_srs_validate).
 * plonk_srs_validate: CommitKey::from_raw_var_bytes' per-point is_on_curve & is_torsion_free
 * (key.rs:283-294) for x||y points as taken by plonk_srs_load. */
/* The other direction.  plonk_prover_to_bytes writes what the reference's Prover::to_bytes() writes
 * (prover.rs:238-263) for this prover and the commit key its context holds — the key polynomials, their 8n coset
 * evaluations (computed on the device and streamed out: 17 x 8 x size scalars, 4.6 GB at 2^20 gates), the raw commit
 * key and the VerifierKey — so a circuit compiled by plonk_compile can be loaded by the unmodified reference
 * (Prover::try_from_bytes) or by plonk_prover_from_bytes later.  plonk_verifier_to_bytes writes Verifier::to_bytes()
 * (src/compiler/verifier.rs:88-117) from the prover's label / sizes / VerifierKey, the caller's OpeningKey::to_bytes()
 * (opaque here: G2 never enters this library) and the public-input indexes.
 * Both: out == NULL reports the length in *len; otherwise cap >= *len bytes are written.  Single-GPU provers only.
 * Known differences from the reference's bytes (format parity is pinned only to the restated layout):
 * plonk_prover_from_bytes keeps the size + 8 commit-key points prove() can touch, so from_bytes -> to_bytes writes a
 * SHORTER commit key than a blob whose key was trimmed to (constraints + 6).next_power_of_two() + 7 points (the proofs
 * are identical; the blob is not byte-identical, and generic plonk_msm calls on that context cannot use the dropped
 * degrees).  No new format entry points are planned until a reference-produced blob pins these (tools/dump_kat_blob.rs). */
int plonk_prover_to_bytes(plonk_prover* p, uint8_t* out, uint64_t cap, uint64_t* len);
int plonk_verifier_to_bytes(plonk_prover* p, const uint8_t* opening_key, uint64_t opening_key_len,
                            const uint64_t* pi_idx, uint64_t pi_count, uint8_t* out, uint64_t cap, uint64_t* len);
typedef struct {
  uint64_t size, constraints;
  uint64_t label_off, label_len;
  uint64_t poly_off[15], poly_len[15]; /* canonical 32-byte little-endian scalars */
  uint64_t srs_off, srs_points;        /* 97-byte raw points (x || y || infinity) */
  uint64_t vk_off;                     /* 15 x 48 bytes, VerifierKey::to_bytes order */
} plonk_prover_blob_info;
int plonk_prover_blob_check(const uint8_t* blob, uint64_t len, plonk_prover_blob_info* info);
int plonk_prover_from_bytes(plonk_ctx* ctx, const uint8_t* blob, uint64_t len, plonk_prover** out);
int plonk_srs_validate(plonk_ctx* ctx, const uint8_t* xy96, uint64_t npoints);

/* ---- PublicParameters on disk -> commit key on the device ------------------------------------
 * The files a dusk-plonk user keeps (src/commitment_scheme/kzg10/srs.rs:103-178).  Both start with
 * OpeningKey::to_bytes() (240 B: g 48 B, h 96 B, x_h 96 B, compressed; key.rs:436-452); then
 *   RAW        PublicParameters::to_raw_var_bytes(): CommitKey::to_raw_var_bytes() = u64 LE count, count x 97 B raw points
 *              x || y || infinity flag (key.rs:215-229);
 *   COMPRESSED PublicParameters::to_var_bytes(): CommitKey::to_var_bytes() = 48-byte compressed points, no count (key.rs:303-308).
 * mode:
 *   PLONK_PP_RAW_UNCHECKED  PublicParameters::from_slice_unchecked (srs.rs:131-146): the commit-key points are trusted; like
 *                           CommitKey::from_slice_unchecked (key.rs:243-258) it takes min(count, whole 97-byte chunks present) points;
 *   PLONK_PP_RAW            CommitKey::from_raw_var_bytes (key.rs:263-300): count != 0 (PLONK_ERR_DATA), exact length
 *                           (PLONK_ERR_BYTES), EVERY point of the file is_on_curve & is_torsion_free (PLONK_ERR_POINT; on the
 *                           GPU) — also the ones a trim drops afterwards, as in the reference;
 *   PLONK_PP_COMPRESSED     PublicParameters::from_slice (srs.rs:164-178) = one G1Affine::from_slice per 48-byte chunk
 *                           (key.rs:319-326): compression flag, x < p, x^3 + 4 a square, torsion-free — any failure, a short
 *                           last chunk included, is a dusk_bytes error (PLONK_ERR_DATA), wherever in the file.  Decompression
 *                           (one square root per point, g1codec.cuh) and the subgroup test run on the GPU.
 *   All modes: OpeningKey::from_bytes + try_new (key.rs:596-648) — g a valid compressed G1 point, h and x_h valid compressed G2
 *   points (flags, canonical coordinates, on the twist curve, of order q; hostg2.hpp, on the host) and NONE of the three the
 *   identity (try_new refuses a degenerate key: the pairing check would be trivially satisfiable) — PLONK_ERR_DATA otherwise.
 *   The prover never uses them; they are handed back as the 240 bytes they came as.
 *   truncated_degree > 0: PublicParameters::trim (srs.rs:188-196) = CommitKey::truncate(truncated_degree + 6)
 *   (key.rs:336-355): PLONK_ERR_DEGREE when the key is shorter (Error::TruncatedDegreeTooLarge); 0 keeps every point.  The
 *   unchecked mode never reads beyond the trim.  At most 240 bytes: PLONK_ERR_BYTES (Error::NotEnoughBytes, srs.rs:165-167).
 *   ONE deliberate divergence: an identity point in the commit key (flag byte 1 / the 0xC0 encoding) is refused with
 *   PLONK_ERR_POINT in every mode, although G1Affine::from_bytes and is_on_curve & is_torsion_free accept it: no SRS
 *   [tau^i] G holds one and the precomputed table rows cannot represent it (tests/test_public_parameters.py names this case).
 * plonk_public_parameters_check is the host-side part (no GPU): structure, opening key, trim, flags and coordinate ranges.
 * plonk_srs_load_public_parameters = check + the per-point work on the GPU + window tables of the kept points. */
enum { PLONK_PP_RAW_UNCHECKED = 0, PLONK_PP_RAW = 1, PLONK_PP_COMPRESSED = 2 };
typedef struct plonk_public_parameters_info {
  uint64_t opening_key_off;   /* 240 bytes */
  uint64_t points_off;        /* first point */
  uint64_t point_stride;      /* 97 (raw) or 48 (compressed) */
  uint64_t points_total;      /* points the file holds */
  uint64_t points_kept;       /* after the trim */
} plonk_public_parameters_info;
int plonk_public_parameters_check(const uint8_t* bytes, uint64_t len, uint64_t truncated_degree, int mode,
                                  plonk_public_parameters_info* info);
int plonk_srs_load_public_parameters(plonk_ctx* ctx, const uint8_t* bytes, uint64_t len, uint64_t truncated_degree,
                                     int mode, uint8_t opening_key_out[240] /* may be NULL */,
                                     uint64_t* points_loaded /* may be NULL */);

/* ---- measurement --------------------------------------------------------------
 * When enabled, every launch of the dominant kernels is bracketed by a hipEvent
 * pair ON THE LIBRARY'S STREAM and accumulated per slot.  Slots:
 *   0 ntt pass kernels, 1 msm bucket accumulation, 2 msm (all other kernels),
 *   3 quotient/pointwise kernels, 4 the polynomial work of prove() rounds 1-2 (wire / permutation
 *   polynomials: what stays replicated on every rank of a multi-GPU run).
 *   With PLONK_PROF_FINE=1 in the environment also, per phase of a commitment group (groups of >= 3 commitments: 16 + phase,
 *   smaller groups: 24 + phase): 0 bucket sort, 1 accumulation, 2 bucket sums, 3 heavy buckets, 4 row / column sums, 5 bit sums.
 *   HOST time of plonk_prover_prove* (wall clock, no events; `launches` = occurrences): 8 the arithmetic
 *   that turns the bit sums of a commitment group into compressed commitments, 9 from the return of each of the five
 *   synchronisations of a proof (six or seven for a rank of a sharded proof: the grand product's and the opening
 *   quotients' range totals are exchanged too) to the next launch (slot 8 included: the device's main stream is idle for that long),
 *   10 the time blocked in those synchronisations; 11 (a count, not a time) the helper threads each commitment group had.
 *   Slots 0 .. 31 are valid.
 *
 * Host threads.  A context starts up to 3 helper threads (none when the process may run on fewer than 8 CPUs: sched_getaffinity) with its first
 * commitment group of more than one commitment: they take the independent finishing chains of a group (slot 8) beside the
 * calling thread.  They are woken when the caller blocks in the group's synchronisation and SPIN until the bit sums have
 * arrived — up to one device phase per group — and sleep otherwise; plonk_ctx_destroy joins them.  PLONK_HOST_THREADS=k in
 * the environment (read at context creation) sets their number, 0 switches them off. */
int plonk_profile_enable(plonk_ctx* ctx, int on);
int plonk_profile_read(plonk_ctx* ctx, int slot, double* total_ms, uint64_t* launches);
int plonk_profile_reset(plonk_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* PLONK_HIP_H */
