// Argument blocks + launchers of poly.hip (device-resident polynomial kernels).
#pragma once
#include "plonk_internal.hpp"

namespace plonk {

enum { QS_M = 0, QS_L, QS_R, QS_O, QS_F, QS_C, QS_ARITH, QS_RANGE, QS_LOGIC, QS_FIXED, QS_VAR, QS_COUNT };

struct BlindArgs {
  int count;
  Fr b[3];
};
struct SplitArgs {
  Fr b[3];          // b12, b13, b14 (prover.rs:553-555)
  uint64_t len4;    // coefficients of t beyond 3n that belong to t_fourth
};
struct PermArgs {
  uint64_t n;
  uint64_t first = 0, count = 0;   // evaluation indices [first, first + count) only (a rank's share of the grand product); count == 0: all n
  const Fr* wires[4];
  const Fr* sigma[4];   // sigma_evaluations over the n-domain (prover.rs:95-100)
  Fr beta, gamma;
  Fr ks[4];             // 1, K1, K2, K3
  const void* tw_lo29;  // w_n^i two-level table in twiddle form (Fr29Slot, forward NTT tables of log n)
  const void* tw_hi29;
  uint32_t lobits;
  int use_hi;
  Fr* num;
  Fr* den;
};
// challenge-derived constants of the fast quotient path as 9 x 29-bit limbs (fr29.cuh):
// c * 2^(5k) * R'' for multipliers, plain re-sliced R-form values for addends.
struct QuotientConst {
  uint32_t gamma[9];       // gamma (R form, addend)
  uint32_t one[9];         // 1 (R form, subtrahend of z - 1)
  uint32_t beta_k[4][9];   // beta * {1, K1, K2, K3} * R''
  uint32_t alpha_pos[9];   // alpha * 2^20 * R''   (absorbs the 2^-5 of 4 data x data products)
  uint32_t alpha_neg[9];   // -alpha * 2^20 * R''
  uint32_t alpha_sq[9];    // alpha^2 * R''
  uint32_t vinv[8][9];     // vanishing_coset_inverses * R''
};
struct QuotientArgs {
  uint64_t n8;               // quotient-domain size (8n as the reference, or 4n: prover.hip)
  uint32_t rot;              // index distance of the X -> omega X rotation = n8 / n
  QuotientConst k;
  Fr inv32;                // 2^-5: undoes the pre-scaling of q_l / q_r for the fixed-base widget
  const Fr *a, *b, *c, *d, *z, *pi;
  const Fr *q_m, *q_l, *q_r, *q_o, *q_f, *q_c, *q_arith, *q_range, *q_logic, *q_fixed, *q_var;
  const Fr *s1, *s2, *s3, *s4, *linear, *l1;
  bool has[QS_COUNT];   // selector polynomial is not identically zero
  Fr range_ch, logic_ch, fixed_ch, var_ch, edwards_d;   // exact path (rarely active widgets)
  Fr* out;
};
struct L1Args {
  Fr vh[8];
  Fr n_inv;
};
struct SigmaArgs {
  Fr ks[4];   // 1, K1, K2, K3 (src/composer/permutation/constants.rs)
};
struct EvalItem {
  const Fr* poly;
  uint64_t len;
  Fr x;
};
struct EvalArgs {
  EvalItem items[16];
  Fr* partial;
  uint32_t max_blocks;
};
struct LinTerm {
  const Fr* p;
  uint64_t len;
  Fr s;
};
struct LinCombArgs {
  LinTerm t[24];
  int count;
  uint64_t len;
  Fr constant;
  Fr* out;
};

// ---- multi-GPU: residue classes of the quotient coset (prover.hip) -------------------------------
struct ShardPackArgs {
  const Fr* F[8];      // per owned class: n coefficients of t mod (X^n - s_j^n)
  Fr* send;            // [peer][class][stride]
  uint64_t n, per, stride;   // stride = per + 8
  uint32_t cpr;
};
struct ShardCombineArgs {
  const Fr* recv;      // [source rank][class of that rank][stride]
  uint64_t stride, per, lo, hi, cnt, n;   // this rank owns coefficient indices [lo, hi); cnt = those below n
  uint32_t W, cpr, Q;
  Fr coef[5][8];       // g^(-n i1) w_Q^(-j i1) / Q
  Fr low[7];           // true lowest coefficients of t (quotient_low), Q == 4 only
  Fr g4n_inv;
  Fr* parts[4];        // t_low, t_mid, t_high, t_fourth (indexed by global coefficient)
};
struct ShardSplitFix {
  Fr* parts[4];
  uint64_t lo, hi, n;
  Fr b[3];
};
int poly_fold(Ctx* c, const Fr* src, Fr* dst, uint64_t n, uint32_t extra, const Fr& cn);
int poly_shard_pack(Ctx* c, const ShardPackArgs& a, uint32_t world);
int poly_shard_combine(Ctx* c, const ShardCombineArgs& a);
int poly_shard_split_fix(Ctx* c, const ShardSplitFix& a);
int poly_ruffini_local(Ctx* c, const Fr* src, uint64_t lo, uint64_t len, const Fr& z, Fr* scratch, Fr* totals);
int poly_ruffini_finish(Ctx* c, const Fr* scratch, Fr* dst, uint64_t lo, uint64_t len, const Fr& zinv, const Fr& carry, uint64_t last);

void prof_begin(Ctx* c, int slot);
void prof_end(Ctx* c, int slot);
void prof_host_add(Ctx* c, int slot, double ms);   // host wall time on a slot (no events): prover.hip HostGap

int poly_fill_zero(Ctx* c, Fr* p, uint64_t n);
int poly_blind(Ctx* c, Fr* coeffs, uint64_t n, const BlindArgs& a);
int poly_split_t(Ctx* c, Fr* t, uint64_t n, uint64_t np, Fr* out, const SplitArgs& a);
int poly_trimmed_len(Ctx* c, const Fr* p, uint64_t n, unsigned long long* out_dev);
int poly_scatter_pi(Ctx* c, Fr* dense, const uint64_t* idx, const Fr* val, uint64_t count);
// Compiler::preprocess: sigma evaluations from the packed mappings of permutation.hpp; wire columns from witness values
int poly_sigma_evals(Ctx* c, const uint32_t* map_dev, const Fr* roots, Fr* out, uint64_t n, uint64_t stride, const SigmaArgs& a);
int poly_gather_wires(Ctx* c, const uint32_t* idx_dev, const Fr* values, Fr* wires, uint64_t constraints, uint64_t n);
int poly_batch_inverse(Ctx* c, Fr* v, uint64_t n, bool twiddle_form = false);
int poly_perm_terms(Ctx* c, const PermArgs& a);
int poly_mul_arrays(Ctx* c, Fr* a, const Fr* b, uint64_t n, int* zero_flag);   // operands in twiddle form
int poly_quotient(Ctx* c, const QuotientArgs& q);
// de-aliasing of a quotient interpolated on the 4n coset (prover.hip)
int poly_dealias(Ctx* c, Fr* t, uint64_t nq, const Fr low[7], const Fr& g_inv);
int poly_l1(Ctx* c, const Fr* linear, Fr* l1, uint64_t n8, const L1Args& a);
int poly_scale_array(Ctx* c, Fr* v, uint64_t n, const Fr& s);
int poly_from_mont(Ctx* c, const Fr* src, Fr* dst, uint64_t n);   // canonical limbs (BlsScalar::to_bytes); dst may equal src
void quotient_const(const Fr& c, int shift, uint32_t out[9]);
void quotient_data(const Fr& c, uint32_t out[9]);
int poly_eval(Ctx* c, EvalArgs& a, int count, uint64_t max_len, Fr* out_dev);
int poly_lincomb(Ctx* c, const LinCombArgs& a);
int poly_ruffini(Ctx* c, const Fr* src, Fr* dst, uint64_t len, const Fr& z, const Fr& zinv, Fr* scratch, Fr* totals);
int scan_prefix_product(Ctx* c, Fr* data, uint64_t n, Fr* totals);
// the same scan in two steps for a RANGE of a longer product (sharded grand product): _local leaves the inclusive products of
// the range's blocks in totals (the range's own product, twiddle form, at totals[scan_prefix_blocks(n) - 1]); _apply multiplies
// every element by `carry` (twiddle form: the product of everything before the range) and converts to the data domain
uint32_t scan_prefix_blocks(uint64_t n);
int scan_prefix_product_local(Ctx* c, Fr* data, uint64_t n, Fr* totals);
int scan_prefix_product_apply(Ctx* c, Fr* data, uint64_t n, const Fr* totals, const Fr& carry_twiddle);
int scan_suffix_sum(Ctx* c, Fr* data, uint64_t n, Fr* totals);

}  // namespace plonk
