// Host-side shim exposing the product's __host__ __device__ field/curve code
// (plonk_amd/csrc/field.cuh, curve.cuh) to ctypes so it can be checked bit for
// bit against the big-int oracle on a CPU-only box.  Test code only.
#include <cstring>
#include "../../plonk_amd/csrc/curve.cuh"
using namespace plonk;
extern "C" {
void h_fr_mul(const uint32_t* a, const uint32_t* b, uint32_t* o) { Fr x, y; memcpy(&x, a, 32); memcpy(&y, b, 32); Fr r = x * y; memcpy(o, &r, 32); }
void h_fr_add(const uint32_t* a, const uint32_t* b, uint32_t* o) { Fr x, y; memcpy(&x, a, 32); memcpy(&y, b, 32); Fr r = x + y; memcpy(o, &r, 32); }
void h_fr_sub(const uint32_t* a, const uint32_t* b, uint32_t* o) { Fr x, y; memcpy(&x, a, 32); memcpy(&y, b, 32); Fr r = x - y; memcpy(o, &r, 32); }
void h_fr_inv(const uint32_t* a, uint32_t* o) { Fr x; memcpy(&x, a, 32); Fr r = x.inv(); memcpy(o, &r, 32); }
void h_fr_from_mont(const uint32_t* a, uint32_t* o) { Fr x; memcpy(&x, a, 32); Fr r = x.from_mont(); memcpy(o, &r, 32); }
void h_fr_consts(uint32_t* o) { Fr g = fr_generator(), w = fr_root_of_unity(), one = Fr::one(); memcpy(o, &g, 32); memcpy(o + 8, &w, 32); memcpy(o + 16, &one, 32); }
void h_fp_mul(const uint32_t* a, const uint32_t* b, uint32_t* o) { Fp x, y; memcpy(&x, a, 48); memcpy(&y, b, 48); Fp r = x * y; memcpy(o, &r, 48); }
void h_fp_add(const uint32_t* a, const uint32_t* b, uint32_t* o) { Fp x, y; memcpy(&x, a, 48); memcpy(&y, b, 48); Fp r = x + y; memcpy(o, &r, 48); }
void h_fp_sub(const uint32_t* a, const uint32_t* b, uint32_t* o) { Fp x, y; memcpy(&x, a, 48); memcpy(&y, b, 48); Fp r = x - y; memcpy(o, &r, 48); }
void h_fp_inv(const uint32_t* a, uint32_t* o) { Fp x; memcpy(&x, a, 48); Fp r = x.inv(); memcpy(o, &r, 48); }
// points: affine 96 B (x||y Montgomery); result affine 96 B + return 1, or 0 for identity
static int out_aff(const G1& p, uint8_t* o) { G1Affine a; bool ok = p.to_affine(&a); memcpy(o, &a, 96); return ok ? 1 : 0; }
int h_g1_add_aff(const uint8_t* a, const uint8_t* b, uint8_t* o) { G1Affine x, y; memcpy(&x, a, 96); memcpy(&y, b, 96); return out_aff(G1::from_affine(x).add_affine(y), o); }
int h_g1_add_full(const uint8_t* a, const uint8_t* b, uint8_t* o) {
  G1Affine x, y; memcpy(&x, a, 96); memcpy(&y, b, 96);
  // de-normalise both operands so the general formulas are exercised
  G1 p = G1::from_affine(x).dbl().add_affine(x).add(G1::from_affine(x).dbl().neg());   // = x, with ZZ != 1
  G1 q = G1::from_affine(y).dbl().add_affine(y).add(G1::from_affine(y).dbl().neg());
  return out_aff(p.add(q), o);
}
int h_g1_mul_u32(const uint8_t* a, uint32_t k, uint8_t* o) { G1Affine x; memcpy(&x, a, 96); return out_aff(G1::from_affine(x).mul_u32(k), o); }
int h_g1_neg_add(const uint8_t* a, uint8_t* o) { G1Affine x; memcpy(&x, a, 96); G1Affine n = x; n.y = x.y.neg(); return out_aff(G1::from_affine(x).add_affine(n), o); }
}
// ---- reduced-radix Fp (fp28.cuh) ----
#include "../../plonk_amd/csrc/fp28.cuh"
extern "C" {
// inputs/outputs in the 12 x 32-bit R = 2^384 form; computation done in Fp28
void h_fp28_mul(const uint32_t* a, const uint32_t* b, uint32_t* o) { Fp x, y; memcpy(&x, a, 48); memcpy(&y, b, 48); Fp r = Fp28::mul(Fp28::from_fp(x), Fp28::from_fp(y)).to_fp(); memcpy(o, &r, 48); }
void h_fp28_chain(const uint32_t* a, const uint32_t* b, uint32_t* o) {
  // exercises lazy add/sub bounds: ((a + b) * (a - b + 4p)) - (a*a) + (b*b) ... = 0 ; returns a*b + that
  Fp x, y; memcpy(&x, a, 48); memcpy(&y, b, 48);
  Fp28 A = Fp28::from_fp(x), Bv = Fp28::from_fp(y);
  Fp28 s = Fp28::add(A, Bv), d = Fp28::sub<4>(A, Bv);
  Fp28 t = Fp28::mul(s, d);                         // a^2 - b^2
  Fp28 u = Fp28::sub<4>(t, A.sqr());                // -b^2 (+4p)
  Fp28 v = Fp28::add(u, Bv.sqr());                  // 0 mod p, value < 8p
  Fp28 w = Fp28::add(Fp28::mul(A, Bv), v);
  Fp r = w.to_fp(); memcpy(o, &r, 48);
}
int h_fp28_zero_test(const uint32_t* a) { Fp x; memcpy(&x, a, 48); Fp28 A = Fp28::from_fp(x); Fp28 z = Fp28::sub<4>(A, A); Fp28 z2 = Fp28::sub<32>(Fp28::add(Fp28::add(A, A), A.dbl().dbl()), Fp28::add(A.dbl(), A.dbl().dbl())); return (z.is_zero_mod() ? 1 : 0) | (z2.is_zero_mod() ? 2 : 0) | (A.is_zero_mod() ? 4 : 0); }
void h_fp28_roundtrip(const uint32_t* a, uint32_t* o) { Fp x; memcpy(&x, a, 48); Fp r = Fp28::from_fp(x).to_fp(); memcpy(o, &r, 48); }
}
// raw-limb access (14 x u32, possibly lazy): op 0 = mul(a,b), 1 = a.sqr(), 2 = mul2(a,b,c,d)
extern "C" void h_fp28_raw(int op, const uint32_t* a, const uint32_t* b, const uint32_t* c, const uint32_t* d, uint32_t* o) {
  Fp28 A, Bv, C, D;
  memcpy(A.l, a, 56); memcpy(Bv.l, b, 56); memcpy(C.l, c, 56); memcpy(D.l, d, 56);
  Fp28 r = op == 0 ? Fp28::mul(A, Bv) : op == 1 ? A.sqr() : Fp28::mul2(A, Bv, C, D);
  memcpy(o, r.l, 56);
}
// lazy helpers on normalised inputs: op 0 = sub_lazy<32>(a,b), 1 = neg_lazy<16>(b), 2 = add_lazy(a,b)
extern "C" void h_fp28_lazy(int op, const uint32_t* a, const uint32_t* b, uint32_t* o) {
  Fp28 A, Bv;
  memcpy(A.l, a, 56); memcpy(Bv.l, b, 56);
  Fp28 r = op == 0 ? Fp28::sub_lazy<32>(A, Bv) : op == 1 ? Fp28::neg_lazy<16>(Bv) : Fp28::add_lazy(A, Bv);
  memcpy(o, r.l, 56);
}

// ---- XYZZ over Fp28 (curve28.cuh) ----
#include "../../plonk_amd/csrc/curve28.cuh"
extern "C" {
static int out_aff_r(const G1R& p, uint8_t* o) { return out_aff(p.to_g1(), o); }
// sum_{i<n} (neg[i] ? -P_i : P_i) with mixed additions, points 96 B each in the 32-bit form
int h_g1r_accumulate(const uint8_t* pts, const uint8_t* neg, int n, uint8_t* o) {
  G1R acc = G1R::identity();
  for (int i = 0; i < n; ++i) {
    G1Affine a; memcpy(&a, pts + 96 * i, 96);
    Fp28 x = Fp28::from_fp(a.x), y = Fp28::from_fp(a.y);
    if (neg[i]) y = Fp28::sub<4>(Fp28::zero(), y);
    acc = acc.add_affine(x, y);
  }
  return out_aff_r(acc, o);
}
// (sum of first half) + (sum of second half) via the full addition; then * k
int h_g1r_tree(const uint8_t* pts, int n, uint32_t k, uint8_t* o) {
  G1R a = G1R::identity(), b = G1R::identity();
  for (int i = 0; i < n; ++i) {
    G1Affine p; memcpy(&p, pts + 96 * i, 96);
    Fp28 x = Fp28::from_fp(p.x), y = Fp28::from_fp(p.y);
    if (i < n / 2) a = a.add_affine(x, y); else b = b.add_affine(x, y);
  }
  G1R s = a.add(b);
  s = s.add(s);            // doubling through add()
  return out_aff_r(s.mul_u32(k), o);
}
// The accumulation lanes' first step (msm.hip ACC_FIRST_PAIR): entries 0 and 1 through add_affine_pair when their x differ
// (signs applied lazily as 4p - y, as the kernels do), then the rest through add_affine.  Returns like h_g1r_accumulate;
// *used_pair = 1 when the pair formula ran.
int h_g1r_accumulate_pair_first(const uint8_t* pts, const uint8_t* neg, int n, uint8_t* o, int* used_pair) {
  auto signed_y = [](const Fp28& y, bool ng) {
    Fp28 r;
    for (int i = 0; i < Fp28::N; ++i) r.l[i] = ng ? Fp28::pad<4>(i) - y.l[i] : y.l[i];
    return r;
  };
  G1R acc = G1R::identity();
  int k0 = 0;
  *used_pair = 0;
  if (n >= 2) {
    G1Affine a, b; memcpy(&a, pts, 96); memcpy(&b, pts + 96, 96);
    const Fp28 xa = Fp28::from_fp(a.x), ya = Fp28::from_fp(a.y), xb = Fp28::from_fp(b.x), yb = Fp28::from_fp(b.y);
    if (G1R::pair_distinct(xa, xb)) {
      acc = G1R::add_affine_pair(xa, signed_y(ya, neg[0] != 0), xb, signed_y(yb, neg[1] != 0));
      k0 = 2;
      *used_pair = 1;
    }
  }
  for (int i = k0; i < n; ++i) {
    G1Affine a; memcpy(&a, pts + 96 * i, 96);
    acc = acc.add_affine(Fp28::from_fp(a.x), signed_y(Fp28::from_fp(a.y), neg[i] != 0));
  }
  return out_aff_r(acc, o);
}
// [k] P through the endomorphism (curve28.cuh g1r_mul_glv: the group FFT's scalar multiplication); k: 8 x 32-bit limbs, canonical.
// pre = doublings applied to P first (an operand with the bounds the FFT's butterflies hand over, not a fresh affine point)
int h_g1r_mul_glv(const uint8_t* pt, const uint32_t* k, int pre, uint8_t* o) {
  G1Affine p; memcpy(&p, pt, 96);
  G1R q = G1R::from_affine(Fp28::from_fp(p.x), Fp28::from_fp(p.y));
  for (int i = 0; i < pre; ++i) q = q.add(q);
  return out_aff_r(g1r_mul_glv(q, k), o);
}
void h_glv_split(const uint32_t* k, uint64_t* out4) {
  const GlvScalar g = glv_split(k);
  out4[0] = g.k1[0]; out4[1] = g.k1[1]; out4[2] = g.k2[0]; out4[3] = g.k2[1];
}
int h_g1r_affine_roundtrip(const uint8_t* pt, uint8_t* o) {
  G1Affine p; memcpy(&p, pt, 96);
  G1R q = G1R::from_affine(Fp28::from_fp(p.x), Fp28::from_fp(p.y)).dbl().dbl();
  Fp28 x, y; g1r_to_affine(q, &x, &y);
  G1Affine r; r.x = x.to_fp(); r.y = y.to_fp(); memcpy(o, &r, 96); return 1;
}
}
// ---- reduced-radix Fr (fr29.cuh) ----
#include "../../plonk_amd/csrc/fr29.cuh"
extern "C" {
// DIF butterfly on Montgomery (R = 2^256) inputs: out0 = a + b, out1 = (a - b) * w
void h_fr29_butterfly(const uint32_t* a, const uint32_t* b, const uint32_t* w, uint32_t* o0, uint32_t* o1) {
  Fr x, y, t; memcpy(&x, a, 32); memcpy(&y, b, 32); memcpy(&t, w, 32);
  Fr29 A = Fr29::from_fr(x), Bv = Fr29::from_fr(y), W = Fr29::twiddle_from_fr(t);
  Fr r0 = Fr29::add_csub(A, Bv).to_fr();
  Fr r1 = Fr29::mul(Fr29::sub_lazy(A, Bv), W).to_fr();
  memcpy(o0, &r0, 32); memcpy(o1, &r1, 32);
}
// 9 chained stages on a vector of 2 elements: exercises the lazy ranges (sum path and product path)
void h_fr29_chain(const uint32_t* a, const uint32_t* b, const uint32_t* w, int stages, uint32_t* o0, uint32_t* o1) {
  Fr x, y, t; memcpy(&x, a, 32); memcpy(&y, b, 32); memcpy(&t, w, 32);
  Fr29 A = Fr29::from_fr(x), Bv = Fr29::from_fr(y), W = Fr29::twiddle_from_fr(t);
  for (int s = 0; s < stages; ++s) {
    Fr29 n0 = Fr29::add_csub(A, Bv);
    Fr29 n1 = Fr29::mul(Fr29::sub_lazy(A, Bv), W);
    A = n0; Bv = n1;
  }
  Fr r0 = A.to_fr(), r1 = Bv.to_fr();
  memcpy(o0, &r0, 32); memcpy(o1, &r1, 32);
}
void h_fr29_mul2(const uint32_t* a, const uint32_t* w1, const uint32_t* w2, uint32_t* o) {   // a * (w1 * w2)
  Fr x, t1, t2; memcpy(&x, a, 32); memcpy(&t1, w1, 32); memcpy(&t2, w2, 32);
  Fr29 W = Fr29::mul(Fr29::twiddle_from_fr(t1), Fr29::twiddle_from_fr(t2));
  Fr r = Fr29::mul(Fr29::from_fr(x), W).to_fr(); memcpy(o, &r, 32);
}
}
extern "C" void h_fr29_sub_reduce(const uint32_t* a, const uint32_t* b, uint32_t* o) {
  Fr x, y; memcpy(&x, a, 32); memcpy(&y, b, 32);
  // operands first pushed to the top of the lazy range: (x + 0) via add_csub keeps them, so use doubled values
  Fr29 A = Fr29::from_fr(x), Bv = Fr29::from_fr(y);
  Fr r = Fr29::sub_reduce(Fr29::add_csub(A, A), Fr29::add_csub(Bv, Bv)).to_fr(); memcpy(o, &r, 32);
}

// ---- host transcript (transcript.hpp): Merlin's published test protocol ----
#include "../../plonk_amd/csrc/transcript.hpp"
extern "C" void h_merlin_simple(uint8_t out[32]) {
  plonk::Transcript t((const uint8_t*)"test protocol", 13);
  t.append_message("some label", (const uint8_t*)"some data", 9);
  t.challenge_bytes("challenge", out, 32);
}

// ---- widgets.hpp: lowest 7 coefficients of the quotient from the lowest 7 of every polynomial ----
#include "../../plonk_amd/csrc/widgets.hpp"
// key_low: 15 x 7 Fr (PolyId order), has: 11 flags, low: a b c d z pi (6 x 7 Fr),
// ch: alpha beta gamma range logic fixed var edwards_d omega n_inv (10 Fr); out: 7 Fr.  All Montgomery limbs.
extern "C" void h_quotient_low(const uint32_t* key_low, const uint8_t* has, const uint32_t* low, const uint32_t* ch, uint32_t* out) {
  using namespace plonk;
  Fr kl[P_COUNT][7];
  memcpy(kl, key_low, sizeof kl);
  bool hs[WQS_COUNT];
  for (int i = 0; i < WQS_COUNT; ++i) hs[i] = has[i] != 0;
  Fr lows[42], c[10], o[7];
  memcpy(lows, low, sizeof lows);
  memcpy(c, ch, sizeof c);
  QuotientLowIn in;
  in.low = lows;
  in.alpha = c[0]; in.beta = c[1]; in.gamma = c[2]; in.range_ch = c[3]; in.logic_ch = c[4]; in.fixed_ch = c[5];
  in.var_ch = c[6]; in.edwards_d = c[7]; in.omega = c[8]; in.n_inv = c[9];
  quotient_low(kl, hs, in, o);
  memcpy(out, o, sizeof o);
}

// ---- hostg1.hpp: MSM finishing (Horner over the 16 bit sums), group normalisation, compression ----
#include "../../plonk_amd/csrc/hostg1.hpp"
// pts: 16 affine points (96 B raw each; a point with x = y = 0 stands for the identity) -> XYZZ bit sums
// -> finish_bit_sums -> batch affine -> 48-byte compressed.  out48: the commitment.
// rb row bit sums, 7 column bit sums, C_128, and (bitpos != 0) S: the result is 2 W - S (bit-position entries weigh 2 b + 1)
extern "C" void h_finish_bit_sums(const uint8_t* pts96, int rb, int bitpos, uint8_t out48[48]) {
  using namespace plonk;
  G1 bits[MSM_ROWBITS_MAX + 9];
  bits[rb + 8] = G1::identity();
  for (int k = 0; k < rb + 8 + (bitpos ? 1 : 0); ++k) {
    G1Affine a;
    memcpy(&a, pts96 + 96 * k, 96);
    bits[k] = (a.x.is_zero() && a.y.is_zero()) ? G1::identity() : G1::from_affine(a);
    if (k & 1) bits[k] = bits[k].dbl().add(bits[k].neg());   // a non-trivial ZZ: 2P - P
  }
  const G1 w = finish_bit_sums(bits, rb, bitpos != 0);
  uint8_t aff[1][97];
  batch_xyzz_to_affine97(&w, 1, aff);
  g1_compress97(aff[0], out48);
}
// count <= 16 XYZZ points given as affine (made projective with odd scalings) -> compressed encodings
extern "C" void h_batch_compress(const uint8_t* pts96, int count, uint8_t* out48) {
  using namespace plonk;
  G1 p[16];
  for (int k = 0; k < count; ++k) {
    G1Affine a;
    memcpy(&a, pts96 + 96 * k, 96);
    if (a.x.is_zero() && a.y.is_zero()) { p[k] = G1::identity(); continue; }
    p[k] = G1::from_affine(a);
    for (int j = 0; j < k % 3; ++j) p[k] = p[k].dbl().add(p[k].neg()).add(G1::identity());
  }
  uint8_t aff[16][97];
  batch_xyzz_to_affine97(p, count, aff);
  for (int k = 0; k < count; ++k) g1_compress97(aff[k], out48 + 48 * k);
}

// sum of `n` points (affine in, made projective with odd scalings) with the 64-bit-limb additions of a sharded proof's
// partial-sum step (h1_sum_strided) -> compressed
extern "C" void h_sum_strided(const uint8_t* pts96, int n, uint8_t out48[48]) {
  using namespace plonk;
  G1 p[64];
  for (int k = 0; k < n; ++k) {
    G1Affine a;
    memcpy(&a, pts96 + 96 * k, 96);
    if (a.x.is_zero() && a.y.is_zero()) { p[k] = G1::identity(); continue; }
    p[k] = G1::from_affine(a);
    for (int j = 0; j < k % 3; ++j) p[k] = p[k].dbl().add(p[k].neg());
  }
  const G1 w = h1_sum_strided(reinterpret_cast<const uint8_t*>(p), sizeof(G1), n);
  uint8_t aff[1][97];
  batch_xyzz_to_affine97(&w, 1, aff);
  g1_compress97(aff[0], out48);
}
// ---- finish_pool.hpp: the host helper threads of fetch_commitments ----
#include "../../plonk_amd/csrc/finish_pool.hpp"
namespace {
struct PoolProbe { std::atomic<int> hits[16]; std::atomic<long> sum; int spin; };
void pool_probe_task(void* arg, int i) {
  PoolProbe* pp = (PoolProbe*)arg;
  volatile unsigned x = 1;
  for (int k = 0; k < pp->spin; ++k) x = x * 1664525u + 1013904223u;   // a few hundred ns .. a few us of work
  pp->hits[i].fetch_add(1);
  pp->sum.fetch_add(i + 1);
}
}
// `rounds` jobs of 0..16 tasks with every arming pattern fetch_commitments can produce (armed + run, armed + withdrawn,
// late arming, jobs back to back): returns 0 when every task of every job ran exactly once and run() returned only after
// the last one; workers == 0 is the inline path.
extern "C" int h_finish_pool_selftest(int workers, int rounds) {
  using namespace plonk;
  FinishPool pool(workers);
  if (pool.workers() > (workers < 0 ? 0 : (workers > 7 ? 7 : workers))) return -1;   // (fewer only if the process is out of threads)
  unsigned lcg = 12345u + (unsigned)workers;
  for (int r = 0; r < rounds; ++r) {
    lcg = lcg * 1664525u + 1013904223u;
    const int count = (int)((lcg >> 8) % 17);
    const int pattern = (int)((lcg >> 16) % 4);
    PoolProbe probe;
    for (auto& h : probe.hits) h.store(0);
    probe.sum.store(0);
    probe.spin = (int)((lcg >> 20) % 2000);
    if (pattern == 1) { Armed withdrawn(workers > 0 ? &pool : nullptr); }   // armed, nothing to do (an early return)
    {
      Armed a(pattern == 2 ? nullptr : (workers > 0 ? &pool : nullptr));     // pattern 2: the inline path of a one-commitment group
      if (pattern == 3) std::this_thread::sleep_for(std::chrono::microseconds(200));   // workers awake and spinning before the job
      a.run(pool_probe_task, &probe, count);
    }
    long want = 0;
    for (int i = 0; i < 16; ++i) {
      if (probe.hits[i].load() != (i < count ? 1 : 0)) return 100 + r;
      if (i < count) want += i + 1;
    }
    if (probe.sum.load() != want) return 200 + r;
  }
  return 0;
}

// ---- permutation.hpp: copy constraints -> sigma mappings (Permutation::compute_sigma_permutations) ----
#include "../../plonk_amd/csrc/permutation.hpp"
extern "C" int h_sigma_mappings(const uint32_t* a, const uint32_t* b, const uint32_t* c, const uint32_t* d, uint64_t constraints,
                                uint64_t n, uint64_t witnesses, uint32_t* out) {
  const uint32_t* wires[4] = {a, b, c, d};
  return plonk::sigma_mappings(wires, constraints, n, witnesses, out) ? 0 : -1;
}

#include "../../plonk_amd/csrc/g1codec.cuh"
// G1Affine::from_bytes on a 48-byte compressed encoding (g1codec.cuh): returns the decoder's code, out = x || y (96 B, Montgomery)
extern "C" int h_g1_decompress48(const uint8_t* in, uint8_t* out96) {
  G1Affine a;
  memset(&a, 0, sizeof a);
  const int rc = g1_decompress48(in, &a);
  memcpy(out96, &a, 96);
  return rc;
}

// ---- msm_recode.cuh: scalar -> (row, bucket, sign) digits of both recodings ----
#include "../../plonk_amd/csrc/msm_recode.cuh"
// canonical scalar (8 x u32) -> out[4 j .. 4 j + 3] = slot, row, bucket, sign; returns the number of digits
extern "C" int h_msm_recode(const uint32_t* scalar, int bitpos, uint32_t* out) {
  Big<8> s;
  memcpy(s.l, scalar, 32);
  int n = 0;
  auto emit = [&](int slot, uint32_t row, uint32_t bucket, uint32_t sign) {
    out[4 * n] = (uint32_t)slot; out[4 * n + 1] = row; out[4 * n + 2] = bucket; out[4 * n + 3] = sign;
    ++n;
  };
  if (bitpos == 120 || bitpos == 116) {   // even-position digits (half-density tables), width 20 / 16, LDS-parked form
    uint32_t park[9 * 3];
    for (int k = 0; k < 8; ++k) park[3 * k + 1] = s.l[k];
    park[3 * 8 + 1] = 0;
    if (bitpos == 120) for_each_digit_even<20>(StridedLimbs{park + 1, 3}, emit);
    else for_each_digit_even<16>(StridedLimbs{park + 1, 3}, emit);
    return n;
  }
  if (bitpos == 21) {  // the 2^19-bucket variant's digit width
    for_each_digit_naf<21>(s, emit);
  } else if (bitpos == 2) {   // the kernels' form: the scalar parked limb-major with a stride (StridedLimbs)
    uint32_t park[9 * 3];
    for (int j = 0; j < 9; ++j) { park[3 * j] = 0xdeadbeefu; park[3 * j + 1] = j < 8 ? s.l[j] : 0u; park[3 * j + 2] = 0x12345678u; }
    for_each_digit_bitpos(StridedLimbs{park + 1, 3}, emit);
  } else {
    for_each_digit(s, bitpos ? MSM_ROWS_BITPOS : MSM_ROWS_WINDOW, emit);
  }
  return n;
}

// ---- fp_safegcd.cuh: Bernstein-Yang inversion against Fermat (fp28_inv) and the oracle ----
#include "../../plonk_amd/csrc/fp_safegcd.cuh"
// a: Fp (12 x u32, R = 2^384 Montgomery) -> o: its inverse in the same form, through Fp28 and the safegcd inverse
extern "C" void h_fp_inv_gcd(const uint32_t* a, uint32_t* o) {
  Fp x; memcpy(&x, a, 48);
  Fp r = fp28_inv_gcd(Fp28::from_fp(x)).to_fp();
  memcpy(o, &r, 48);
}
// same input scaled lazily (value 5x + 3x = 8x as unreduced limbs < 64p): the inverse of 8x
extern "C" void h_fp_inv_gcd_lazy(const uint32_t* a, uint32_t* o) {
  Fp x; memcpy(&x, a, 48);
  const Fp28 A = Fp28::from_fp(x);
  const Fp28 A8 = Fp28::add(Fp28::add(A.dbl().dbl(), A), Fp28::add(A.dbl(), A));
  Fp r = fp28_inv_gcd(A8).to_fp();
  memcpy(o, &r, 48);
}

// Fr: a (8 x u32, R = 2^256 Montgomery) -> its inverse in the same form, through twiddle form and the safegcd inverse
extern "C" void h_fr_inv_gcd(const uint32_t* a, uint32_t* o) {
  Fr x; memcpy(&x, a, 32);
  const Fr29 inv_t = fr29_inv_gcd_tw(Fr29::twiddle_from_fr(x));          // x^-1 * 2^261
  Fr r = Fr29::mul(inv_t, Fr29::from_fr(Fr::one())).to_fr();              // * R / 2^261 = x^-1 R
  memcpy(o, &r, 32);
}

// ---- hostg2.hpp: validity of a compressed G2 encoding (OpeningKey::from_slice's test of h and x_h) ----
#include "../../plonk_amd/csrc/hostg2.hpp"
extern "C" int h_g2_compressed_valid(const uint8_t in[96]) { return plonk::g2_compressed_valid(in) ? 1 : 0; }
extern "C" int h_g1_compressed_valid(const uint8_t in[48]) { return plonk::g1_compressed_valid(in) ? 1 : 0; }

// ---- hostg1.hpp: the host-side Fp inverse (Montgomery in, Montgomery out): safegcd (mode 0) and the Fermat chain (mode 1) ----
extern "C" void h_fp64_inv(const uint8_t in[48], int fermat, uint8_t out[48]) {
  plonk::Fp64 a;
  memcpy(a.l, in, 48);
  const plonk::Fp64 r = fermat ? plonk::fp64_inv_fermat(a) : plonk::fp64_inv(a);
  memcpy(out, r.l, 48);
}
// the Montgomery-in / Montgomery-out Fr inverse the host driver uses per proof (fp_safegcd.cuh fr_inv_gcd)
extern "C" void h_fr_inv_gcd_mont(const uint32_t* a, uint32_t* o) {
  Fr x; memcpy(&x, a, 32);
  const Fr r = fr_inv_gcd(x);
  memcpy(o, &r, 32);
}

// ---- api_guard.hpp: the exception barrier every int-returning C-ABI entry point runs inside ----
#include <cstdio>
#include <stdexcept>
#include "../../plonk_amd/csrc/api_guard.hpp"
static char g_guard_msg[256];
namespace plonk {
void set_last_error(const char* what, const char* detail, const char*, int) { snprintf(g_guard_msg, sizeof g_guard_msg, "%s -> %s", what, detail); }
}
// kind 0: the body's own return value; 1: std::bad_alloc; 2: std::runtime_error; 3: a non-std exception
extern "C" int h_api_guard(int kind, char msg_out[256]) {
  g_guard_msg[0] = 0;
  const int rc = plonk::api_guard("h_api_guard", [&]() -> int {
    if (kind == 1) throw std::bad_alloc();
    if (kind == 2) throw std::runtime_error("boom");
    if (kind == 3) throw 42;
    return 7;
  });
  memcpy(msg_out, g_guard_msg, 256);
  return rc;
}

// ---- tools/ubench/coop_mul.hpp: the 16-lane cooperative Montgomery product, lanes emulated with arrays ----
#include "../../tools/ubench/coop_mul.hpp"
namespace {
struct HostLanes {   // a value per lane of one 16-lane DPP row; shifts fill with zero like row_shr / row_shl with bound_ctrl
  struct u32 { uint32_t v[16]; };
  struct u64 { uint64_t v[16]; };
  static u64 zero64() { u64 r; for (int k = 0; k < 16; ++k) r.v[k] = 0; return r; }
  static u64 mad(const u64& acc, uint32_t uni, const u32& lane) { u64 r; for (int k = 0; k < 16; ++k) r.v[k] = acc.v[k] + (uint64_t)uni * lane.v[k]; return r; }
  template <int S> static u32 shr32(const u32& x) { u32 r; for (int k = 0; k < 16; ++k) r.v[k] = k - S >= 0 ? x.v[k - S] : 0; return r; }
  template <int S> static u32 shl32(const u32& x) { u32 r; for (int k = 0; k < 16; ++k) r.v[k] = k + S <= 15 ? x.v[k + S] : 0; return r; }
  template <int S> static u64 shr64(const u64& x) { u64 r; for (int k = 0; k < 16; ++k) r.v[k] = k - S >= 0 ? x.v[k - S] : 0; return r; }
  template <int S> static u64 shl64(const u64& x) { u64 r; for (int k = 0; k < 16; ++k) r.v[k] = k + S <= 15 ? x.v[k + S] : 0; return r; }
  static u32 lane_lt(int n) { u32 r; for (int k = 0; k < 16; ++k) r.v[k] = k < n ? 0xffffffffu : 0u; return r; }
  static u32 lane_eq(int n) { u32 r; for (int k = 0; k < 16; ++k) r.v[k] = k == n ? 0xffffffffu : 0u; return r; }
  static u64 select64(const u32& m, const u64& a, const u64& b) { u64 r; for (int k = 0; k < 16; ++k) r.v[k] = m.v[k] ? a.v[k] : b.v[k]; return r; }
  static u32 and32(const u32& a, uint32_t c) { u32 r; for (int k = 0; k < 16; ++k) r.v[k] = a.v[k] & c; return r; }
  static u32 and32(const u32& a, const u32& b) { u32 r; for (int k = 0; k < 16; ++k) r.v[k] = a.v[k] & b.v[k]; return r; }
  static u32 add32(const u32& a, const u32& b) { u32 r; for (int k = 0; k < 16; ++k) r.v[k] = a.v[k] + b.v[k]; return r; }
  static u64 add64(const u64& a, const u64& b) { u64 r; for (int k = 0; k < 16; ++k) r.v[k] = a.v[k] + b.v[k]; return r; }
  static u32 lo32(const u64& a) { u32 r; for (int k = 0; k < 16; ++k) r.v[k] = (uint32_t)a.v[k]; return r; }
  static u64 widen(const u32& a) { u64 r; for (int k = 0; k < 16; ++k) r.v[k] = a.v[k]; return r; }
  static u64 shr64_bits(const u64& a, int s) { u64 r; for (int k = 0; k < 16; ++k) r.v[k] = a.v[k] >> s; return r; }
  static u32 nonzero32(const u32& a) { u32 r; for (int k = 0; k < 16; ++k) r.v[k] = a.v[k] ? 0xffffffffu : 0u; return r; }
};
}  // namespace
// a: 14 uniform limbs, b: 14 limbs (lane j holds limb j) -> the 16 lanes of the result (lanes 14, 15 must come out zero)
extern "C" void h_coop_mul(const uint32_t a[14], const uint32_t b[14], uint32_t out[16]) {
  HostLanes::u32 bd;
  for (int k = 0; k < 16; ++k) bd.v[k] = k < 14 ? b[k] : 0;
  coop::Uniform au;
  for (int k = 0; k < 14; ++k) au.l[k] = a[k];
  const HostLanes::u32 r = coop::Mul<HostLanes>::mul(au, bd);
  for (int k = 0; k < 16; ++k) out[k] = r.v[k];
}
