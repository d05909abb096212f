// Modular inversion by Bernstein-Yang "safegcd" division steps (constant time: no data-dependent branch, so the 64
// lanes of a wave stay converged), in signed limbs of 30 bits — the layout of libsecp256k1's modinv32, re-derived
// here for the two BLS12-381 moduli: Fp (381 bits, 13 limbs) and Fr (255 bits, 9 limbs).
//
// Why: a Fermat inversion is one squaring per bit of the modulus + ~half as many products, every one of them dependent
// on the last — ~570 products (~300 k instructions) in Fp, ~380 (~65 k) in Fr — and it sits on latency-critical paths:
// ONE lane inverts the total of a 4096-element workgroup in batch_inverse_kernel (poly.hip) while 255 lanes wait.
// Division steps work on the LOW 30 bits of (f, g) for 30 steps at a time, collect them in a 2 x 2 transition matrix
// with entries |.| <= 2^30, and only then touch the full-size numbers: a batch is ~500 instructions of 32-bit work +
// 2 x 4 N signed 32 x 32 -> 64 multiply-accumulates.  30 batches (900 steps) cover 381-bit inputs, 20 (600) cover
// 255-bit ones; the loops below ALSO run until g = 0, so the counts are not a correctness assumption.  ~23 k (Fp) /
// ~13 k (Fr) instructions: 13x / 5x cheaper than Fermat.
// Also measured here: shared-inversion affine additions for the MSM accumulation (tools/ubench/affine_batch.hip,
// profiles/r03a/affine_batch_prototype.txt) — even with this inverse they cost 1.07-1.5x the XYZZ mixed addition.
//
// Interfaces: fp28_inv_gcd: x R' -> x^-1 R' (Fp28); fr29_inv_gcd_tw: x 2^261 -> x^-1 2^261 (Fr29 twiddle form). 0 -> 0.
#pragma once
#include "fp28.cuh"
#include "fr29.cuh"

namespace plonk {

namespace safegcd {

static constexpr int32_t M30 = (1 << 30) - 1;

struct FpMod {   // p
  static constexpr int N = 13;
  static constexpr int MIN_BATCHES = 30;
  static constexpr uint32_t INV30 = 0x00030003u;   // p^-1 mod 2^30
  HD static constexpr int32_t limb(int i) {
    constexpr int32_t P[N] = {0x3fffaaab, 0x27fbffff, 0x153ffffb, 0x2affffac, 0x30f6241e, 0x034a83da, 0x112bf673,
                              0x12e13ce1, 0x2cd76477, 0x1ed90d2e, 0x29a4b1ba, 0x3a8e5ff9, 0x001a0111};
    return P[i];
  }
};
struct FrMod {   // q
  static constexpr int N = 9;
  static constexpr int MIN_BATCHES = 20;
  static constexpr uint32_t INV30 = 0x00000001u;   // q = 1 mod 2^32
  HD static constexpr int32_t limb(int i) {
    constexpr int32_t Q[N] = {0x00000001, 0x3ffffffc, 0x3fe5bfef, 0x2f6900bf, 0x21d80553, 0x27602026, 0x17d48333,
                              0x29d4ca67, 0x000073ed};
    return Q[i];
  }
};

template <int N>
struct Signed30 {
  int32_t v[N];
};

struct Trans {
  int32_t u, v, q, r;
};

// 30 division steps on the low bits of (f, g); zeta = -(delta + 1/2)
HD int32_t divsteps_30(int32_t zeta, uint32_t f0, uint32_t g0, Trans* t) {
  uint32_t u = 1, v = 0, q = 0, r = 1;
  uint32_t f = f0, g = g0;
#pragma unroll
  for (int i = 0; i < 30; ++i) {
    uint32_t c1 = (uint32_t)(zeta >> 31);      // all ones if zeta < 0
    const uint32_t c2 = 0u - (g & 1u);         // all ones if g is odd
    const uint32_t x = (f ^ c1) - c1, y = (u ^ c1) - c1, z = (v ^ c1) - c1;   // conditionally negated f, u, v
    g += x & c2;
    q += y & c2;
    r += z & c2;
    c1 &= c2;                                  // zeta < 0 and g odd: swap roles
    zeta = (zeta ^ (int32_t)c1) - 1;
    f += g & c1;
    u += q & c1;
    v += r & c1;
    g >>= 1;
    u <<= 1;
    v <<= 1;
  }
  t->u = (int32_t)u; t->v = (int32_t)v; t->q = (int32_t)q; t->r = (int32_t)r;
  return zeta;
}

// (f, g) <- t * (f, g) / 2^30  (exact)
template <int N>
HD void update_fg(Signed30<N>* f, Signed30<N>* g, const Trans& t) {
  const int64_t u = t.u, v = t.v, q = t.q, r = t.r;
  int64_t cf = u * f->v[0] + v * g->v[0];
  int64_t cg = q * f->v[0] + r * g->v[0];
  cf >>= 30;
  cg >>= 30;
#pragma unroll
  for (int i = 1; i < N; ++i) {
    const int64_t fi = f->v[i], gi = g->v[i];
    cf += u * fi + v * gi;
    cg += q * fi + r * gi;
    f->v[i - 1] = (int32_t)cf & M30;
    g->v[i - 1] = (int32_t)cg & M30;
    cf >>= 30;
    cg >>= 30;
  }
  f->v[N - 1] = (int32_t)cf;
  g->v[N - 1] = (int32_t)cg;
}

// (d, e) <- t * (d, e) / 2^30 mod m, with d, e kept in (-2m, m)
template <class M>
HD void update_de(Signed30<M::N>* d, Signed30<M::N>* e, const Trans& t) {
  constexpr int N = M::N;
  const int64_t u = t.u, v = t.v, q = t.q, r = t.r;
  const int32_t sd = d->v[N - 1] >> 31, se = e->v[N - 1] >> 31;   // sign masks
  int32_t md = (t.u & sd) + (t.v & se), me = (t.q & sd) + (t.r & se);
  const int64_t di = d->v[0], ei = e->v[0];
  int64_t cd = u * di + v * ei, ce = q * di + r * ei;
  md -= (int32_t)((M::INV30 * (uint32_t)cd + (uint32_t)md) & (uint32_t)M30);
  me -= (int32_t)((M::INV30 * (uint32_t)ce + (uint32_t)me) & (uint32_t)M30);
  cd += (int64_t)M::limb(0) * md;
  ce += (int64_t)M::limb(0) * me;
  cd >>= 30;
  ce >>= 30;
#pragma unroll
  for (int i = 1; i < N; ++i) {
    const int64_t dk = d->v[i], ek = e->v[i];
    cd += u * dk + v * ek + (int64_t)M::limb(i) * md;
    ce += q * dk + r * ek + (int64_t)M::limb(i) * me;
    d->v[i - 1] = (int32_t)cd & M30;
    e->v[i - 1] = (int32_t)ce & M30;
    cd >>= 30;
    ce >>= 30;
  }
  d->v[N - 1] = (int32_t)cd;
  e->v[N - 1] = (int32_t)ce;
}

// r in (-2m, m) -> [0, m), negated first when sign < 0
template <class M>
HD void normalize(Signed30<M::N>* r, int32_t sign) {
  constexpr int N = M::N;
  int32_t cond_add = r->v[N - 1] >> 31;
  const int32_t cond_neg = sign >> 31;
  int32_t c = 0;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    int32_t x = r->v[i] + (M::limb(i) & cond_add);
    x = (x ^ cond_neg) - cond_neg;
    x += c;
    if (i < N - 1) { c = x >> 30; x &= M30; }
    r->v[i] = x;
  }
  cond_add = r->v[N - 1] >> 31;
  c = 0;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    int32_t x = r->v[i] + (M::limb(i) & cond_add) + c;
    if (i < N - 1) { c = x >> 30; x &= M30; }
    r->v[i] = x;
  }
}

template <int N>
HD bool is_zero(const Signed30<N>& a) {
  int32_t acc = 0;
#pragma unroll
  for (int i = 0; i < N; ++i) acc |= a.v[i];
  return acc == 0;
}

// x^-1 mod m for a canonical integer 0 <= x < m in 30-bit limbs (0 -> 0)
template <class M>
HD Signed30<M::N> inverse(const Signed30<M::N>& x) {
  constexpr int N = M::N;
  Signed30<N> d, e, f, g = x;
#pragma unroll
  for (int i = 0; i < N; ++i) { d.v[i] = 0; e.v[i] = 0; f.v[i] = M::limb(i); }
  e.v[0] = 1;
  int32_t zeta = -1;
  for (int it = 0; it < M::MIN_BATCHES || !is_zero(g); ++it) {   // the test on g makes the batch count a fact, not an assumption
    Trans t;
    zeta = divsteps_30(zeta, (uint32_t)f.v[0] | ((uint32_t)f.v[1] << 30), (uint32_t)g.v[0] | ((uint32_t)g.v[1] << 30), &t);
    update_de<M>(&d, &e, t);
    update_fg<N>(&f, &g, t);
  }
  normalize<M>(&d, f.v[N - 1]);   // f = +-1 (or +-m for x = 0, where d = 0)
  return d;
}

// re-slicing between unsigned limbs of B bits (canonical value) and 30-bit limbs
template <int NO, int B, int NI>
HD Signed30<NO> to30(const uint32_t (&l)[NI]) {
  Signed30<NO> r;
#pragma unroll
  for (int i = 0; i < NO; ++i) {
    const int bit = 30 * i, w = bit / B, sh = bit % B;
    uint64_t v = w < NI ? (uint64_t)l[w] >> sh : 0;
    if (w + 1 < NI) v |= (uint64_t)l[w + 1] << (B - sh);
    if (w + 2 < NI && 2 * B - sh < 64) v |= (uint64_t)l[w + 2] << (2 * B - sh);
    r.v[i] = (int32_t)(v & (uint64_t)M30);
  }
  return r;
}
template <int NO, int B, int NI>
HD void from30(const Signed30<NI>& a, uint32_t (&l)[NO]) {
#pragma unroll
  for (int i = 0; i < NO; ++i) {
    const int bit = B * i, w = bit / 30, sh = bit % 30;
    uint64_t v = w < NI ? (uint64_t)(uint32_t)a.v[w] >> sh : 0;
    if (w + 1 < NI) v |= (uint64_t)(uint32_t)a.v[w + 1] << (30 - sh);
    l[i] = (uint32_t)v & ((1u << B) - 1u);
  }
}

}  // namespace safegcd

// x R' -> x^-1 R' (value of the input below 64 p, as everywhere in curve28.cuh); 0 -> 0
HD Fp28 fp28_inv_gcd(const Fp28& a) {
  // the plain inverse of the integer x R' is x^-1 R'^-1; one Montgomery product with R'^3 mod p (a plain integer)
  // gives x^-1 R'^-1 R'^3 / R' = x^-1 R'
  constexpr uint32_t R3[Fp28::N] = {0x1f7b890u, 0x294cc4du, 0x9f3af22u, 0xb5ba56cu, 0xcb5c0ccu, 0xc0d975cu, 0xc89a8c5u,
                                    0x6c968b4u, 0x22672eau, 0x91de8c9u, 0x35652a6u, 0x84977c8u, 0x424bbb9u, 0x00141abu};
  Fp28 c, y;
#pragma unroll
  for (int i = 0; i < Fp28::N; ++i) c.l[i] = R3[i];
  const Fp28 ac = a.canon();
  safegcd::from30<Fp28::N, 28>(safegcd::inverse<safegcd::FpMod>(safegcd::to30<13, 28>(ac.l)), y.l);
  return Fp28::mul(y, c);
}

// twiddle form in, twiddle form out: x 2^261 -> x^-1 2^261 (input below 2q, as Fr29::mul returns it); 0 -> 0
HD Fr29 fr29_inv_gcd_tw(const Fr29& a) {
  // plain inverse of x 2^261 is x^-1 2^-261; times 2^783 / 2^261 (one Fr29 product) = x^-1 2^261
  constexpr uint32_t C[Fr29::N] = {0x19d7065du, 0x0020db85u, 0x16122e43u, 0x0edb1ff8u, 0x0fda6124u,
                                   0x0517ac72u, 0x12e6a522u, 0x19d54edau, 0x0009750bu};
  Fr29 c, y;
#pragma unroll
  for (int i = 0; i < Fr29::N; ++i) c.l[i] = C[i];
  const Fr29 ac = a.csub_q();
  safegcd::from30<Fr29::N, 29>(safegcd::inverse<safegcd::FrMod>(safegcd::to30<9, 29>(ac.l)), y.l);
  return Fr29::mul(y, c);
}

// Montgomery form in, Montgomery form out for the 8 x 32-bit Fr of field.cuh (x R -> x^-1 R, R = 2^256); 0 -> 0.  The host
// driver's per-proof inversions (1 / (z (z - 1)) in round 5, the public-input denominators) use it instead of the a^(q-2)
// chain: ~3 us instead of ~45 us between two GPU phases.
HD Fr fr_inv_gcd(const Fr& a) {
  Fr r2;
#pragma unroll
  for (int i = 0; i < 8; ++i) r2.l[i] = FrP::R2[i];
  const Fr r3 = r2 * r2;                                    // stored R^2 R^2 / R = R^3
  const safegcd::Signed30<9> d = safegcd::inverse<safegcd::FrMod>(safegcd::to30<9, 32>(a.l));
  Fr y;
#pragma unroll
  for (int i = 0; i < 8; ++i) {                             // 9 x 30 bits -> 8 x 32
    const int bit = 32 * i, w = bit / 30, sh = bit % 30;
    uint64_t v = (uint64_t)(uint32_t)d.v[w] >> sh;
    if (w + 1 < 9) v |= (uint64_t)(uint32_t)d.v[w + 1] << (30 - sh);
    if (w + 2 < 9) v |= (uint64_t)(uint32_t)d.v[w + 2] << (60 - sh);
    y.l[i] = (uint32_t)v;
  }
  return y * r3;                                            // x^-1 R^-1 R^3 / R
}

}  // namespace plonk
