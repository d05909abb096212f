// Modular inversion in BLS12-381 Fp by Bernstein-Yang "safegcd" division steps (constant time: no data-dependent
// branch, so the 64 lanes of a wave stay converged), in 13 signed limbs of 30 bits — the layout of libsecp256k1's
// modinv32, re-derived here for a 381-bit modulus.
//
// Why: a Fermat inversion (fp28_inv) is 380 squarings + ~190 products = ~570 Montgomery products = ~300 k
// instructions.  Division steps work on the LOW 30 bits of (f, g) for 30 steps at a time, collect them in a 2 x 2
// transition matrix with entries |.| <= 2^30, and only then touch the full-size numbers: a batch is ~500 instructions
// of 32-bit work + 2 x 52 signed 32 x 32 -> 64 multiply-accumulates, and 30 batches (900 steps; the loop below also
// runs until g = 0, so the count is not a correctness assumption) finish a 381-bit inversion in ~23 k instructions:
// ~13x cheaper.  Used where an inversion sits on a latency- or throughput-critical path: shared-inversion affine
// additions (tools/ubench/affine_batch.hip measures them against the XYZZ mixed addition of msm_accumulate).
//
// Interface: value in, value out, both as Fp28 Montgomery residues (x R' -> x^-1 R').  0 -> 0.
#pragma once
#include "fp28.cuh"

namespace plonk {

struct Signed30 {
  static constexpr int N = 13;
  int32_t v[N];
};

namespace safegcd {

static constexpr int32_t M30 = (1 << 30) - 1;

// p in 30-bit limbs, and p^-1 mod 2^30
HD constexpr int32_t mod30(int i) {
  constexpr int32_t P[Signed30::N] = {0x3fffaaab, 0x27fbffff, 0x153ffffb, 0x2affffac, 0x30f6241e, 0x034a83da, 0x112bf673,
                                      0x12e13ce1, 0x2cd76477, 0x1ed90d2e, 0x29a4b1ba, 0x3a8e5ff9, 0x001a0111};
  return P[i];
}
static constexpr uint32_t MOD_INV30 = 0x00030003u;   // p^-1 mod 2^30

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
HD void update_fg(Signed30* f, Signed30* g, const Trans& t) {
  const int64_t u = t.u, v = t.v, q = t.q, r = t.r;
  int64_t cf = u * f->v[0] + v * g->v[0];
  int64_t cg = q * f->v[0] + r * g->v[0];
  cf >>= 30;
  cg >>= 30;
#pragma unroll
  for (int i = 1; i < Signed30::N; ++i) {
    const int64_t fi = f->v[i], gi = g->v[i];
    cf += u * fi + v * gi;
    cg += q * fi + r * gi;
    f->v[i - 1] = (int32_t)cf & M30;
    g->v[i - 1] = (int32_t)cg & M30;
    cf >>= 30;
    cg >>= 30;
  }
  f->v[Signed30::N - 1] = (int32_t)cf;
  g->v[Signed30::N - 1] = (int32_t)cg;
}

// (d, e) <- t * (d, e) / 2^30 mod p, with d, e kept in (-2p, p)
HD void update_de(Signed30* d, Signed30* e, const Trans& t, uint32_t mod_inv30) {
  const int64_t u = t.u, v = t.v, q = t.q, r = t.r;
  const int32_t sd = d->v[Signed30::N - 1] >> 31, se = e->v[Signed30::N - 1] >> 31;   // sign masks
  int32_t md = (t.u & sd) + (t.v & se), me = (t.q & sd) + (t.r & se);
  const int64_t di = d->v[0], ei = e->v[0];
  int64_t cd = u * di + v * ei, ce = q * di + r * ei;
  md -= (int32_t)((mod_inv30 * (uint32_t)cd + (uint32_t)md) & (uint32_t)M30);
  me -= (int32_t)((mod_inv30 * (uint32_t)ce + (uint32_t)me) & (uint32_t)M30);
  cd += (int64_t)mod30(0) * md;
  ce += (int64_t)mod30(0) * me;
  cd >>= 30;
  ce >>= 30;
#pragma unroll
  for (int i = 1; i < Signed30::N; ++i) {
    const int64_t dk = d->v[i], ek = e->v[i];
    cd += u * dk + v * ek + (int64_t)mod30(i) * md;
    ce += q * dk + r * ek + (int64_t)mod30(i) * me;
    d->v[i - 1] = (int32_t)cd & M30;
    e->v[i - 1] = (int32_t)ce & M30;
    cd >>= 30;
    ce >>= 30;
  }
  d->v[Signed30::N - 1] = (int32_t)cd;
  e->v[Signed30::N - 1] = (int32_t)ce;
}

// r in (-2p, p) -> [0, p), negated first when sign < 0
HD void normalize(Signed30* r, int32_t sign) {
  int32_t cond_add = r->v[Signed30::N - 1] >> 31;
  const int32_t cond_neg = sign >> 31;
  int32_t c = 0;
#pragma unroll
  for (int i = 0; i < Signed30::N; ++i) {
    int32_t x = r->v[i] + (mod30(i) & cond_add);
    x = (x ^ cond_neg) - cond_neg;
    x += c;
    if (i < Signed30::N - 1) { c = x >> 30; x &= M30; }
    r->v[i] = x;
  }
  cond_add = r->v[Signed30::N - 1] >> 31;
  c = 0;
#pragma unroll
  for (int i = 0; i < Signed30::N; ++i) {
    int32_t x = r->v[i] + (mod30(i) & cond_add) + c;
    if (i < Signed30::N - 1) { c = x >> 30; x &= M30; }
    r->v[i] = x;
  }
}

HD bool is_zero(const Signed30& a) {
  int32_t acc = 0;
#pragma unroll
  for (int i = 0; i < Signed30::N; ++i) acc |= a.v[i];
  return acc == 0;
}

}  // namespace safegcd

// x^-1 mod p for a canonical integer 0 <= x < p given in 30-bit limbs (0 -> 0)
HD Signed30 fp_inv_plain30(const Signed30& x, uint32_t mod_inv30) {
  Signed30 d, e, f, g = x;
#pragma unroll
  for (int i = 0; i < Signed30::N; ++i) { d.v[i] = 0; e.v[i] = 0; f.v[i] = safegcd::mod30(i); }
  e.v[0] = 1;
  int32_t zeta = -1;
  for (int it = 0; it < 30 || !safegcd::is_zero(g); ++it) {   // 900 steps cover 381-bit inputs; the test on g makes that a fact, not an assumption
    safegcd::Trans t;
    zeta = safegcd::divsteps_30(zeta, (uint32_t)f.v[0] | ((uint32_t)f.v[1] << 30), (uint32_t)g.v[0] | ((uint32_t)g.v[1] << 30), &t);
    safegcd::update_de(&d, &e, t, mod_inv30);
    safegcd::update_fg(&f, &g, t);
  }
  safegcd::normalize(&d, f.v[Signed30::N - 1]);   // f = +-1 (or +-p for x = 0, where d = 0)
  return d;
}

// Fp28 (14 x 28 bits, canonical value) <-> 13 x 30 bits
HD Signed30 to_signed30(const Fp28& a) {
  Signed30 r;
#pragma unroll
  for (int i = 0; i < Signed30::N; ++i) {
    const int bit = 30 * i, w = bit / 28, sh = bit % 28;
    uint64_t v = (uint64_t)a.l[w] >> sh;
    if (w + 1 < Fp28::N) v |= (uint64_t)a.l[w + 1] << (28 - sh);
    if (w + 2 < Fp28::N) v |= (uint64_t)a.l[w + 2] << (56 - sh);
    r.v[i] = (int32_t)(v & (uint64_t)safegcd::M30);
  }
  return r;
}
HD Fp28 from_signed30(const Signed30& a) {
  Fp28 r;
#pragma unroll
  for (int i = 0; i < Fp28::N; ++i) {
    const int bit = 28 * i, w = bit / 30, sh = bit % 30;
    uint64_t v = (uint64_t)(uint32_t)a.v[w] >> sh;
    if (w + 1 < Signed30::N) v |= (uint64_t)(uint32_t)a.v[w + 1] << (30 - sh);
    r.l[i] = (uint32_t)v & Fp28::MASK;
  }
  return r;
}

// x R' -> x^-1 R' (value of the input below 64 p, as everywhere in curve28.cuh); 0 -> 0
HD Fp28 fp28_inv_gcd(const Fp28& a) {
  // the plain inverse of the integer x R' is x^-1 R'^-1; one Montgomery product with R'^3 mod p (a plain integer)
  // gives x^-1 R'^-1 R'^3 / R' = x^-1 R'
  constexpr uint32_t R3[Fp28::N] = {0x1f7b890u, 0x294cc4du, 0x9f3af22u, 0xb5ba56cu, 0xcb5c0ccu, 0xc0d975cu, 0xc89a8c5u,
                                    0x6c968b4u, 0x22672eau, 0x91de8c9u, 0x35652a6u, 0x84977c8u, 0x424bbb9u, 0x00141abu};
  Fp28 c;
#pragma unroll
  for (int i = 0; i < Fp28::N; ++i) c.l[i] = R3[i];
  const Signed30 y = fp_inv_plain30(to_signed30(a.canon()), safegcd::MOD_INV30);
  return Fp28::mul(from_signed30(y), c);
}

}  // namespace plonk
