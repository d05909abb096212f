// Host-side G1 work of the device prover (no HIP): finishing an MSM from its 16 bit sums, the shared
// affine normalisation of a commitment group, and the 48-byte compressed encoding.  64-bit-limb
// Montgomery arithmetic — this code sits between two GPU phases with the GPU idle.  Included by
// prover.hip and by the CPU test harness (tests/csrc/host_arith.cpp).
#pragma once
#include <cstring>

#include "curve28.cuh"   // curve.cuh + the constant-time safegcd inverse (fp_safegcd.cuh)

namespace plonk {

// G1Affine::to_bytes: 48-byte BE x, flags 0x80 | 0x40 inf | 0x20 y > -y  (commitment.rs:46-57)
static void g1_compress97(const uint8_t in[97], uint8_t out[48]) {
  if (in[96]) {
    memset(out, 0, 48);
    out[0] = 0xC0;
    return;
  }
  Fp x, y;
  memcpy(x.l, in, 48);
  memcpy(y.l, in + 48, 48);
  const Fp xc = x.from_mont(), yc = y.from_mont(), nyc = y.neg().from_mont();
  for (int i = 0; i < 12; ++i) {
    const uint32_t w = xc.l[11 - i];
    out[4 * i] = (uint8_t)(w >> 24); out[4 * i + 1] = (uint8_t)(w >> 16);
    out[4 * i + 2] = (uint8_t)(w >> 8); out[4 * i + 3] = (uint8_t)w;
  }
  bool greater = false;   // y > -y lexicographically (as integers)
  for (int i = 11; i >= 0; --i) {
    if (yc.l[i] != nyc.l[i]) { greater = yc.l[i] > nyc.l[i]; break; }
  }
  out[0] |= 0x80;
  if (greater) out[0] |= 0x20;
}

// ---- host-side affine normalisation of the MSM results ----------------------------------------
// The transcript needs the commitments of a group before the next kernels can be queued, so this
// sits on the critical path with the GPU idle: 64-bit-limb Montgomery arithmetic (the generic
// 32-bit Field<> costs 130 us per Fp inversion on the host) and ONE shared inversion per group.
struct Fp64 {
  uint64_t l[6];
};
static uint64_t fp64_ninv() {   // -p^-1 mod 2^64 by Newton iteration
  uint64_t p0 = (uint64_t)FpP::MOD[0] | ((uint64_t)FpP::MOD[1] << 32), inv = 1;
  for (int i = 0; i < 6; ++i) inv *= 2 - p0 * inv;
  return 0 - inv;
}
static Fp64 fp64_mod() {
  Fp64 m;
  for (int i = 0; i < 6; ++i) m.l[i] = (uint64_t)FpP::MOD[2 * i] | ((uint64_t)FpP::MOD[2 * i + 1] << 32);
  return m;
}
// Montgomery product, R = 2^384 (same Montgomery form as Fp).  The product row a * b_i and the reduction row m * p run as
// two interleaved carry chains over SIX NAMED limbs, written out so that everything stays in registers (mulx / adcx-
// friendly): 50 ns against 88 ns for the looped CIOS on the build host — this code runs with the GPU idle.
static inline Fp64 fp64_mul(const Fp64& a, const Fp64& b) {
  static const uint64_t NINV = fp64_ninv();
  static const Fp64 MM = fp64_mod();
  typedef unsigned __int128 u128;
  const uint64_t m0 = MM.l[0], m1 = MM.l[1], m2 = MM.l[2], m3 = MM.l[3], m4 = MM.l[4], m5 = MM.l[5];
  uint64_t t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0, t6 = 0;
#define PLONK_FP64_ROW(bi)                                                                                  \
  {                                                                                                         \
    u128 c = (u128)a.l[0] * (bi) + t0;                                                                      \
    const uint64_t lo = (uint64_t)c;                                                                        \
    c >>= 64;                                                                                               \
    const uint64_t m = lo * NINV;                                                                           \
    u128 d = ((u128)m * m0 + lo) >> 64;                                                                     \
    c += (u128)a.l[1] * (bi) + t1; d += (u128)m * m1 + (uint64_t)c; t0 = (uint64_t)d; c >>= 64; d >>= 64;   \
    c += (u128)a.l[2] * (bi) + t2; d += (u128)m * m2 + (uint64_t)c; t1 = (uint64_t)d; c >>= 64; d >>= 64;   \
    c += (u128)a.l[3] * (bi) + t3; d += (u128)m * m3 + (uint64_t)c; t2 = (uint64_t)d; c >>= 64; d >>= 64;   \
    c += (u128)a.l[4] * (bi) + t4; d += (u128)m * m4 + (uint64_t)c; t3 = (uint64_t)d; c >>= 64; d >>= 64;   \
    c += (u128)a.l[5] * (bi) + t5; d += (u128)m * m5 + (uint64_t)c; t4 = (uint64_t)d; c >>= 64; d >>= 64;   \
    d += (u128)t6 + (uint64_t)c;                                                                            \
    t5 = (uint64_t)d;                                                                                       \
    t6 = (uint64_t)(d >> 64);                                                                               \
  }
  PLONK_FP64_ROW(b.l[0]) PLONK_FP64_ROW(b.l[1]) PLONK_FP64_ROW(b.l[2])
  PLONK_FP64_ROW(b.l[3]) PLONK_FP64_ROW(b.l[4]) PLONK_FP64_ROW(b.l[5])
#undef PLONK_FP64_ROW
  const uint64_t t[6] = {t0, t1, t2, t3, t4, t5};
  Fp64 r, d;
  uint64_t borrow = 0;
  for (int j = 0; j < 6; ++j) {
    r.l[j] = t[j];
    const u128 s = (u128)t[j] - MM.l[j] - borrow;
    d.l[j] = (uint64_t)s;
    borrow = (uint64_t)(s >> 64) & 1;
  }
  return (t6 || !borrow) ? d : r;
}
static Fp64 fp64_inv_fermat(const Fp64& a) {   // a^(p-2), fixed 4-bit windows: 380 squarings + <= 95 + 14 products (the cross-check of fp64_inv)
  Fp64 e = fp64_mod();
  e.l[0] -= 2;   // p is odd and p mod 2^64 > 2: no borrow
  Fp64 tab[16];  // tab[k] = a^k, k >= 1
  tab[1] = a;
  for (int k = 2; k < 16; ++k) tab[k] = fp64_mul(tab[k - 1], a);
  Fp64 acc = a;
  bool started = false;
  for (int w = 5; w >= 0; --w)
    for (int b = 60; b >= 0; b -= 4) {
      const unsigned nib = (unsigned)(e.l[w] >> b) & 15u;
      if (!started) {
        if (nib) { acc = tab[nib]; started = true; }
        continue;
      }
      acc = fp64_mul(acc, acc); acc = fp64_mul(acc, acc); acc = fp64_mul(acc, acc); acc = fp64_mul(acc, acc);
      if (nib) acc = fp64_mul(acc, tab[nib]);
    }
  return acc;
}
// a R -> a^-1 R with the Bernstein-Yang divstep inverse of fp_safegcd.cuh (the same code the kernels use, run on the host):
// ~5 us instead of the ~27 us of the Fermat chain, once per commitment group between two GPU phases.  The plain inverse of
// the integer a R is a^-1 R^-1; one Montgomery product with R^3 gives a^-1 R.  0 -> 0 like the Fermat chain.
static Fp64 fp64_inv(const Fp64& a) {
  static const Fp64 R3 = [] {
    Fp r2;
    for (int i = 0; i < 12; ++i) r2.l[i] = FpP::R2[i];
    Fp64 x;
    memcpy(x.l, r2.l, 48);
    return fp64_mul(x, x);   // R^2 R^2 / R
  }();
  uint32_t w[12];
  memcpy(w, a.l, 48);
  const safegcd::Signed30<13> d = safegcd::inverse<safegcd::FpMod>(safegcd::to30<13, 32>(w));
  Fp64 y;
  memset(&y, 0, sizeof y);
  for (int i = 0; i < 13; ++i) {   // 13 x 30 bits -> 6 x 64 (the value is < p < 2^381)
    const int bit = 30 * i, q = bit >> 6, sh = bit & 63;
    const uint64_t v = (uint64_t)(uint32_t)d.v[i];
    y.l[q] |= v << sh;
    if (sh > 34 && q + 1 < 6) y.l[q + 1] |= v >> (64 - sh);
  }
  return fp64_mul(y, R3);
}
static Fp64 fp64_add(const Fp64& a, const Fp64& b) {
  static const Fp64 M = fp64_mod();
  typedef unsigned __int128 u128;
  Fp64 r, d;
  u128 c = 0;
  for (int j = 0; j < 6; ++j) { c += (u128)a.l[j] + b.l[j]; r.l[j] = (uint64_t)c; c >>= 64; }
  uint64_t borrow = 0;
  for (int j = 0; j < 6; ++j) {
    const u128 s = (u128)r.l[j] - M.l[j] - borrow;
    d.l[j] = (uint64_t)s;
    borrow = (uint64_t)(s >> 64) & 1;
  }
  return ((uint64_t)c || !borrow) ? d : r;
}
static Fp64 fp64_sub(const Fp64& a, const Fp64& b) {
  static const Fp64 M = fp64_mod();
  typedef unsigned __int128 u128;
  Fp64 r;
  uint64_t borrow = 0;
  for (int j = 0; j < 6; ++j) {
    const u128 s = (u128)a.l[j] - b.l[j] - borrow;
    r.l[j] = (uint64_t)s;
    borrow = (uint64_t)(s >> 64) & 1;
  }
  if (borrow) {
    u128 c = 0;
    for (int j = 0; j < 6; ++j) { c += (u128)r.l[j] + M.l[j]; r.l[j] = (uint64_t)c; c >>= 64; }
  }
  return r;
}
static bool fp64_is_zero(const Fp64& a) { return (a.l[0] | a.l[1] | a.l[2] | a.l[3] | a.l[4] | a.l[5]) == 0; }

// XYZZ group law on the host (same formulas as curve.cuh: EFD dbl-2008-s-1 / add-2008-s), canonical
// coordinates; ZZ == 0 marks the identity.
struct H1 {
  Fp64 X, Y, ZZ, ZZZ;
  bool inf() const { return fp64_is_zero(ZZ); }
};
static H1 h1_dbl(const H1& p) {
  if (p.inf()) return p;
  const Fp64 U = fp64_add(p.Y, p.Y), V = fp64_mul(U, U), W = fp64_mul(U, V), S = fp64_mul(p.X, V);
  const Fp64 XX = fp64_mul(p.X, p.X), M = fp64_add(fp64_add(XX, XX), XX);
  H1 r;
  r.X = fp64_sub(fp64_mul(M, M), fp64_add(S, S));
  r.Y = fp64_sub(fp64_mul(M, fp64_sub(S, r.X)), fp64_mul(W, p.Y));
  r.ZZ = fp64_mul(V, p.ZZ);
  r.ZZZ = fp64_mul(W, p.ZZZ);
  return r;
}
static H1 h1_add(const H1& a, const H1& b) {
  if (a.inf()) return b;
  if (b.inf()) return a;
  const Fp64 U1 = fp64_mul(a.X, b.ZZ), U2 = fp64_mul(b.X, a.ZZ), S1 = fp64_mul(a.Y, b.ZZZ), S2 = fp64_mul(b.Y, a.ZZZ);
  const Fp64 P = fp64_sub(U2, U1), R = fp64_sub(S2, S1);
  if (fp64_is_zero(P)) {
    if (fp64_is_zero(R)) return h1_dbl(a);
    H1 id;
    memset(&id, 0, sizeof id);
    return id;
  }
  const Fp64 PP = fp64_mul(P, P), PPP = fp64_mul(P, PP), Q = fp64_mul(U1, PP);
  H1 r;
  r.X = fp64_sub(fp64_sub(fp64_mul(R, R), PPP), fp64_add(Q, Q));
  r.Y = fp64_sub(fp64_mul(R, fp64_sub(Q, r.X)), fp64_mul(S1, PPP));
  r.ZZ = fp64_mul(fp64_mul(a.ZZ, b.ZZ), PP);
  r.ZZZ = fp64_mul(fp64_mul(a.ZZZ, b.ZZZ), PPP);
  return r;
}
// W = sum_j 2^j T'_j + 2^7 C_128 + sum_j 2^(7+j) T_j  from the bit sums of msm_bits_kernel: rows T_0..T_{rb-1} (rb = 8 for
// 2^15 buckets = 256 rows of 128, 12 for 2^19 buckets), columns T'_0..T'_6, C_128 — Horner over U_0..U_{6+rb}.  W weighs
// bucket b with b + 1: the commitment for window-table entries.  Bit-position entries (NAF digits, msm_recode.cuh) weigh
// 2 b + 1: their commitment is 2 W - S with S = the sum of all buckets.  Layout: rows [0, rb), columns [rb, rb + 7),
// C_128 at rb + 7, S at rb + 8.
static constexpr int MSM_ROWBITS_MAX = 12;
static G1 finish_bit_sums(const G1* bits, int rb, bool bitpos) {
  H1 u[MSM_ROWBITS_MAX + 9];
  for (int k = 0; k < rb + 9; ++k) {
    memcpy(u[k].X.l, bits[k].X.l, 48); memcpy(u[k].Y.l, bits[k].Y.l, 48);
    memcpy(u[k].ZZ.l, bits[k].ZZ.l, 48); memcpy(u[k].ZZZ.l, bits[k].ZZZ.l, 48);
  }
  H1 U[MSM_ROWBITS_MAX + 7];
  for (int j = 0; j < 7; ++j) U[j] = u[rb + j];
  U[7] = h1_add(u[rb + 7], u[0]);
  for (int j = 1; j < rb; ++j) U[7 + j] = u[j];
  H1 acc = U[6 + rb];
  for (int j = 5 + rb; j >= 0; --j) acc = h1_add(h1_dbl(acc), U[j]);
  if (bitpos) {
    H1 negS = u[rb + 8];
    Fp64 zero;
    memset(&zero, 0, sizeof zero);
    if (!negS.inf()) negS.Y = fp64_sub(zero, negS.Y);   // p - Y
    acc = h1_add(h1_dbl(acc), negS);
  }
  G1 r;
  if (acc.inf()) return G1::identity();
  memcpy(r.X.l, acc.X.l, 48); memcpy(r.Y.l, acc.Y.l, 48); memcpy(r.ZZ.l, acc.ZZ.l, 48); memcpy(r.ZZZ.l, acc.ZZZ.l, 48);
  return r;
}

// sum of n XYZZ points `stride` bytes apart (the per-rank partial sums of one commitment after the all-gather of a sharded
// proof): the 64-bit-limb additions above, ~0.4 us each — the generic 32-bit-limb G1::add this replaces took ~3 us, i.e.
// ~0.25 ms of host time per 8-rank proof between device phases
static G1 h1_sum_strided(const uint8_t* first, size_t stride, int n) {
  H1 acc;
  memset(&acc, 0, sizeof acc);
  for (int r = 0; r < n; ++r) {
    G1 g;
    memcpy(&g, first + (size_t)r * stride, sizeof(G1));
    H1 h;
    memcpy(h.X.l, g.X.l, 48); memcpy(h.Y.l, g.Y.l, 48); memcpy(h.ZZ.l, g.ZZ.l, 48); memcpy(h.ZZZ.l, g.ZZZ.l, 48);
    acc = h1_add(acc, h);
  }
  if (acc.inf()) return G1::identity();
  G1 out;
  memcpy(out.X.l, acc.X.l, 48); memcpy(out.Y.l, acc.Y.l, 48); memcpy(out.ZZ.l, acc.ZZ.l, 48); memcpy(out.ZZZ.l, acc.ZZZ.l, 48);
  return out;
}

static Fp64 to64(const Fp& x) { Fp64 r; memcpy(r.l, x.l, 48); return r; }
static Fp from64(const Fp64& x) { Fp r; memcpy(r.l, x.l, 48); return r; }

// x = X / ZZ, y = Y / ZZZ for a group of XYZZ points -> 97-byte raw affine (x || y || infinity)
static void batch_xyzz_to_affine97(const G1* pts, int count, uint8_t (*out)[97]) {
  Fp64 den[16], pre[16];
  int idx[16], m = 0;
  for (int i = 0; i < count; ++i) {
    memset(out[i], 0, 97);
    if (pts[i].is_identity()) { out[i][96] = 1; continue; }
    den[m] = fp64_mul(to64(pts[i].ZZ), to64(pts[i].ZZZ));
    pre[m] = m ? fp64_mul(pre[m - 1], den[m]) : den[m];
    idx[m++] = i;
  }
  if (!m) return;
  Fp64 inv = fp64_inv(pre[m - 1]);
  for (int k = m - 1; k >= 0; --k) {
    const Fp64 dinv = k ? fp64_mul(inv, pre[k - 1]) : inv;
    if (k) inv = fp64_mul(inv, den[k]);
    const G1& p = pts[idx[k]];
    const Fp x = from64(fp64_mul(to64(p.X), fp64_mul(dinv, to64(p.ZZZ))));
    const Fp y = from64(fp64_mul(to64(p.Y), fp64_mul(dinv, to64(p.ZZ))));
    memcpy(out[idx[k]], x.l, 48);
    memcpy(out[idx[k]] + 48, y.l, 48);
  }
}

// G1Affine::from_bytes on a 48-byte compressed encoding, as Commitment::from_reader applies it to the
// VerifierKey commitments (widget.rs:113-134): compression flag set; infinity <=> every other bit clear;
// x canonical (< p); x^3 + 4 a square; the point in the prime-order subgroup.  Validity only — the
// bytes themselves seed the transcript.
static Fp64 fp64_pow(const Fp64& a, const Fp64& e) {
  Fp64 acc = to64(Fp::one());
  for (int w = 5; w >= 0; --w)
    for (int b = 63; b >= 0; --b) {
      acc = fp64_mul(acc, acc);
      if ((e.l[w] >> b) & 1) acc = fp64_mul(acc, a);
    }
  return acc;
}
static bool g1_compressed_valid(const uint8_t in[48]) {
  const uint8_t flags = in[0];
  if (!(flags & 0x80)) return false;
  Fp64 raw;
  for (int i = 0; i < 6; ++i) {
    uint64_t v = 0;
    for (int b = 0; b < 8; ++b) {
      const int pos = 8 * (5 - i) + b;
      v = (v << 8) | (pos == 0 ? (uint8_t)(in[0] & 0x1f) : in[pos]);
    }
    raw.l[i] = v;
  }
  if (flags & 0x40) return !(flags & 0x20) && fp64_is_zero(raw);
  const Fp64 M = fp64_mod();
  bool lt = false;
  for (int i = 5; i >= 0; --i)
    if (raw.l[i] != M.l[i]) { lt = raw.l[i] < M.l[i]; break; }
  if (!lt) return false;
  Fp r2;
  for (int i = 0; i < 12; ++i) r2.l[i] = FpP::R2[i];
  const Fp64 x = fp64_mul(raw, to64(r2));                       // Montgomery form
  const Fp64 y2 = fp64_add(fp64_mul(fp64_mul(x, x), x), to64(Fp::from_u64(4)));
  Fp64 e = M;                                                    // (p + 1) / 4  (p = 3 mod 4)
  e.l[0] += 1;                                                   // p mod 2^64 is not all-ones: no carry
  for (int i = 0; i < 6; ++i) e.l[i] = (e.l[i] >> 2) | (i < 5 ? e.l[i + 1] << 62 : 0);
  const Fp64 y = fp64_pow(y2, e);
  if (!fp64_is_zero(fp64_sub(fp64_mul(y, y), y2))) return false;   // not on the curve
  H1 P, acc;
  P.X = x; P.Y = y; P.ZZ = to64(Fp::one()); P.ZZZ = P.ZZ;
  memset(&acc, 0, sizeof acc);
  for (int b = 254; b >= 0; --b) {                               // [q]P == O  (is_torsion_free)
    acc = h1_dbl(acc);
    if ((FrP::MOD[b >> 5] >> (b & 31)) & 1) acc = h1_add(acc, P);
  }
  return acc.inf();
}


}  // namespace plonk
