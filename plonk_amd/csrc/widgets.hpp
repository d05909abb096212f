// Host-side algebra of the device prover (no HIP): polynomial ids, the widget identities as
// templates over the value type, truncated power series, and the low coefficients of the quotient
// (the de-aliasing input of the 4n quotient domain).  Included by prover.hip and by
// the CPU test harness (tests/csrc/host_arith.cpp).
#pragma once
#include "field.cuh"

namespace plonk {

// selector families the quotient kernel can skip when their polynomial is identically zero (poly.hpp
// repeats this enum for the device side)
enum { WQS_RANGE = 7, WQS_LOGIC = 8, WQS_FIXED = 9, WQS_VAR = 10, WQS_COUNT = 11 };

enum PolyId {
  P_QM = 0, P_QL, P_QR, P_QO, P_QF, P_QC, P_QARITH, P_QRANGE, P_QLOGIC, P_QFIXED, P_QVAR,
  P_S1, P_S2, P_S3, P_S4, P_COUNT
};

static inline Fr fr_small(uint64_t v) { return Fr::from_u64(v); }

// Truncated power series mod X^7 over Fr: the widget formulas below are evaluated in this ring
// to obtain the 7 lowest coefficients of the quotient numerator (quotient_low()).
struct Ser {
  static constexpr int K = 7;
  Fr c[K];
  static Ser zero() { Ser r; for (int i = 0; i < K; ++i) r.c[i] = Fr::zero(); return r; }
  static Ser constant(const Fr& v) { Ser r = zero(); r.c[0] = v; return r; }
  static Ser load(const Fr* p) { Ser r; for (int i = 0; i < K; ++i) r.c[i] = p[i]; return r; }
  Ser rotated(const Fr& w) const {   // p(wX)
    Ser r;
    Fr pw = Fr::one();
    for (int i = 0; i < K; ++i) { r.c[i] = c[i] * pw; pw = pw * w; }
    return r;
  }
  Ser sqr() const { return *this * *this; }
  Ser dbl() const { return *this + *this; }
  friend Ser operator+(const Ser& a, const Ser& b) { Ser r; for (int i = 0; i < K; ++i) r.c[i] = a.c[i] + b.c[i]; return r; }
  friend Ser operator-(const Ser& a, const Ser& b) { Ser r; for (int i = 0; i < K; ++i) r.c[i] = a.c[i] - b.c[i]; return r; }
  friend Ser operator*(const Ser& a, const Ser& b) {
    Ser r = zero();
    for (int i = 0; i < K; ++i)
      for (int j = 0; i + j < K; ++j) r.c[i + j] = r.c[i + j] + a.c[i] * b.c[j];
    return r;
  }
  friend Ser operator*(const Ser& a, const Fr& k) { Ser r; for (int i = 0; i < K; ++i) r.c[i] = a.c[i] * k; return r; }
  friend Ser operator*(const Fr& k, const Ser& a) { return a * k; }
  friend Ser operator+(const Ser& a, const Fr& k) { Ser r = a; r.c[0] = r.c[0] + k; return r; }
  friend Ser operator-(const Ser& a, const Fr& k) { Ser r = a; r.c[0] = r.c[0] - k; return r; }
};

template <class T>
struct EvalsT {
  T a, b, c, d, a_w, b_w, d_w, q_arith, q_c, q_l, q_r, s1, s2, s3, z;
};
using Evals = EvalsT<Fr>;

template <class T>
static T delta_h(const T& f) {   // f (f-1)(f-2)(f-3)
  return f * (f - Fr::one()) * (f - fr_small(2)) * (f - fr_small(3));
}

// widget identities: at the evaluation point (T = Fr, the scalar factors of compute_linearization)
// and as power series in X (T = Ser, the same expressions inside compute_quotient_i)
template <class T>
static T range_identity(const Fr& ch, const EvalsT<T>& e) {            // range/proverkey.rs:60-85
  const Fr four = fr_small(4);
  const Fr k1 = ch.sqr(), k2 = k1.sqr(), k3 = k2 * k1;
  return delta_h(e.c - four * e.d) + delta_h(e.b - four * e.c) * k1 + delta_h(e.a - four * e.b) * k2 +
         delta_h(e.d_w - four * e.a) * k3;
}
template <class T>
static T logic_identity(const Fr& ch, const EvalsT<T>& e) {            // logic/proverkey.rs:72-144
  const Fr four = fr_small(4);
  const Fr k1 = ch.sqr(), k2 = k1.sqr(), k3 = k2 * k1, k4 = k3 * k1;
  const T a = e.a_w - four * e.a, b = e.b_w - four * e.b, d = e.d_w - four * e.d, w = e.c;
  const T ab = a + b;
  const T F = w * (w * (four * w - fr_small(18) * ab + fr_small(81)) + fr_small(18) * (a.sqr() + b.sqr()) -
                   fr_small(81) * ab + fr_small(83));
  const T Ee = fr_small(3) * (ab + d) - F.dbl();
  const T Bb = e.q_c * (fr_small(9) * d - fr_small(3) * ab);
  return delta_h(a) + delta_h(b) * k1 + delta_h(d) * k2 + (w - a * b) * k3 + (Bb + Ee) * k4;
}
template <class T>
static T fixed_identity(const Fr& ch, const EvalsT<T>& e, const Fr& ed) {   // fixed_base/proverkey.rs:103-159
  const Fr one = Fr::one();
  const Fr k1 = ch.sqr(), k2 = k1.sqr(), k3 = k2 * k1;
  const T bit = e.d_w - e.d - e.d;
  const T bit_cons = bit * (bit - one) * (bit + one);
  const T y_alpha = bit.sqr() * (e.q_r - one) + one;
  const T x_alpha = e.q_l * bit;
  const T xy_cons = (bit * e.q_c - e.c) * k1;
  const T cab = e.c * e.a * e.b * ed;
  const T x_acc = ((e.a_w + e.a_w * cab) - (x_alpha * e.b + y_alpha * e.a)) * k2;
  const T y_acc = ((e.b_w - e.b_w * cab) - (x_alpha * e.a + y_alpha * e.b)) * k3;
  return bit_cons + x_acc + y_acc + xy_cons;
}
template <class T>
static T var_identity(const Fr& ch, const EvalsT<T>& e, const Fr& ed) {     // curve_addition/proverkey.rs:79-120
  const Fr k1 = ch.sqr();
  const T x1y2 = e.d_w, y1x2 = e.b * e.c, y1y2 = e.b * e.d, x1x2 = e.a * e.c;
  const T dxy = ed * x1y2 * y1x2;
  return (e.a * e.d - x1y2) + ((x1y2 + y1x2) - (e.a_w + e.a_w * dxy)) * k1 +
         ((y1y2 + x1x2) - (e.b_w - e.b_w * dxy)) * k1.sqr();
}

// ---- quotient on the 4n coset ---------------------------------------------------------------
// The reference interpolates t = num / Z_H from 8n evaluations (quotient_poly.rs:96-137).  t has at
// most 4n + 7 coefficients, so 4n evaluations determine it up to aliasing: the inverse coset FFT
// on 4n returns A = t mod (X^4n - g^4n), i.e. A[k] = t[k] + g^4n t[4n + k] for k < 7.  The 7
// lowest coefficients of t come for free: modulo X^7 (and n >= 8) 1/Z_H = 1/(X^n - 1) = -1, so
// t = -num mod X^7, and num mod X^7 only needs the 7 lowest coefficients of every polynomial —
// the numerator formula evaluated in F[X]/(X^7) on the host (~10^4 field multiplications).
// Result: the same t, bit for bit, from half the coset FFT / point-wise work and half the key
// memory.  What changes is how an UNSATISFIED circuit is noticed: the reference sees non-zero
// coefficients above 7n (quotient_poly.rs:132); here the quotient identity is checked at the
// Fiat-Shamir point z (the remainder of the W_z division, free by-product of ruffini), which
// fails to flag an unsatisfied circuit with probability <= 5n/q ~ 2^-230.
// PLONK_QUOTIENT_DOMAIN=8 selects the reference-shaped 8n path (also used when n < 8).
struct QuotientLowIn {
  const Fr* low;   // a b c d z pi, 7 coefficients each
  Fr alpha, beta, gamma, range_ch, logic_ch, fixed_ch, var_ch, edwards_d, omega, n_inv;
};
static void quotient_low(const Fr key_low[P_COUNT][7], const bool has[WQS_COUNT], const QuotientLowIn& in, Fr out[7]) {
  const Fr one = Fr::one();
  const Ser a = Ser::load(in.low), b = Ser::load(in.low + 7), c = Ser::load(in.low + 14), d = Ser::load(in.low + 21);
  const Ser z = Ser::load(in.low + 28), pi = Ser::load(in.low + 35);
  Ser K[P_COUNT];
  for (int k = 0; k < P_COUNT; ++k) K[k] = Ser::load(key_low[k]);
  // arithmetic (arithmetic/proverkey.rs:44-71)
  Ser num = pi + (K[P_QM] * a * b + K[P_QL] * a + K[P_QR] * b + K[P_QO] * c + K[P_QF] * d + K[P_QC]) * K[P_QARITH];
  if (has[WQS_RANGE] || has[WQS_LOGIC] || has[WQS_FIXED] || has[WQS_VAR]) {
    EvalsT<Ser> e;
    e.a = a; e.b = b; e.c = c; e.d = d;
    e.a_w = a.rotated(in.omega); e.b_w = b.rotated(in.omega); e.d_w = d.rotated(in.omega);
    e.q_c = K[P_QC]; e.q_l = K[P_QL]; e.q_r = K[P_QR];
    if (has[WQS_RANGE]) num = num + K[P_QRANGE] * range_identity(in.range_ch, e) * in.range_ch;
    if (has[WQS_LOGIC]) num = num + K[P_QLOGIC] * logic_identity(in.logic_ch, e) * in.logic_ch;
    if (has[WQS_FIXED]) num = num + K[P_QFIXED] * fixed_identity(in.fixed_ch, e, in.edwards_d) * in.fixed_ch;
    if (has[WQS_VAR]) num = num + K[P_QVAR] * var_identity(in.var_ch, e, in.edwards_d) * in.var_ch;
  }
  // permutation (permutation/proverkey.rs:40-125)
  Ser X = Ser::zero();
  X.c[1] = one;
  const Fr ks[4] = {one, fr_small(7), fr_small(13), fr_small(17)};
  const Ser* w[4] = {&a, &b, &c, &d};
  Ser p1 = Ser::constant(one), p2 = Ser::constant(one);
  for (int k = 0; k < 4; ++k) {
    p1 = p1 * (*w[k] + X * (in.beta * ks[k]) + in.gamma);
    p2 = p2 * (*w[k] + K[P_S1 + k] * in.beta + in.gamma);
  }
  num = num + p1 * z * in.alpha - p2 * z.rotated(in.omega) * in.alpha;
  Ser l1;   // L1(X) = (X^n - 1) / (n (X - 1)) = (1 + X + X^2 + ...) / n  mod X^n
  for (int k = 0; k < Ser::K; ++k) l1.c[k] = in.n_inv;
  num = num + (z - one) * l1 * in.alpha.sqr();
  for (int k = 0; k < 7; ++k) out[k] = num.c[k].neg();
}


}  // namespace plonk
