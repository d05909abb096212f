// BLS12-381 G1 (y^2 = x^3 + 4) in extended-Jacobian "XYZZ" coordinates
// (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2; ZZ == 0 encodes the identity).
//
// Role on the path: the group law under msm_variable_base
// (reference src/commitment_scheme/kzg10/key.rs:384, external crate
// dusk-bls12_381).  The MSM result is a unique group element, so any complete
// addition law gives bit-identical commitments after affine normalisation.
// Formulas: EFD "xyzz" madd-2008-s / add-2008-s / dbl-2008-s-1 with explicit
// handling of the P == +-Q and identity cases (needed e.g. for repeated bases).
#pragma once
#include "field.cuh"

namespace plonk {

struct G1Affine {   // Montgomery x, y -- first 96 bytes of G1Affine::to_raw_bytes (key.rs:215-229)
  Fp x, y;
};

struct G1 {
  Fp X, Y, ZZ, ZZZ;

  HD static G1 identity() {
    G1 r;
    r.X = Fp::one();
    r.Y = Fp::one();
    r.ZZ = Fp::zero();
    r.ZZZ = Fp::zero();
    return r;
  }
  HD bool is_identity() const { return ZZ.is_zero(); }

  HD static G1 from_affine(const G1Affine& a) {
    G1 r;
    r.X = a.x;
    r.Y = a.y;
    r.ZZ = Fp::one();
    r.ZZZ = Fp::one();
    return r;
  }

  HD G1 neg() const {
    G1 r = *this;
    r.Y = Y.neg();
    return r;
  }

  // dbl-2008-s-1 (general ZZ)
  HD G1 dbl() const {
    if (is_identity() || Y.is_zero()) return identity();
    Fp U = Y.dbl();
    Fp V = U.sqr();
    Fp W = U * V;
    Fp S = X * V;
    Fp XX = X.sqr();
    Fp M = XX.dbl() + XX;   // a = 0
    G1 r;
    r.X = M.sqr() - S.dbl();
    r.Y = M * (S - r.X) - W * Y;
    r.ZZ = V * ZZ;
    r.ZZZ = W * ZZZ;
    return r;
  }

  // doubling of an affine point (ZZ = ZZZ = 1): mdbl-2008-s-1
  HD static G1 dbl_affine(const G1Affine& a) {
    if (a.y.is_zero()) return identity();
    Fp U = a.y.dbl();
    Fp V = U.sqr();
    Fp W = U * V;
    Fp S = a.x * V;
    Fp XX = a.x.sqr();
    Fp M = XX.dbl() + XX;
    G1 r;
    r.X = M.sqr() - S.dbl();
    r.Y = M * (S - r.X) - W * a.y;
    r.ZZ = V;
    r.ZZZ = W;
    return r;
  }

  // mixed addition this + (x2, y2), affine operand never the identity
  HD G1 add_affine(const G1Affine& b) const {
    if (is_identity()) return from_affine(b);
    Fp U2 = b.x * ZZ;
    Fp S2 = b.y * ZZZ;
    Fp P_ = U2 - X;
    Fp R_ = S2 - Y;
    if (P_.is_zero()) {
      if (R_.is_zero()) return dbl_affine(b);
      return identity();
    }
    Fp PP = P_.sqr();
    Fp PPP = P_ * PP;
    Fp Q_ = X * PP;
    G1 r;
    r.X = R_.sqr() - PPP - Q_.dbl();
    r.Y = R_ * (Q_ - r.X) - Y * PPP;
    r.ZZ = ZZ * PP;
    r.ZZZ = ZZZ * PPP;
    return r;
  }

  // full addition
  HD G1 add(const G1& b) const {
    if (is_identity()) return b;
    if (b.is_identity()) return *this;
    Fp U1 = X * b.ZZ;
    Fp U2 = b.X * ZZ;
    Fp S1 = Y * b.ZZZ;
    Fp S2 = b.Y * ZZZ;
    Fp P_ = U2 - U1;
    Fp R_ = S2 - S1;
    if (P_.is_zero()) {
      if (R_.is_zero()) return dbl();
      return identity();
    }
    Fp PP = P_.sqr();
    Fp PPP = P_ * PP;
    Fp Q_ = U1 * PP;
    G1 r;
    r.X = R_.sqr() - PPP - Q_.dbl();
    r.Y = R_ * (Q_ - r.X) - S1 * PPP;
    r.ZZ = ZZ * b.ZZ * PP;
    r.ZZZ = ZZZ * b.ZZZ * PPP;
    return r;
  }

  // k * this for a small public k (double-and-add, variable time)
  HD G1 mul_u32(uint32_t k) const {
    G1 acc = identity();
    for (int b = 31; b >= 0; --b) {
      acc = acc.dbl();
      if ((k >> b) & 1) acc = acc.add(*this);
    }
    return acc;
  }

  // affine normalisation: x = X/ZZ, y = Y/ZZZ (one Fp inversion of ZZZ:
  // 1/ZZ = ZZZ^-2 * ZZ^2 ... we invert both via a shared inverse of ZZ*ZZZ)
  HD bool to_affine(G1Affine* out) const {
    if (is_identity()) {
      out->x = Fp::zero();
      out->y = Fp::zero();
      return false;
    }
    Fp inv = (ZZ * ZZZ).inv();
    Fp izz = inv * ZZZ;
    Fp izzz = inv * ZZ;
    out->x = X * izz;
    out->y = Y * izzz;
    return true;
  }
};

}  // namespace plonk
