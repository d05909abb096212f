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
