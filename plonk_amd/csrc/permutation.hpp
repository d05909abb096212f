// Copy constraints of a composed circuit -> the four sigma mappings, HIP-free (compiled for the host by
// prover.hip and by tests/csrc/host_arith.cpp).
//
// Permutation::compute_sigma_permutations (reference src/composer/permutation.rs:106-139): every witness keeps the
// list of wire positions it was attached to, in the order Composer::append_custom_gate_internal pushed them
// (src/composer.rs:119-167: gate by gate, a then b then c then d, permutation.rs:69-89); position k of a list maps to
// position k + 1, the last one back to the first, and a position no gate uses maps to itself.  The reference walks a
// HashMap of Vec<WireData>; the cycles do not depend on the iteration order, so one pass over the gates that
// remembers each witness's first and latest position produces the same mapping with two words per witness.
#pragma once
#include <cstdint>
#include <vector>

namespace plonk {

// A wire position is packed as (column << 30) | gate row; column 0..3 = Left, Right, Output, Fourth.
static constexpr uint32_t SIGMA_ROW_BITS = 30;
inline uint32_t sigma_pack(uint32_t col, uint64_t row) { return (col << SIGMA_ROW_BITS) | (uint32_t)row; }

// wires[col][i] = witness index on column `col` of gate i (i < constraints), every index < witnesses.
// out[col * n + i] = packed position that (col, i) maps to, for all i < n (rows >= constraints: identity).
// Returns false when an index is out of range or the sizes do not fit the packing.
inline bool sigma_mappings(const uint32_t* const wires[4], uint64_t constraints, uint64_t n, uint64_t witnesses, uint32_t* out) {
  if (constraints > n || n >= (1ull << SIGMA_ROW_BITS) || witnesses > 0xFFFFFFFFull) return false;
  static constexpr uint32_t NONE = 0xFFFFFFFFu;   // never a packed position: n < 2^30, so column 3's rows stop below 2^30 - 1
  for (uint32_t col = 0; col < 4; ++col)
    for (uint64_t i = 0; i < n; ++i) out[col * n + i] = sigma_pack(col, i);
  std::vector<uint32_t> first(witnesses, NONE), latest(witnesses, NONE);
  for (uint64_t i = 0; i < constraints; ++i)
    for (uint32_t col = 0; col < 4; ++col) {
      const uint32_t w = wires[col][i];
      if (w >= witnesses) return false;
      const uint32_t here = sigma_pack(col, i);
      const uint32_t prev = latest[w];
      if (prev == NONE) first[w] = here;
      else out[(uint64_t)(prev >> SIGMA_ROW_BITS) * n + (prev & ((1u << SIGMA_ROW_BITS) - 1))] = here;
      latest[w] = here;
    }
  for (uint64_t w = 0; w < witnesses; ++w) {
    const uint32_t last = latest[w];
    if (last != NONE) out[(uint64_t)(last >> SIGMA_ROW_BITS) * n + (last & ((1u << SIGMA_ROW_BITS) - 1))] = first[w];
  }
  return true;
}

}  // namespace plonk
