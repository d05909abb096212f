"""plonk_amd/csrc/permutation.hpp compiled for the host: the one-pass sigma mapping against
Permutation::compute_sigma_permutations restated literally (reference src/composer/permutation.rs:106-139:
a map witness -> list of wire positions in push order, each position sent to the next of its list) and
against the oracle Composer.  CPU-only."""
import ctypes
import random

import pytest

from oracle import plonk as O
from tests import circuits as C
from tests.test_field_host import build_host_lib


@pytest.fixture(scope="module")
def lib():
    return build_host_lib()


def reference_mapping(wires, constraints, n, witnesses):
    """permutation.rs:106-139 with a dict of lists (the reference's HashMap<Witness, Vec<WireData>>)."""
    lists = {w: [] for w in range(witnesses)}
    for i in range(constraints):
        for col in range(4):                                  # add_witnesses_to_map: a, b, c, d (permutation.rs:69-89)
            lists[wires[col][i]].append((col, i))
    sig = [[(col, i) for i in range(n)] for col in range(4)]
    for positions in lists.values():
        for k, (col, i) in enumerate(positions):
            sig[col][i] = positions[(k + 1) % len(positions)]
    return sig


def host_mapping(lib, wires, constraints, n, witnesses):
    cols = [(ctypes.c_uint32 * max(constraints, 1))(*w) for w in wires]
    out = (ctypes.c_uint32 * (4 * n))()
    rc = lib.h_sigma_mappings(*cols, ctypes.c_uint64(constraints), ctypes.c_uint64(n), ctypes.c_uint64(witnesses), out)
    if rc:
        return None
    return [[(out[col * n + i] >> 30, out[col * n + i] & 0x3FFFFFFF) for i in range(n)] for col in range(4)]


@pytest.mark.parametrize("constraints,n,witnesses,seed", [(1, 2, 1, 0), (5, 8, 3, 1), (8, 8, 40, 2), (100, 128, 17, 3),
                                                          (1000, 1024, 900, 4), (4096, 4096, 5, 5)])
def test_mapping_equals_the_reference_walk(lib, constraints, n, witnesses, seed):
    r = random.Random(seed)
    wires = [[r.randrange(witnesses) for _ in range(constraints)] for _ in range(4)]
    got = host_mapping(lib, wires, constraints, n, witnesses)
    assert got == reference_mapping(wires, constraints, n, witnesses)
    # a permutation of the 4n positions that fixes the padding rows
    flat = [p for col in got for p in col]
    assert sorted(flat) == [(col, i) for col in range(4) for i in range(n)]
    assert all(got[col][i] == (col, i) for col in range(4) for i in range(constraints, n))


def test_mapping_of_composed_circuits_equals_the_oracle_composer(lib):
    for build in (C.big_widget_circuit(300, seed=9), C.big_widget_circuit(1 << 11, seed=10)):
        comp = build()
        cols = C.circuit_columns(comp)
        m = len(comp.constraints)
        n = C.next_pow2(m)
        assert host_mapping(lib, cols["wires"], m, n, cols["witnesses"]) == comp.sigma_mappings(n)


def test_out_of_range_index_and_bad_sizes_are_rejected(lib):
    wires = [[0, 1, 2], [0, 0, 0], [1, 1, 1], [2, 2, 3]]
    assert host_mapping(lib, wires, 3, 4, 3) is None          # witness 3 of 3
    assert host_mapping(lib, wires, 3, 4, 4) is not None
    assert host_mapping(lib, wires, 3, 2, 4) is None          # more gates than the domain
