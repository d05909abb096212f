"""CPU check of the NTT kernel's index arithmetic model (tests/ntt_model.py mirrors
plonk_amd/csrc/ntt.hip) against the oracle restatement of domain.rs."""
import random

import tests.ntt_model as M
from oracle.bls12_381 import Q
from oracle.fft import EvaluationDomain


def _check(L, radices=None):
    r = random.Random(L)
    N = 1 << L
    a = [r.randrange(Q) for _ in range(N)]
    d = EvaluationDomain(N)
    assert M.ntt_model(a, L, radices=radices) == d.fft(a)
    assert M.ntt_model(a, L, inverse=True, radices=radices) == d.ifft(a)
    il = N // 8 + 3
    assert M.ntt_model(a[:il], L, coset=True, in_len=il, radices=radices) == d.coset_fft(a[:il])
    assert M.ntt_model(a, L, inverse=True, coset=True, radices=radices) == d.coset_ifft(a)


def test_single_and_two_pass_plans():
    for L in (3, 6, 11, 12):
        _check(L)


def test_three_pass_decomposition_small_tile():
    old = (M.TILE_LOG, M.THREADS, M.ELOG)
    try:
        M.set_geometry(8, 32, 3)
        _check(13, [5, 4, 4])
        _check(14, [5, 5, 4])
    finally:
        M.set_geometry(*old)


def test_four_elements_per_lane_geometry():
    """ELOG = 2 (ntt.hip, PLONK_NTT_ELOG=2): radix-4 register rounds over 1024-element tiles — the kernel's own geometry on
    two-pass plans, a scaled-down one (same thread / element split) on three-pass plans."""
    old = (M.TILE_LOG, M.THREADS, M.ELOG)
    try:
        M.set_geometry(10, 256, 2)
        for L in (11, 12):
            _check(L)
        _check(13, [8, 5])                            # radix 2^8: four rounds of 2 bits
        _check(12, [7, 5])                            # radix 2^7: rounds of 2, 2, 2, 1 bits
        M.set_geometry(7, 32, 2)
        _check(13, [5, 4, 4])
        _check(14, [5, 5, 4])
    finally:
        M.set_geometry(*old)


def test_plan_radices_within_kernel_limits():
    for L in range(11, 28):
        p = M.plan(L)
        assert sum(p) == L and all(5 <= r <= 9 for r in p), (L, p)
        C = 1 << (M.TILE_LOG - p[0])
        assert C <= 1 << p[-1]                       # pass A tile never straddles a hi block
        if len(p) == 3:
            assert 1 << (M.TILE_LOG - p[1]) <= 1 << p[0]
