"""Pins the C restatement (oracle/c/oracle.c) to the big-int oracle, which in turn is
pinned to the reference's KAT digest.  CPU-only."""
import random

from oracle import bls12_381 as E
from oracle import cbind
from oracle.fft import EvaluationDomain

Q = E.Q


def fr_bytes(vals):
    return b"".join((v * E.FR_R % Q).to_bytes(32, "little") for v in vals)


def fr_vals(buf):
    return [int.from_bytes(buf[i:i + 32], "little") * E.FR_RINV % Q for i in range(0, len(buf), 32)]


def test_c_ntt_matches_bigint_oracle_all_modes():
    r = random.Random(1)
    for L, threads in ((3, 1), (8, 1), (12, 1), (12, 4), (13, 8)):
        n = 1 << L
        a = [r.randrange(Q) for _ in range(n)]
        d = EvaluationDomain(n)
        assert fr_vals(cbind.ntt_bytes(fr_bytes(a), L, False, False, n, threads)) == d.fft(a)
        assert fr_vals(cbind.ntt_bytes(fr_bytes(a), L, True, False, n, threads)) == d.ifft(a)
        il = n // 8 + 3
        assert fr_vals(cbind.ntt_bytes(fr_bytes(a[:il]), L, False, True, il, threads)) == d.coset_fft(a[:il])
        assert fr_vals(cbind.ntt_bytes(fr_bytes(a), L, True, True, n, threads)) == d.coset_ifft(a)


def test_short_inputs_take_the_closed_form_and_equal_the_full_transform():
    """round 6: forward transforms of at most two coefficients are written out (a0 + a1 s w^i) instead of run as butterflies
    over zeros — the reference's own closed-form tests (domain.rs:620-651: coset_fft([0, 1]) on 2^8 = 7 w^i) and the full
    transform of the same input padded with an explicit zero (three terms: the butterfly path) must agree"""
    r = random.Random(11)
    for L in (0, 1, 3, 8, 15):     # 2^15: more than one range of the parallel loops
        n = 1 << L
        d = EvaluationDomain(n)
        for coset in (False, True):
            for coeffs in ([], [r.randrange(Q)], [0, 1], [r.randrange(Q), r.randrange(Q)]):
                if len(coeffs) > n:
                    continue
                got = cbind.ntt_bytes(fr_bytes(coeffs), L, False, coset, len(coeffs), 4)
                if n >= 4:
                    assert got == cbind.ntt_bytes(fr_bytes(coeffs + [0] * (3 - len(coeffs))), L, False, coset, 3, 4)   # butterflies
                if L <= 8:
                    assert fr_vals(got) == (d.coset_fft(coeffs) if coset else d.fft(coeffs))
    d = EvaluationDomain(256)
    w = d.group_gen
    assert fr_vals(cbind.ntt_bytes(fr_bytes([0, 1]), 8, False, True, 2, 2)) == [7 * pow(w, i, Q) % Q for i in range(256)]   # domain.rs:620-636


def test_coset_scaling_ranges_match_the_serial_running_product():
    """distribute_powers and the inverse coset post-scale run as ranges seeded with g^start (round 6): 2^16 elements = four
    ranges; thread counts 1 and 5 must agree with each other and round-trip"""
    L = 16
    r = random.Random(12)
    a = fr_bytes([r.randrange(Q) for _ in range(1 << L)])
    one = cbind.ntt_bytes(a, L, False, True, 1 << L, 1)
    assert cbind.ntt_bytes(a, L, False, True, 1 << L, 5) == one
    assert cbind.ntt_bytes(one, L, True, True, 1 << L, 5) == a == cbind.ntt_bytes(one, L, True, True, 1 << L, 1)
    # a spot value against the definition: evaluation at 7 w^j
    vals = fr_vals(a)
    d = EvaluationDomain(1 << L)
    j = 40507
    x = 7 * pow(d.group_gen, j, Q) % Q
    acc = 0
    for c in reversed(vals):
        acc = (acc * x + c) % Q
    assert fr_vals(one[32 * j:32 * j + 32]) == [acc]


def test_parallel_fft_matches_serial_fft_for_large_domain():
    """reference domain.rs:570-618: 2^12, inputs i+1, thread counts 3 / 4 / 9 vs serial."""
    L = 12
    a = fr_bytes([i + 1 for i in range(1 << L)])
    serial = cbind.ntt_bytes(a, L, False, False, 1 << L, 1)
    for threads in (3, 4, 9):
        assert cbind.ntt_bytes(a, L, False, False, 1 << L, threads) == serial
    assert cbind.ntt_bytes(serial, L, True, False, 1 << L, 4) == a


def test_c_msm_matches_bigint_oracle():
    r = random.Random(2)
    pts = [E.g1_mul(E.G1_GEN, r.randrange(1, Q)) for _ in range(80)]
    raw = b"".join(E.g1_to_raw96(p) for p in pts)
    for m in (1, 5, 31, 32, 80):
        sc = [r.randrange(Q) for _ in range(m)]
        sc[0] = Q - 1
        out = cbind.msm_bytes(raw, fr_bytes(sc), m, 4)
        assert out[96] == 0 and E.g1_from_raw96(out[:96]) == E.msm_naive(pts, sc)
    assert cbind.msm_bytes(raw, fr_bytes([0] * 10), 10)[96] == 1
    assert cbind.msm_bytes(raw, b"", 0)[96] == 1
    rep = b"".join(E.g1_to_raw96(E.G1_GEN) for _ in range(3))
    out = cbind.msm_bytes(rep, fr_bytes([1, 1, 1]), 3)
    assert E.g1_from_raw96(out[:96]) == E.g1_mul(E.G1_GEN, 3)
