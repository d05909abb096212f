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
