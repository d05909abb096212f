"""GPU parity: plonk_ntt (HIP, through the C-ABI) vs the oracle restatement of
EvaluationDomain::{fft,ifft,coset_fft,coset_ifft} (reference src/fft/domain.rs:166-232).
Bit-exact: values are compared as canonical integers recovered from the Montgomery
limbs the library returns."""
import random

import pytest

from oracle.bls12_381 import GENERATOR, Q
from oracle.fft import EvaluationDomain

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import plonk_amd
    c = plonk_amd.Context(0)
    yield c
    c.close()


def _all_modes(ctx, a, L, in_len):
    d = EvaluationDomain(1 << L)
    assert ctx.ntt(a, L) == d.fft(a)
    assert ctx.ntt(a, L, inverse=True) == d.ifft(a)
    assert ctx.ntt(a[:in_len], L, coset=True) == d.coset_fft(a[:in_len])
    assert ctx.ntt(a, L, inverse=True, coset=True) == d.coset_ifft(a)


@pytest.mark.parametrize("L", [0, 1, 3, 6, 10])
def test_small_single_kernel(ctx, L):
    r = random.Random(100 + L)
    a = [r.randrange(Q) for _ in range(1 << L)]
    _all_modes(ctx, a, L, max(1, (1 << L) // 8 + 3) if L >= 4 else 1 << L)


@pytest.mark.parametrize("L", [11, 12, 13, 16])
def test_two_pass(ctx, L):
    r = random.Random(200 + L)
    a = [r.randrange(Q) for _ in range(1 << L)]
    _all_modes(ctx, a, L, (1 << L) // 8 + 3)


def test_reference_inputs_i_plus_1(ctx):
    """reference domain.rs:570-618: 2^12 domain, inputs i+1; ifft(fft(x)) == x."""
    L = 12
    a = [i + 1 for i in range(1 << L)]
    d = EvaluationDomain(1 << L)
    ev = ctx.ntt(a, L)
    assert ev == d.fft(a)
    assert ctx.ntt(ev, L, inverse=True) == a


def test_linear_coset_closed_form(ctx):
    """reference domain.rs:620-636: coset_fft([0,1]) on 2^8 equals 7*w^i; same on 2^16."""
    for L in (8, 16):
        d = EvaluationDomain(1 << L)
        ev = ctx.ntt([0, 1], L, coset=True)
        cur, exp = GENERATOR, []
        for _ in range(1 << L):
            exp.append(cur)
            cur = cur * d.group_gen % Q
        assert ev == exp


def test_batch_of_five_coset_ffts(ctx):
    """reference quotient_poly.rs:315-349: 5 polys on 2^12, values idx*n+i+1."""
    L = 12
    n = 1 << L
    d = EvaluationDomain(n)
    for idx in range(5):
        poly = [idx * n + i + 1 for i in range(n)]
        assert ctx.ntt(poly, L, coset=True) == d.coset_fft(poly)


def test_ntt_batch_entry_point_five_coset_ffts(ctx):
    """The same five polynomials (quotient_poly.rs:315-349) through ONE plonk_ntt_batch call — the
    replacement of compute_coset_evaluations' rayon fan-out (:139-157): results in input order, equal
    to the single-call transforms; also an inverse batch with unequal in_len and a batch of one."""
    import plonk_amd
    from oracle import cbind
    L = 12
    n = 1 << L
    polys = [plonk_amd.fr_to_bytes_mont([idx * n + i + 1 for i in range(n)]) for idx in range(5)]
    got = ctx.ntt_batch_bytes(polys, L, False, True)
    for idx in range(5):
        assert got[idx] == cbind.ntt_bytes(polys[idx], L, False, True, n), idx
    # the quotient-domain shape: 8n transform of n + 2 / n + 3 coefficients (in_len per element)
    L8 = 15
    lens = [n + 2, n + 3, n, 1, n + 2, 0, n + 2]
    short = [polys[i % 5][:32 * min(l, n)] + bytes(32 * max(0, l - n)) for i, l in enumerate(lens)]
    got = ctx.ntt_batch_bytes(short, L8, False, True, lens)
    for i, l in enumerate(lens):
        assert got[i] == cbind.ntt_bytes(short[i], L8, False, True, l), i
    back = ctx.ntt_batch_bytes(got[:3], L8, True, True)
    for i in range(3):
        assert back[i][:32 * lens[i]] == short[i][:32 * lens[i]] and not any(back[i][32 * lens[i]:])
    assert ctx.ntt_batch_bytes([polys[2]], L, True, False) == [cbind.ntt_bytes(polys[2], L, True, False, n)]
    assert ctx.ntt_batch_bytes([], L, False, False) == []


def test_truncates_longer_input(ctx):
    """Vec::resize truncation (domain.rs:174)."""
    L = 11
    r = random.Random(5)
    a = [r.randrange(Q) for _ in range((1 << L) + 100)]
    assert ctx.ntt(a, L) == EvaluationDomain(1 << L).fft(a[: 1 << L])


def test_rejects_bad_log_n(ctx):
    import plonk_amd
    with pytest.raises(plonk_amd.PlonkError):
        ctx.ntt_bytes(b"", 28, False, False, 0)


@pytest.mark.parametrize("L", [19, 20])
def test_three_pass_vs_oracle(ctx, L):
    """three-pass plans against the C restatement of best_fft (oracle/c, which tests/test_oracle_c.py pins to the big-int
    oracle and the reference's closed forms): the Python oracle takes 10-20 s per transform at these sizes, and this test runs
    three times per session (default kernels, PLONK_NTT_DIRECT=0, PLONK_NTT_ELOG=3).  Bytes in, bytes out."""
    import numpy as np
    from oracle import cbind
    N = 1 << L
    raw = np.random.default_rng(300 + L).integers(0, 256, size=(N, 32), dtype=np.uint8)
    raw[:, 31] &= 0x3F                                      # Montgomery limbs below 2^254 < q
    a = raw.tobytes()
    assert ctx.ntt_bytes(a, L, False, False, N) == cbind.ntt_bytes(a, L, False, False, N)
    il = N // 8 + 3
    got = ctx.ntt_bytes(a[:32 * il], L, False, True, il)
    assert got == cbind.ntt_bytes(a[:32 * il], L, False, True, il)
    back = ctx.ntt_bytes(got, L, True, True, N)
    assert back == cbind.ntt_bytes(got, L, True, True, N) and back[32 * il:] == bytes(32 * (N - il))
    # the inverse of the coset transform returns the coefficients in canonical Montgomery form: compare as field elements
    import plonk_amd
    assert plonk_amd.fr_from_bytes_mont(back[:32 * 64]) == plonk_amd.fr_from_bytes_mont(a[:32 * 64])


@pytest.mark.parametrize("L", [23])
def test_full_size_properties(ctx, L):
    """BASELINE quotient-domain size 8n = 2^23 (n = 2^20): size-independent checks.
    A sparse input has a closed-form spectrum: X[k] = sum_j x_j w^(j k)."""
    import plonk_amd
    r = random.Random(23)
    N = 1 << L
    d = EvaluationDomain(N)
    pos = [0, 1, 12345, N // 2 + 7, N - 1]
    val = [r.randrange(Q) for _ in pos]
    raw = bytearray(32 * N)
    mont = plonk_amd.fr_to_bytes_mont(val)
    for j, p in enumerate(pos):
        raw[32 * p:32 * p + 32] = mont[32 * j:32 * j + 32]
    out = ctx.ntt_bytes(bytes(raw), L, False, False, N)
    ks = [0, 1, 2, 255, 256, 65537, N // 2, N - 1] + [r.randrange(N) for _ in range(40)]
    for k in ks:
        exp = sum(v * pow(d.group_gen, p * k, Q) for p, v in zip(pos, val)) % Q
        assert plonk_amd.fr_from_bytes_mont(out[32 * k:32 * k + 32])[0] == exp
    back = ctx.ntt_bytes(out, L, True, False, N)
    assert back == bytes(raw)                          # ifft(fft(x)) == x, bit for bit
    # coset round trip at full size
    cf = ctx.ntt_bytes(bytes(raw), L, False, True, N)
    assert ctx.ntt_bytes(cf, L, True, True, N) == bytes(raw)
    k = 4242
    exp = sum(v * pow(GENERATOR, p, Q) * pow(d.group_gen, p * k, Q) for p, v in zip(pos, val)) % Q
    assert plonk_amd.fr_from_bytes_mont(cf[32 * k:32 * k + 32])[0] == exp


def test_concurrent_callers_share_one_context(ctx):
    """The reference calls the transform from rayon workers concurrently (prover.rs:174-177,
    quotient_poly.rs:150-152) and commits 4-way in parallel (prover.rs:194-197): every C-ABI
    entry point takes the per-context mutex, so threads may share a context.  8 threads mix
    NTTs of different sizes/modes with MSMs; every result must equal the sequential one."""
    import threading

    from oracle import bls12_381 as E
    r = random.Random(77)
    pts = [E.g1_mul(E.G1_GEN, r.randrange(1, Q)) for _ in range(48)]
    ctx.srs_load(pts)
    jobs = []
    for k in range(8):
        L = (9, 11, 12, 13)[k % 4]
        a = [r.randrange(Q) for _ in range(1 << L)]
        sc = [r.randrange(Q) for _ in range(48)]
        jobs.append((L, a, sc, bool(k & 1), bool(k & 2)))
    expected = [(ctx.ntt(a, L, inverse=inv, coset=cos), ctx.msm(sc)) for (L, a, sc, inv, cos) in jobs]
    got = [None] * len(jobs)
    errs = []

    def work(i):
        try:
            L, a, sc, inv, cos = jobs[i]
            for _ in range(3):
                got[i] = (ctx.ntt(a, L, inverse=inv, coset=cos), ctx.msm(sc))
        except Exception as e:   # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=work, args=(i,)) for i in range(len(jobs))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    assert got == expected
    d = EvaluationDomain(1 << jobs[0][0])
    assert expected[0][0] == d.fft(jobs[0][1])


@pytest.mark.parametrize("L", [5, 12, 19])
def test_empty_input_is_the_zero_polynomial(ctx, L):
    """Vec::resize pads an empty coefficient vector with zeros (domain.rs:174): every variant returns zeros (ntt.hip
    answers in_len == 0 with a memset, the pass kernels always read element 0)."""
    z = [0] * (1 << L)
    assert ctx.ntt([], L) == z
    assert ctx.ntt([], L, coset=True) == z


def test_two_level_twiddle_fallback_matches_the_oracle():
    """PLONK_NTT_DIRECT=0: the inter-pass twiddles as TWLO x TWHI products (the path taken above 2^25 or when memory is
    short) — the same transforms, in a child process because the switch is read once."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_ntt.py", "-x", "-q", "-m", "gpu", "-k",
                        "two_pass or three_pass or batch_of_five"], cwd=root, env=dict(os.environ, PLONK_NTT_DIRECT="0"),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout


@pytest.mark.parametrize("elog", ["3"])
def test_elements_per_lane_variants_match_the_oracle(elog):
    """PLONK_NTT_ELOG=3: the pass kernels with 8 elements per lane (radix-8 register rounds over 2048-element tiles, two waves
    per SIMD).  Every other test of this file runs the default, 4 elements per lane (radix-4 rounds over 1024-element tiles,
    four waves); the 8-element kernels otherwise only run inside provers of more than 2^18 gates (side-stream transforms).
    Same transforms, in a child process because the switch is read once.  ("2" forces the default explicitly: add it to the
    list when the default changes.)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_ntt.py", "-x", "-q", "-m", "gpu", "-k",
                        "two_pass or batch_of_five or full_size_properties or empty_input or (three_pass and 19)"], cwd=root,
                       env=dict(os.environ, PLONK_NTT_ELOG=elog), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout
