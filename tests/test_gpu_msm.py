"""GPU parity: plonk_msm / CommitKey::commit (HIP, through the C-ABI) vs the oracle's
definition of msm_variable_base (reference src/commitment_scheme/kzg10/key.rs:376-388).
Bit-exact on the affine coordinates (and therefore on the 48-byte compressed form)."""
import random

import pytest

from oracle import bls12_381 as E

pytestmark = pytest.mark.gpu
Q = E.Q


@pytest.fixture(scope="module")
def ctx():
    import plonk_amd
    c = plonk_amd.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def srs300():
    r = random.Random(11)
    tau, g = r.randrange(1, Q), r.randrange(1, Q)
    base = E.g1_mul(E.G1_GEN, g)
    pts, p = [], 1
    for _ in range(300):
        pts.append(E.g1_mul(base, p))
        p = p * tau % Q
    return pts


def test_basic_sizes(ctx, srs300):
    r = random.Random(1)
    ctx.srs_load(srs300)
    for m in (1, 2, 3, 31, 32, 33, 64, 300):
        sc = [r.randrange(Q) for _ in range(m)]
        assert ctx.msm(sc) == E.msm_pippenger(srs300, sc), m


def test_basic_plan_is_the_variant_that_was_asked_for(ctx, srs300):
    """What the library reports it ran (plonk_ctx_last_msm) against what the switches of this process ask for:
    tests/test_gpu_msm_variants.py passes the expectation of its variant as JSON in PLONK_TEST_EXPECT_PLAN, so a switch
    that is silently ignored fails here instead of re-testing the default kernels.  Without the variable: the defaults of a
    300-point key (window rows, 2^15 buckets, 4-entry slices, unordered lanes, a quad per bucket sum)."""
    import json
    import os
    ctx.srs_load(srs300)
    sc = [random.Random(5).randrange(Q) for _ in range(300)]
    assert ctx.msm(sc) == E.msm_pippenger(srs300, sc)
    plan = ctx.last_msm()
    want = json.loads(os.environ.get("PLONK_TEST_EXPECT_PLAN") or
                      '{"table_rows": 16, "bucket_bits": 15, "digit_width": 16, "slice_entries": 4, "ordered_lanes": 0, '
                      '"flags": 0, "accumulate_kernel": "nb15::msm_accumulate_kernel"}')
    assert {k: plan[k] for k in want} == want, plan
    assert plan["terms"] == 300 and ctx.describe_msm(300) == plan     # the prediction is the same function
    assert ctx.table_rows() == plan["table_rows"]


def test_edge_scalars(ctx, srs300):
    """Digit-recoding boundaries (16-bit signed windows), 0, 1, q-1."""
    ctx.srs_load(srs300)
    edge = [0, 1, 2, Q - 1, Q - 2, 32767, 32768, 32769, 65535, 65536, 65537,
            (1 << 254) + 12345, (1 << 16) * 32768, (1 << 32) - 1, 0x8000800080008000, 0xFFFFFFFFFFFFFFFF,
            int("8000" * 15, 16), int("7fff" * 15, 16), int("ffff" * 15, 16) % Q]
    assert ctx.msm(edge) == E.msm_naive(srs300, edge)
    for s in edge:
        assert ctx.msm([s]) == (E.g1_mul(srs300[0], s) if s % Q else None)
    assert ctx.msm([0] * 50) is None
    assert ctx.msm([]) is None
    assert ctx.msm([1] * 300) == E.msm_naive(srs300, [1] * 300)       # one hot bucket
    assert ctx.msm([5, Q - 5]) == E.msm_naive(srs300, [5, Q - 5])


def test_msm_batch_entry_point(ctx, srs300):
    """plonk_msm_batch = Prover::commit_polynomials' 4-way fan-out (prover.rs:187-210) as one grouped
    launch over the shared commit key: 4 sets, then 6 (two groups), unequal lengths, an empty set
    and an all-zero set; every result equals the single-call commitment and the oracle's."""
    import plonk_amd
    from oracle import cbind
    r = random.Random(77)
    ctx.srs_load(srs300)
    raw_srs = b"".join(E.g1_to_raw96(p) for p in srs300)
    for lens in ((300, 300, 300, 300), (300, 1, 0, 17, 299, 64), (5,), ()):
        sets = [[r.randrange(Q) for _ in range(m)] for m in lens]
        if len(sets) > 3:
            sets[3] = [0] * len(sets[3])
        raw = [plonk_amd.fr_to_bytes_mont(s) for s in sets]
        got = ctx.msm_batch_bytes(raw)
        assert len(got) == len(sets)
        for k, s in enumerate(sets):
            assert got[k] == ctx.msm_bytes(raw[k], len(s)), (lens, k)
            assert got[k] == cbind.msm_bytes(raw_srs, raw[k], len(s)), (lens, k)
    with pytest.raises(plonk_amd.PolynomialDegreeTooLarge):
        ctx.msm_batch_bytes([plonk_amd.fr_to_bytes_mont([1] * 301)])


def test_repeated_bases_hit_doubling_branch(ctx):
    G2 = E.g1_mul(E.G1_GEN, 2)
    pts = [E.G1_GEN, E.G1_GEN, E.G1_GEN, G2, E.g1_mul(E.G1_GEN, Q - 1)]
    ctx.srs_load(pts)
    assert ctx.msm([1, 1, 1]) == E.g1_mul(E.G1_GEN, 3)
    assert ctx.msm([7, 7, 7, 7]) == E.g1_mul(E.G1_GEN, 35)
    assert ctx.msm([1, 0, 0, 0, 1]) is None                           # G + (-G)
    assert ctx.msm([3, 5, 0, 9, 2]) == E.g1_mul(E.G1_GEN, 3 + 5 + 18 - 2)


def test_commit_rejects_oversized_polynomial(ctx, srs300):
    """reference key.rs:816-824 test_commit_rejects_oversized_polynomial."""
    import plonk_amd
    ctx.srs_load(srs300[:10])
    with pytest.raises(plonk_amd.PolynomialDegreeTooLarge):
        ctx.commit([1] * 11)
    with pytest.raises(plonk_amd.PolynomialDegreeTooLarge):
        ctx.msm([1] * 11)
    assert ctx.commit([1] * 10 + [0, 0]) == E.msm_naive(srs300[:10], [1] * 10)   # trailing zeros trimmed


def test_msm_before_srs_load_errors():
    import plonk_amd
    c = plonk_amd.Context(0)
    with pytest.raises(plonk_amd.PlonkError):
        c.msm([1, 2, 3])
    c.close()


def test_aggregate_flatten_small_kat(ctx):
    """reference kzg10/proof.rs:120-158: 2G, 3G, 5G with v = 7."""
    G = E.G1_GEN
    ctx.srs_load([E.g1_mul(G, 2), E.g1_mul(G, 3), E.g1_mul(G, 5)])
    assert ctx.msm([1, 7, 49]) == E.g1_mul(G, 2 + 21 + 245)


def _gen_srs_dev(ctx, n, tau, g):
    buf = ctx.alloc(96 * n)
    ctx.srs_generate_dev(tau, g, n, buf.ptr)
    return buf


def test_generated_srs_matches_oracle_and_msm_4096(ctx):
    """PublicParameters::setup semantics (srs.rs:61-100): P_i = (g * tau^i) G."""
    import plonk_amd
    r = random.Random(3)
    n = 4096
    tau, g = r.randrange(1, Q), r.randrange(1, Q)
    buf = _gen_srs_dev(ctx, n, tau, g)
    raw = buf.download()
    pts = [plonk_amd.g1_from_raw97(raw[96 * i:96 * i + 96] + b"\0") for i in range(n)]
    for i in (0, 1, 2, 77, n - 1):
        assert pts[i] == E.g1_mul(E.G1_GEN, g * pow(tau, i, Q) % Q)
    ctx.srs_load_dev(buf.ptr, n)
    sc = [r.randrange(Q) for _ in range(n)]
    # sum_i s_i * g tau^i G == (g * sum_i s_i tau^i) G   — closed form, no big MSM oracle needed
    k = g * sum(s * pow(tau, i, Q) for i, s in enumerate(sc)) % Q
    assert ctx.msm(sc) == E.g1_mul(E.G1_GEN, k)
    assert ctx.msm(sc[:1000]) == E.msm_pippenger(pts[:1000], sc[:1000])
    buf.free()


@pytest.mark.parametrize("logm", [16, 20])
def test_full_size_closed_form(ctx, logm):
    """BASELINE sizes (2^16, 2^20 gates => m = n + 6): the SRS is [g tau^i]G, so the
    commitment of f is (g * f(tau)) G — a size-independent exact check."""
    r = random.Random(logm)
    n = (1 << logm) + 7
    tau, g = r.randrange(1, Q), r.randrange(1, Q)
    buf = _gen_srs_dev(ctx, n, tau, g)
    ctx.srs_load_dev(buf.ptr, n)
    buf.free()
    m = (1 << logm) + 6
    import plonk_amd
    sc = [r.randrange(Q) for _ in range(m)]
    acc, p = 0, 1
    for s in sc:
        acc = (acc + s * p) % Q
        p = p * tau % Q
    got = ctx.msm(sc)
    assert got == E.g1_mul(E.G1_GEN, g * acc % Q)
    assert len(plonk_amd.g1_compress(got)) == 48


def test_ordered_lanes_from_16_entry_slices_on(ctx):
    """Round 6: a 2^17-point MSM takes 16-entry slices with the lanes in order of slice length (until then only from
    32-entry slices on) — the layout msm_slices_kernel<true> now writes itself; the plan says so and the result is the
    closed form.  Skewed scalars in the same key: full slices, partial slices of every length and heavy buckets together."""
    if any(k in __import__("os").environ for k in ("PLONK_MSM_ORDER", "PLONK_MSM_KSL", "PLONK_MSM_TABLE", "PLONK_MSM_BUCKETS", "PLONK_MSM_ACC")):
        pytest.skip("the default plan is what this test pins")
    import json
    r = random.Random(1717)
    n = (1 << 17) + 7
    tau, g = r.randrange(1, Q), r.randrange(1, Q)
    buf = _gen_srs_dev(ctx, n, tau, g)
    ctx.srs_load_dev(buf.ptr, n)
    buf.free()
    m = (1 << 17) + 6
    for sc in ([r.randrange(Q) for _ in range(m)],
               [r.randrange(Q) if i % 5 else r.randrange(4) for i in range(m)]):       # a fifth of the scalars < 4: heavy low buckets
        acc, p = 0, 1
        for s in sc:
            acc = (acc + s * p) % Q
            p = p * tau % Q
        assert ctx.msm(sc) == E.g1_mul(E.G1_GEN, g * acc % Q)
        plan = ctx.last_msm()
        assert (plan["bucket_bits"], plan["slice_entries"], plan["ordered_lanes"], plan["accumulate_kernel"]) == \
            (15, 16, 1, "nb15::msm_accumulate_ordered_kernel"), json.dumps(plan)


def test_skewed_scalars_take_the_heavy_bucket_path(ctx):
    """Equal coefficients put every term of a window into ONE bucket (m/32 slices): the
    segmented heavy-bucket kernels must give the same group element; mixed with uniform
    scalars so that both bucket-sum kernels contribute to the same commitment."""
    r = random.Random(41)
    n = 20000
    tau, g = r.randrange(1, Q), r.randrange(1, Q)
    buf = _gen_srs_dev(ctx, n, tau, g)
    ctx.srs_load_dev(buf.ptr, n)
    buf.free()
    geo = [pow(tau, i, Q) for i in range(n)]
    for sc in ([0x1234567890ABCDEF1234567890ABCDEF % Q] * n,                       # one bucket per window
               [r.randrange(Q) if i % 3 else 7 for i in range(n)],                  # heavy + uniform buckets
               [Q - 1] * 9000):                                                     # all-negative digits
        k = g * sum(s * t for s, t in zip(sc, geo)) % Q
        assert ctx.msm(sc) == E.g1_mul(E.G1_GEN, k)


def test_small_scalars_take_the_chunked_bin_and_segmented_bucket_paths(ctx):
    """Witness-like scalars (bits, quads, small range accumulators): most digits are zero and dropped, the rest pile
    into a handful of buckets — one coarse bin of the sort far above 2^16 words (msm_big_hist / msm_big_scatter) and
    buckets of thousands of slices (msm_heavy_seg / msm_heavy_bucket).  Closed form on the [g tau^i] G key."""
    r = random.Random(43)
    n = 150000
    tau, g = r.randrange(1, Q), r.randrange(1, Q)
    buf = _gen_srs_dev(ctx, n, tau, g)
    ctx.srs_load_dev(buf.ptr, n)
    buf.free()
    geo, p = [], 1
    for _ in range(n):
        geo.append(p)
        p = p * tau % Q
    for sc in ([r.randrange(4) for _ in range(n)],                                    # 2-bit quads: buckets 0..2 only
               [r.randrange(4) if i % 2 else r.randrange(Q) for i in range(n)],       # the bench-like mix
               [(1 << 16) * r.randrange(1, 3) + r.randrange(2) for _ in range(n)],    # two windows, both concentrated
               [Q - 1 - r.randrange(3) for _ in range(n)]):                           # small negatives: every window, sign set
        k = g * sum(s * t for s, t in zip(sc, geo)) % Q
        assert ctx.msm(sc) == E.g1_mul(E.G1_GEN, k)


def test_lagrange_key_matches_the_definition_and_commits_like_the_monomial_key(ctx):
    """plonk_lagrange_key: out[i] = [L_i(tau)] G on the size-n domain, then [tau^n] G - G and [tau^(n+1)] G - [tau] G.
    With the key [g tau^j] G the points are known in closed form (L_i(tau) = w^i (tau^n - 1) / (n (tau - w^i))), and
    committing to evaluations over the Lagrange key must equal committing to the interpolated coefficients over
    the monomial key (what the prover relies on for the wire commitments)."""
    import plonk_amd
    from oracle.fft import EvaluationDomain
    r = random.Random(61)
    L = 6
    n = 1 << L
    tau, g = r.randrange(2, Q), r.randrange(1, Q)
    buf = _gen_srs_dev(ctx, n + 5, tau, g)
    ctx.srs_load_dev(buf.ptr, n + 5)
    buf.free()
    key = ctx.lagrange_key(L)
    assert len(key) == 96 * (n + 2)
    d = EvaluationDomain(n)
    w = d.group_gen
    tn = pow(tau, n, Q)
    for i in (0, 1, 5, n - 1):
        wi = pow(w, i, Q)
        li = wi * (tn - 1) % Q * pow(n * (tau - wi) % Q, -1, Q) % Q
        assert E.g1_from_raw96(key[96 * i:96 * i + 96]) == E.g1_mul(E.G1_GEN, g * li % Q), i
    assert E.g1_from_raw96(key[96 * n:96 * n + 96]) == E.g1_mul(E.G1_GEN, g * (tn - 1) % Q)
    assert E.g1_from_raw96(key[96 * (n + 1):96 * (n + 2)]) == E.g1_mul(E.G1_GEN, g * (tn * tau - tau) % Q)
    evals = [r.randrange(Q) for _ in range(n)]
    coeffs = d.ifft(evals)
    want = ctx.msm(coeffs)                                        # monomial key, coefficient form
    ctx.srs_load_bytes(key[:96 * n], n)                           # the Lagrange points as a commit key
    assert ctx.msm(evals) == want
    # a key that is too short for the domain is refused like an oversized polynomial
    ctx.srs_load_bytes(key[:96 * 10], 10)
    with pytest.raises(plonk_amd.PolynomialDegreeTooLarge):
        ctx.lagrange_key(L)
