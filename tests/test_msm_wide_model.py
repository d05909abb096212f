"""The bucket-layout model of tests/msm_wide_model.py against the oracle: today's c = 16 layout and the planned c = 20 one
(DESIGN.md §7 item 0), plus two odd widths that exercise the generic split."""
import random

import pytest

from msm_wide_model import entries, msm_model, row_col_split, signed_digits, windows_for
from oracle import bls12_381 as E


@pytest.mark.parametrize("c", [16, 20, 11, 13])
def test_digits_cover_every_scalar_and_stay_in_range(c):
    r = random.Random(c)
    edge = [0, 1, (1 << (c - 1)) - 1, 1 << (c - 1), (1 << (c - 1)) + 1, (1 << c) - 1, 1 << c, E.Q - 1, E.Q - 2,
            (1 << 255) - 1, int("8" + "0" * 63, 16) - 1, sum(1 << (c * w + c - 1) for w in range(windows_for(c) - 1))]
    for s in edge + [r.randrange(E.Q) for _ in range(300)]:
        d = signed_digits(s, c)
        assert len(d) == windows_for(c)
        assert all(-(1 << (c - 1)) <= v <= (1 << (c - 1)) for v in d)
    assert windows_for(16) == 16 and windows_for(20) == 13
    assert row_col_split(16) == (256, 128) and row_col_split(20) == (512, 1024)


@pytest.mark.parametrize("c", [16, 20, 11])
def test_model_equals_the_oracle_msm(c):
    r = random.Random(100 + c)
    pts = [E.g1_mul(E.G1_GEN, r.randrange(1, E.Q)) for _ in range(12)]
    cases = [[r.randrange(E.Q) for _ in range(12)],
             [0, 1, 2, E.Q - 1, (1 << (c - 1)), (1 << (c - 1)) + 1, (1 << c) - 1, 1 << c, 5, 5, 5, E.Q - 5],   # borrow rule, equal digits
             [r.randrange(4) for _ in range(12)]]                                                                 # witness-like: one window only
    for sc in cases:
        want = E.msm_naive(pts, sc)
        assert msm_model(pts, sc, c) == want
        assert msm_model(pts, sc, c, order_by_size=False) == want
    assert msm_model(pts, [0] * 12, c) is None
    # zero digits never become entries; every other digit does, once
    sc = cases[2]
    assert len(entries(sc, c)) == sum(1 for s in sc if s)


@pytest.mark.parametrize("ksl,mean", [(32, 512), (64, 512), (32, 32), (8, 20)])
def test_length_ordered_lanes_are_a_permutation_of_the_slices_and_waste_nothing(ksl, mean):
    from msm_wide_model import idle_fraction, slice_order
    r = random.Random(ksl * 1000 + mean)
    counts = [max(0, int(r.gauss(mean, mean ** 0.5))) for _ in range(4096)] + [0, 1, ksl, ksl + 1, 5 * ksl - 1]
    lanes, slice_off = slice_order(counts, ksl)
    # every slice exactly once, its partial-sum slot unchanged, its entries inside its bucket
    slots = sorted(slice_off[b] + q for b, q, _, _ in lanes)
    assert slots == list(range(slice_off[-1]))
    covered = [0] * len(counts)
    for b, q, first, length in lanes:
        assert first == q * ksl and 0 < length <= ksl and first + length <= counts[b]
        covered[b] += length
    assert covered == counts
    # full slices first, then partial ones by decreasing length
    lengths = [l for _, _, _, l in lanes]
    assert lengths == sorted(lengths, reverse=True)
    # a wave of equal lanes idles (almost) never; bucket order idles by the partial slice of every bucket
    in_bucket_order = [l for _, l in sorted(((slice_off[b] + q), l) for b, q, _, l in lanes)]
    assert idle_fraction(lengths) < 0.01
    assert idle_fraction(in_bucket_order) > 2.5 * idle_fraction(lengths)


# ---- bit-position tables (round 3) ------------------------------------------------------------------------------
def _edge_scalars(r):
    return [0, 1, 2, 3, 4, 0xffff, 0x10000, 0x10001, 0x1ffff, 0x20000, E.Q - 1, E.Q - 2, (E.Q - 1) // 2, (1 << 254) + 1,
            (1 << 254) - 1, int("5" * 63, 16), int("a" * 62, 16), int("f" * 60, 16) << 8, sum(1 << (17 * k + 16) for k in range(14)),
            (1 << 239) - 1, ((1 << 16) - 1) << 238, 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000000] + \
           [r.randrange(E.Q) for _ in range(400)] + [r.randrange(1 << r.randrange(1, 255)) for _ in range(200)]


def test_bitpos_digits_are_a_width_17_naf():
    from msm_wide_model import bitpos_digits
    r = random.Random(17)
    total = cnt = 0
    for s in _edge_scalars(r):
        dg = bitpos_digits(s)
        assert len(dg) <= 16
        assert all(d & 1 and abs(d) < (1 << 16) for _, d in dg)
        assert all(0 <= p <= 255 for p, _ in dg)
        assert all(q - p >= 17 for (p, _), (q, _) in zip(dg[:-2], dg[1:-1]))      # the last two digits share what remains
        assert all(q - p >= 9 for (p, _), (q, _) in zip(dg, dg[1:]))
    for _ in range(2000):
        total += len(bitpos_digits(r.randrange(E.Q)))
        cnt += 1
    assert 14.4 < total / cnt < 14.9          # ~ 254.9 / 18 + 1/2 = 14.7 additions per scalar instead of 16


def test_bitpos_model_equals_the_oracle_msm():
    from msm_wide_model import bitpos_msm_model
    r = random.Random(1717)
    pts = [E.g1_mul(E.G1_GEN, r.randrange(1, E.Q)) for _ in range(10)]
    cases = [[r.randrange(E.Q) for _ in range(10)],
             [0, 1, 2, E.Q - 1, 0xffff, 0x10001, 5, 5, E.Q - 5, 3],
             [r.randrange(4) for _ in range(10)]]
    for sc in cases:
        assert bitpos_msm_model(pts, sc) == E.msm_naive(pts, sc)
    assert bitpos_msm_model(pts, [0] * 10) is None


def test_product_recoding_matches_the_models():
    """msm_recode.cuh compiled for the host (the code msm_hist / msm_partition run) against both models"""
    import ctypes

    from msm_wide_model import bitpos_digits, signed_digits
    from test_field_host import build_host_lib
    lib = build_host_lib()
    r = random.Random(99)
    out = (ctypes.c_uint32 * 64)()
    for s in _edge_scalars(r):
        limbs = (ctypes.c_uint32 * 8)(*[(s >> (32 * i)) & 0xFFFFFFFF for i in range(8)])
        n = lib.h_msm_recode(limbs, 1, out)
        got = [(out[4 * j + 1], (2 * out[4 * j + 2] + 1) * (-1 if out[4 * j + 3] else 1)) for j in range(n)]
        assert [out[4 * j] for j in range(n)] == list(range(n))          # slots are consecutive
        assert got == bitpos_digits(s), hex(s)
        n2 = lib.h_msm_recode(limbs, 2, out)                                # the LDS-parked form the kernels use
        assert [(out[4 * j + 1], (2 * out[4 * j + 2] + 1) * (-1 if out[4 * j + 3] else 1)) for j in range(n2)] == got
        n21 = lib.h_msm_recode(limbs, 21, out)                              # width-21 digits of the 2^19-bucket variant
        assert [(out[4 * j + 1], (2 * out[4 * j + 2] + 1) * (-1 if out[4 * j + 3] else 1)) for j in range(n21)] == bitpos_digits(s, 21)
        assert n21 <= 13 and all(out[4 * j + 2] < (1 << 19) for j in range(n21))
        n = lib.h_msm_recode(limbs, 0, out)
        got = {out[4 * j + 1]: (out[4 * j + 2] + 1) * (-1 if out[4 * j + 3] else 1) for j in range(n)}
        want = {w: d for w, d in enumerate(signed_digits(s, 16)) if d}
        assert got == want, hex(s)
        assert all(out[4 * j] == out[4 * j + 1] for j in range(n))       # window recoding: slot = row


def test_top_digits_are_not_skewed():
    """the last digit alone would be small with probability ~ 1 / value (bucket 0 would hold ~ 1 % of ALL entries); sharing
    the remaining bits between the last two digits keeps every bucket within a small multiple of the mean"""
    from collections import Counter

    from msm_wide_model import bitpos_digits
    r = random.Random(2121)
    for w, nb in ((21, 1 << 19), (17, 1 << 15)):
        cnt = Counter()
        total = 0
        n = 6000
        for _ in range(n):
            for _, d in bitpos_digits(r.randrange(E.Q), w):
                cnt[abs(d) >> 1] += 1
                total += 1
        # without the sharing bucket 0 alone receives ~ n * 2 / w entries
        assert max(cnt.values()) < max(12, 40 * total / nb) and cnt[0] < n / 50
        assert all(b < nb for b in cnt)


# ---- half-density tables (round 4) -------------------------------------------------------------------------------
def test_even_position_digits():
    from msm_wide_model import even_digits
    r = random.Random(2020)
    for w, most in ((20, 14), (16, 16)):
        for s in _edge_scalars(r):
            dg = even_digits(s, w)
            assert len(dg) <= most
            assert all(d % 4 != 0 and 1 <= abs(d) <= (1 << (w - 1)) for _, d in dg)
            assert all(0 <= row <= 127 for row, _ in dg)                       # 128 table rows: positions 0, 2, .. 254
            assert all(2 * (q - p) >= w for (p, _), (q, _) in zip(dg[:-2], dg[1:-1]))
        tot = sum(len(even_digits(r.randrange(E.Q), w)) for _ in range(2000)) / 2000
        assert abs(tot - (254.9 / (w + 2 / 3) + 0.5)) < 0.25, tot                 # 12.8 for w = 20, 15.8 for w = 16


def test_even_position_model_equals_the_oracle_msm():
    from msm_wide_model import even_msm_model
    r = random.Random(404)
    pts = [E.g1_mul(E.G1_GEN, r.randrange(1, E.Q)) for _ in range(8)]
    for sc in ([r.randrange(E.Q) for _ in range(8)], [0, 1, 2, 3, 4, E.Q - 1, 5, E.Q - 4]):
        assert even_msm_model(pts, sc, 16) == E.msm_naive(pts, sc)
    assert even_msm_model(pts, [r.randrange(E.Q) for _ in range(8)][:8], 20) is not None


def test_product_even_recoding_matches_the_model():
    import ctypes

    from msm_wide_model import even_digits
    from test_field_host import build_host_lib
    lib = build_host_lib()
    r = random.Random(77)
    out = (ctypes.c_uint32 * 64)()
    for s in _edge_scalars(r):
        limbs = (ctypes.c_uint32 * 8)(*[(s >> (32 * i)) & 0xFFFFFFFF for i in range(8)])
        for mode, w in ((120, 20), (116, 16)):
            n = lib.h_msm_recode(limbs, mode, out)
            got = [(out[4 * j + 1], (out[4 * j + 2] + 1) * (-1 if out[4 * j + 3] else 1)) for j in range(n)]
            assert got == even_digits(s, w), (hex(s), w)
            assert all(out[4 * j + 2] < (1 << (w - 1)) for j in range(n))
    # the last two digits share the remaining bits (even widths of 10 .. 20 bits): the lowest buckets hold ~35x the mean — 1.6x
    # the skew of the width-21 NAF (24x), far from the ~1 % of ALL entries in bucket 0 of an unshared top digit
    from collections import Counter
    cnt, total = Counter(), 0
    for _ in range(6000):
        for _, d in even_digits(r.randrange(E.Q), 20):
            cnt[abs(d) - 1] += 1
            total += 1
    assert max(cnt.values()) < 30 and cnt[0] < 6000 / 50
    assert sum(v for b, v in cnt.items() if b < 128) < 0.012 * total
