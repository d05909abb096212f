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
