"""Executable model of the NEXT bucket layout of the MSM: wider windows with size-ordered buckets.

Not product code and not used by it — an index-arithmetic model in the style of tests/ntt_model.py, written before the
kernels so that every convention they must agree on is pinned against the oracle on the CPU:

  * signed window digits for ANY window width c (13 windows of 20 bits cover 260 >= 256 bits; today: 16 x 16), the carry
    rule and the range of the top window;
  * the entry word (table row = window, column = base index, sign) and the bucket an entry belongs to;
  * buckets handed to lanes in order of their entry count (a wave's lanes then run the same number of additions) —
    any order gives the same sums, the model checks that the permutation is one;
  * the weighted reduction sum_b b * B_b with b = COLS * h + l, l in [1, COLS]: row sums R_h, column sums C_l, the bit sums
    T_j / T'_j / C_COLS the device leaves to the host, and the host's Horner chain over them.

The generalisation of msm.hip's constants: c = 16: ROWS x COLS = 256 x 128 (msm_rowcol / msm_bits, 8 + 7 + 1 sums);
c = 20: 512 x 1024 (9 + 10 + 1 sums)."""
from oracle import bls12_381 as E


def windows_for(c: int) -> int:
    return -(-256 // c)          # ceil(256 / c): scalars are < 2^255, one spare bit absorbs the last carry


def signed_digits(s: int, c: int):
    """d_w in [-2^(c-1), 2^(c-1)] with sum d_w 2^(c w) = s; the same recoding rule as msm_hist / msm_partition for c = 16
    (a window value above 2^(c-1) borrows 2^c from the next window)."""
    W = windows_for(c)
    half, full = 1 << (c - 1), 1 << c
    out, carry = [], 0
    for w in range(W):
        v = ((s >> (c * w)) & (full - 1)) + carry
        carry = 0
        if v > half:
            v -= full
            carry = 1
        out.append(v)
    assert carry == 0, "the top window cannot overflow for s < 2^255"
    assert sum(d << (c * w) for w, d in enumerate(out)) == s
    return out


def entries(scalars, c: int):
    """(bucket index b - 1, negative?, table row, base index) for every non-zero digit; zero digits never leave the sort."""
    out = []
    for i, s in enumerate(scalars):
        for w, d in enumerate(signed_digits(s % E.Q, c)):
            if d:
                out.append((abs(d) - 1, d < 0, w, i))
    return out


def bucket_sums(points, ents, c: int, order_by_size: bool = True):
    """One lane per bucket; lanes in order of decreasing entry count.  Returns {bucket index: Jacobian sum}."""
    groups = {}
    for b, neg, w, i in ents:
        groups.setdefault(b, []).append((neg, w, i))
    lanes = sorted(groups, key=lambda b: (-len(groups[b]), b)) if order_by_size else sorted(groups)
    assert sorted(lanes) == sorted(groups)                      # a permutation of the non-empty buckets
    table = {}                                                  # T[w][i] = 2^(c w) P_i, built on demand
    sums = {}
    for b in lanes:
        acc = E.JAC_ID
        for neg, w, i in groups[b]:
            if (w, i) not in table:
                table[(w, i)] = E.to_jac(E.g1_mul(points[i], 1 << (c * w)))
            p = table[(w, i)]
            acc = E.jac_add(acc, E.jac_neg(p) if neg else p)
        sums[b] = acc
    return sums


def row_col_split(c: int):
    """ROWS x COLS = 2^(c-1) buckets; the split msm.hip uses for c = 16 and the one planned for c = 20."""
    col_bits = {16: 7, 20: 10}.get(c, (c - 1 + 1) // 2)
    return 1 << (c - 1 - col_bits), 1 << col_bits


def bit_sums(sums, c: int):
    """What the device hands to the host: T_j (rows with bit j of h), T'_j (columns with bit j of their weight l < COLS), C_COLS."""
    ROWS, COLS = row_col_split(c)
    R = [E.JAC_ID] * ROWS
    C = [E.JAC_ID] * COLS                                       # C[l - 1]
    for b, v in sums.items():
        h, l0 = divmod(b, COLS)                                 # bucket weight b + 1 = COLS * h + (l0 + 1)
        R[h] = E.jac_add(R[h], v)
        C[l0] = E.jac_add(C[l0], v)
    rb, cb = ROWS.bit_length() - 1, COLS.bit_length() - 1
    T = []
    for j in range(rb):
        acc = E.JAC_ID
        for h in range(ROWS):
            if (h >> j) & 1:
                acc = E.jac_add(acc, R[h])
        T.append(acc)
    Tp = []
    for j in range(cb):
        acc = E.JAC_ID
        for l in range(1, COLS):
            if (l >> j) & 1:
                acc = E.jac_add(acc, C[l - 1])
        Tp.append(acc)
    return T, Tp, C[COLS - 1]


def host_finish(T, Tp, C_top):
    """W = sum_j 2^j T'_j + COLS * C_COLS + COLS * sum_j 2^j T_j as ONE Horner chain (hostg1.hpp finish_bit_sums for c = 16)."""
    cb = len(Tp)
    U = list(Tp) + [E.jac_add(C_top, T[0])] + list(T[1:])       # weights 2^0 .. 2^(cb - 1), 2^cb, 2^(cb + 1) ...
    acc = U[-1]
    for u in reversed(U[:-1]):
        acc = E.jac_add(E.jac_double(acc), u)
    assert len(U) == cb + len(T)
    return E.to_affine(acc)


def msm_model(points, scalars, c: int, order_by_size: bool = True):
    ents = entries(scalars, c)
    return host_finish(*bit_sums(bucket_sums(points, ents, c, order_by_size), c))


# ---- lanes in order of slice length -------------------------------------------------------
def slice_order(counts, ksl: int):
    """The lane -> slice map msm_accumulate would use when slices are handed out by LENGTH instead of in bucket order.

    counts[b] = entries of bucket b.  Bucket b has ceil(counts[b] / ksl) slices; its partial sums keep their slots
    slice_off[b] + q (bucket order), so msm_bucket_sum and the heavy path do not change.  Lanes: first every FULL slice
    (ksl entries) — lane s < F belongs to the bucket found by a search in full_off, the exclusive scan of
    floor(counts / ksl) — then the one partial slice of every bucket whose count is not a multiple of ksl, in order of
    decreasing length (a counting sort over the ksl - 1 possible lengths, ties by bucket index).
    Returns (lanes, slice_off) with lanes[s] = (bucket, q, first entry offset inside the bucket, length)."""
    full = [c // ksl for c in counts]
    rem = [c % ksl for c in counts]
    slice_off, run = [], 0
    for b in range(len(counts)):
        slice_off.append(run)
        run += full[b] + (1 if rem[b] else 0)
    slice_off.append(run)
    lanes = []
    for b in range(len(counts)):                                   # lane s < F: bucket = upper_bound(full_off, s) - 1
        for q in range(full[b]):
            lanes.append((b, q, q * ksl, ksl))
    by_len = [[] for _ in range(ksl)]                              # counting sort of the partial slices by length
    for b in range(len(counts)):
        if rem[b]:
            by_len[rem[b]].append(b)
    for length in range(ksl - 1, 0, -1):
        for b in by_len[length]:
            lanes.append((b, full[b], full[b] * ksl, length))
    return lanes, slice_off


def idle_fraction(lengths, wave: int = 64):
    """Lane-steps a wave-synchronous machine issues beyond the useful ones: every wave runs as long as its longest lane."""
    issued = sum(max(lengths[k:k + wave]) * wave for k in range(0, len(lengths), wave))
    return issued / max(sum(lengths), 1) - 1


# ---- bit-position tables (round 3): one table row per BIT, width-17 non-adjacent form --------------------------------
NAF_W = 17


def bitpos_digits(s: int, w: int = NAF_W):
    """[(row, d)] with d odd, |d| < 2^(w-1), rows at least w apart, sum d * 2^row = s — msm_recode.cuh for_each_digit_bitpos.
    After a negative digit the remaining value carries 1: a digit starts where the bit differs from the carry."""
    out, p, carry = [], 0, 0
    while (s >> p) + carry:
        if ((s >> p) & 1) == carry:           # remaining value even
            p += 1
            continue
        rem = 256 - p                                     # when at most two digits are left they share the remaining bits (+ 1 of headroom)
        wd = (rem + 1) // 2 if w < rem <= 2 * w else w        # evenly (otherwise the last digit is small with probability ~ 1 / value)
        v = ((s >> p) & ((1 << wd) - 1)) + carry
        d = v - (1 << wd) if v >> (wd - 1) else v
        carry = 1 if d < 0 else 0
        out.append((p, d))
        p += wd
    assert sum(d << p for p, d in out) == s
    return out


def bitpos_entries(scalars, w: int = NAF_W):
    """(bucket = |d| >> 1, negative?, table row = bit position, base index); weight of a bucket = 2 * bucket + 1"""
    return [(abs(d) >> 1, d < 0, p, i) for i, s in enumerate(scalars) for p, d in bitpos_digits(s % E.Q, w)]


def bitpos_msm_model(points, scalars, w: int = NAF_W):
    """The bucket sums of the bit-position entries through the UNCHANGED row / column / bit-sum reduction, which yields
    W = sum (b + 1) B_b; the total S = sum B_b is the 17th sum the device emits, and the host returns 2 W - S."""
    ents = bitpos_entries(scalars, w)
    groups = {}
    for b, neg, p, i in ents:
        groups.setdefault(b, []).append((neg, p, i))
    sums = {}
    for b, lst in groups.items():
        acc = E.JAC_ID
        for neg, p, i in lst:
            t = E.to_jac(E.g1_mul(points[i], 1 << p))              # table row p: 2^p P_i
            acc = E.jac_add(acc, E.jac_neg(t) if neg else t)
        sums[b] = acc
    c = w - 1                                                      # 2^(w - 2) buckets = the c = w - 1 window layout
    T, Tp, C_top = bit_sums(sums, c)
    W = host_finish(T, Tp, C_top)
    S = E.JAC_ID
    for v in sums.values():
        S = E.jac_add(S, v)
    Wj = E.to_jac(W) if W is not None else E.JAC_ID
    return E.to_affine(E.jac_add(E.jac_double(Wj), E.jac_neg(S)))


# ---- half-density tables (round 4): a table row for every EVEN bit position, digits that start at even positions only ----
def even_digits(s: int, w: int = 20):
    """[(row, d)] with row = position / 2, position even, d != 0 mod 4, |d| <= 2^(w-1), sum d * 4^row = s — msm_recode.cuh
    for_each_digit_even.  A digit is taken where the remaining value (s >> p) + carry is not a multiple of 4."""
    assert w % 2 == 0
    out, p, carry = [], 0, 0
    while (s >> p) + carry:
        if ((s >> p) + carry) % 4 == 0:
            p += 2
            continue
        rem = 256 - p
        wd = ((rem // 2 + 1) & ~1) if w < rem <= 2 * w else w     # the last two digits share the remaining bits (even widths)
        v = ((s >> p) & ((1 << wd) - 1)) + carry
        d = v - (1 << wd) if v > (1 << (wd - 1)) else v
        carry = 1 if d < 0 else 0
        out.append((p >> 1, d))
        p += wd
    assert sum(d << (2 * r) for r, d in out) == s
    return out


def even_msm_model(points, scalars, w: int = 20):
    """bucket = |d| - 1 with weight bucket + 1 (the window convention): the entries go through the unchanged reduction, no 2 W - S"""
    sums = {}
    for i, s in enumerate(scalars):
        for r, d in even_digits(s % E.Q, w):
            t = E.to_jac(E.g1_mul(points[i], 1 << (2 * r)))        # table row r: 4^r P_i
            b = abs(d) - 1
            sums[b] = E.jac_add(sums.get(b, E.JAC_ID), E.jac_neg(t) if d < 0 else t)
    return host_finish(*bit_sums(sums, w))
