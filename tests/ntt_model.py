"""Executable model of plonk_amd/csrc/ntt.hip's index arithmetic (test code).

The HIP kernel cannot run in the CPU-only container, so its decomposition
(pass plan, tile/thread -> element maps, DIF twiddle indices, inter-pass twiddle
exponents, transposed store) is mirrored here 1:1 with Python ints and checked
against oracle.fft.  Keep the two in sync: every formula below has the same
name in ntt.hip.
"""
from oracle.bls12_381 import GENERATOR, Q, ROOT_OF_UNITY, fr_inv

TILE_LOG = 11      # 2048 elements per workgroup tile (ntt.hip: NTT_THREAD_BITS + ELOG)
THREADS = 256
ELOG = 3           # log2 of the elements per thread: 3 (radix-8 register rounds) or 2 (round 4: radix-4 rounds, 1024-element tiles)


def set_geometry(tile_log, threads, elog):
    """Switch the model between the kernel's two compiled geometries (or a scaled-down one for fast tests)."""
    global TILE_LOG, THREADS, ELOG
    assert threads << elog == 1 << tile_log
    TILE_LOG, THREADS, ELOG = tile_log, threads, elog


def plan(L):
    """Pass radices (log2).  <= 10: single small kernel; <= 18: two; <= 27: three."""
    if L <= 10:
        return [L]
    if L <= 18:
        r1 = (L + 1) // 2
        return [r1, L - r1]
    r1 = (L + 2) // 3
    r2 = (L - r1 + 1) // 2
    return [r1, r2, L - r1 - r2]


def omega(L, inverse):
    g = pow(ROOT_OF_UNITY, 1 << (32 - L), Q)
    return fr_inv(g) if inverse else g


def elem_index(t, e, pos, rb):
    """LDS linear index (row * C + col) of element e of thread t in a round whose
    active row bits sit at idx bits [pos, pos + rb)."""
    j = e & ((1 << rb) - 1)
    ge = e >> rb
    rest = (ge << (TILE_LOG - ELOG)) | t
    lo = rest & ((1 << pos) - 1)
    hi = rest >> pos
    return (hi << (pos + rb)) | (j << pos) | lo


def bitrev(x, bits):
    r = 0
    for _ in range(bits):
        r = (r << 1) | (x & 1)
        x >>= 1
    return r


def rounds(RLOG):
    out, hi = [], RLOG
    while hi > 0:
        lo = max(hi - ELOG, 0)
        out.append((lo, hi - lo))
        hi = lo
    return out


def tile_dif(vals, RLOG, w512):
    """In-tile DIF over the row bits of a [R][C] tile (vals indexed by idx =
    row * C + col), executed round by round / thread by thread exactly like the
    kernel.  Returns the tile with row position p holding output k = bitrev(p)."""
    CLOG = TILE_LOG - RLOG
    lds = list(vals)
    for (lo, rb) in rounds(RLOG):
        pos = CLOG + lo
        new = list(lds)
        for t in range(THREADS):
            E = 1 << ELOG
            idxs = [elem_index(t, e, pos, rb) for e in range(E)]
            v = [lds[i] for i in idxs]
            for lb in reversed(range(rb)):           # local bit, top first
                bitpos = lo + lb                      # global row bit of this stage
                for e in range(E):
                    if (e >> lb) & 1:
                        continue
                    if (e & ((1 << rb) - 1)) >> rb:   # never (kept for symmetry)
                        continue
                    e2 = e | (1 << lb)
                    row = idxs[e] >> CLOG
                    tw = (row & ((1 << bitpos) - 1)) << (8 - bitpos)
                    a, b = v[e], v[e2]
                    v[e] = (a + b) % Q
                    v[e2] = (a - b) * w512[tw] % Q
            for i, x in zip(idxs, v):
                new[i] = x
        lds = new
    return lds


def direct_index(e, tw_shr):
    """ntt.hip pass_twiddle: slot of w_N^e in a pass's direct table (pass A: tw_shr = 0, every e; pass B: the table
    holds the multiples of R1 = 2^tw_shr only, and e must be one of them)."""
    assert e & ((1 << tw_shr) - 1) == 0
    return e >> tw_shr


def ntt_model(a, L, inverse=False, coset=False, in_len=None, radices=None, direct=True):
    """Full transform as the GPU executes it (natural in, natural out).  direct: inter-pass twiddles read from the
    whole tables tw_a (w^e, e < N) / tw_b (w^(R1 j), j < N / R1) through direct_index, else computed from the exponent."""
    N = 1 << L
    in_len = N if in_len is None else in_len
    w = omega(L, inverse)
    w512 = [pow(omega(9, inverse), e, Q) for e in range(256)]
    n_inv = fr_inv(N)
    g = GENERATOR
    src = [(a[i] % Q if i < min(in_len, len(a)) else 0) for i in range(N)]
    if coset and not inverse:
        src = [src[i] * pow(g, i, Q) % Q for i in range(N)]          # fused into first load
    radices = radices or plan(L)
    if len(radices) == 1:
        # small kernel: bit-reversed load + DIT stages with w^(j * N / 2m)
        buf = [src[bitrev(i, L)] for i in range(N)]
        m = 1
        while m < N:
            for start in range(0, N, 2 * m):
                for j in range(m):
                    tw = pow(w, j * (N // (2 * m)), Q)
                    t_ = buf[start + m + j] * tw % Q
                    l_ = buf[start + j]
                    buf[start + m + j] = (l_ - t_) % Q
                    buf[start + j] = (l_ + t_) % Q
            m *= 2
        out = buf
    else:
        P = len(radices)
        r1 = radices[0]
        r3 = radices[-1]
        R1, R3 = 1 << r1, 1 << r3
        R2 = 1 << radices[1] if P == 3 else 1
        # ---- pass A: rows i1 (stride S), transposed out, twiddle w^(k * cg)
        S = N >> r1
        CLOG = TILE_LOG - r1
        C = 1 << CLOG
        tw_a = tw_b = None
        if direct:
            tw_a = [1] * N
            for e in range(1, N):
                tw_a[e] = tw_a[e - 1] * w % Q
            if P == 3:
                tw_b = tw_a[::R1]                     # w^(R1 j), j < N / R1
        dst = [None] * N
        for blk in range(S // C):
            cg0 = blk * C
            tile = [src[(idx >> CLOG) * S + cg0 + (idx & (C - 1))] for idx in range(1 << TILE_LOG)]
            tile = tile_dif(tile, r1, w512)
            for idx in range(1 << TILE_LOG):
                p, col = idx >> CLOG, idx & (C - 1)
                k = bitrev(p, r1)
                cg = cg0 + col
                ex = (k * cg) & (N - 1)
                val = tile[idx] * (tw_a[direct_index(ex, 0)] if direct else pow(w, ex, Q)) % Q
                lo_, hi_ = cg & (R3 - 1), cg >> r3
                dst[lo_ * (N >> r3) + hi_ * R1 + k] = val
        buf = dst
        # ---- pass B (3-pass only): slab hi = i3, rows i2 (stride R1), cols k1
        if P == 3:
            r2 = radices[1]
            CLOG = TILE_LOG - r2
            C = 1 << CLOG
            ncols = N >> r2
            nxt = list(buf)
            for blk in range(ncols // C):
                cg0 = blk * C
                def addr(row, cg):
                    return (cg >> r1) * (R1 * R2) + row * R1 + (cg & (R1 - 1))
                tile = [buf[addr(idx >> CLOG, cg0 + (idx & (C - 1)))] for idx in range(1 << TILE_LOG)]
                tile = tile_dif(tile, r2, w512)
                for idx in range(1 << TILE_LOG):
                    p, col = idx >> CLOG, idx & (C - 1)
                    k = bitrev(p, r2)
                    cg = cg0 + col
                    twcol = (cg >> r1) << r1
                    ex = (k * twcol) & (N - 1)
                    val = tile[idx] * (tw_b[direct_index(ex, r1)] if direct else pow(w, ex, Q)) % Q
                    nxt[addr(k, cg)] = val
            buf = nxt
        # ---- pass C: rows (stride N / R3), cols contiguous, no twiddle
        CLOG = TILE_LOG - r3
        C = 1 << CLOG
        S3 = N >> r3
        out = [None] * N
        for blk in range(S3 // C):
            cg0 = blk * C
            tile = [buf[(idx >> CLOG) * S3 + cg0 + (idx & (C - 1))] for idx in range(1 << TILE_LOG)]
            tile = tile_dif(tile, r3, w512)
            for idx in range(1 << TILE_LOG):
                p, col = idx >> CLOG, idx & (C - 1)
                out[bitrev(p, r3) * S3 + cg0 + col] = tile[idx]
    if inverse:
        out = [x * n_inv % Q for x in out]
        if coset:
            gi = fr_inv(g)
            out = [out[i] * pow(gi, i, Q) % Q for i in range(N)]
    return out
