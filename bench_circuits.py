"""Synthetic circuits for bench.py / the tests: deterministic, seed-stamped, satisfied by
construction.  Pure Python big-int arithmetic; nothing here touches the GPU library or the oracle.

Profiles (all of size n = 2^log_n, every row an active gate, non-trivial copy permutation):
  dense        arithmetic gates q_M a b + q_L a + q_R b + q_O c + q_F d + q_C = 0 with uniformly random
               selectors and wire values (seed 0x5eed0001) — the headline workload
  bench-like   the same gates with the §8(d) value mix: half of the wire values < 4 (bits / quads as range,
               logic and decomposition gadgets produce them), half uniform (seed 0x5eed0002)
  widgets      BenchCircuit-shaped (reference benches/plonk.rs:33-82): every selector family active — range
               (composer/range.rs:68-130), logic XOR / AND (composer/logic.rs:42-170), fixed-base scalar
               multiplication rounds (composer/fixed_base.rs:160-290), curve addition (composer/point.rs:356-408)
               with honest witnesses (bit quads, truth tables, JubJub points), arithmetic gates in between
               and two public inputs; a block of 256 rows tiled over the domain, tiles linked by copy
               constraints (seed 0x5eed0003)

Everything is returned in Montgomery representation (x~ = x R mod q, R = 2^256) as raw little-endian
limb bytes — the in-memory form of BlsScalar.0 that the C-ABI takes."""
from __future__ import annotations

import random

Q = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
R = (1 << 256) % Q
RINV = pow(R, -1, Q)
K1, K2, K3 = 7, 13, 17
ROOT_OF_UNITY = pow(7, (Q - 1) >> 32, Q)
EDWARDS_D = (-10240 * pow(10241, -1, Q)) % Q      # dusk_jubjub::EDWARDS_D:  -x^2 + y^2 = 1 + d x^2 y^2
SELECTORS = ["q_m", "q_l", "q_r", "q_o", "q_f", "q_c", "q_arith", "q_range", "q_logic",
             "q_fixed_group_add", "q_variable_group_add"]


def _bytes(col):
    tb = int.to_bytes
    return b"".join(tb(x, 32, "little") for x in col)


def _omega_table(log_n):
    """w^i * R for i < n (Montgomery form)."""
    n = 1 << log_n
    omega = pow(ROOT_OF_UNITY, 1 << (32 - log_n), Q)
    t, T = R, [0] * n
    for i in range(n):
        T[i] = t
        t = t * omega % Q
    return T


def _arith_chunk(args):
    """rows [0, cnt) of an independent gate chain (own RNG stream); returns the nine columns as lists"""
    profile, seed, cnt = args
    rnd = random.Random(seed)
    rb = rnd.getrandbits
    if profile == "dense":
        a = [0] * cnt
        b = [rb(254) % Q for _ in range(cnt)]
        d = [rb(254) % Q for _ in range(cnt)]
        qm = [rb(254) % Q for _ in range(cnt)]
        ql = [rb(254) % Q for _ in range(cnt)]
        qr = [rb(254) % Q for _ in range(cnt)]
        qf = [rb(254) % Q for _ in range(cnt)]
        qc = [rb(254) % Q for _ in range(cnt)]
        c = [0] * cnt
        cur = rb(254) % Q
        for i in range(cnt):
            a[i] = cur
            bi = b[i]
            # all values are Montgomery forms: c~ = (qm~ a~ b~ R^-2 + ql~ a~ R^-1 + qr~ b~ R^-1 + qf~ d~ R^-1 + qc~)
            cur = ((qm[i] * cur % Q * bi % Q * RINV + ql[i] * cur + qr[i] * bi + qf[i] * d[i]) % Q * RINV + qc[i]) % Q
            c[i] = cur
    elif profile == "bench-like":
        def val():      # half small (< 4), half uniform — Montgomery form of the value
            return (rb(2) * R) % Q if rb(1) else rb(254) % Q
        b = [val() for _ in range(cnt)]
        d = [val() for _ in range(cnt)]
        c = [val() for _ in range(cnt)]
        a = [val()] + c[:-1]
        qm = [rb(254) % Q for _ in range(cnt)]
        ql = [rb(254) % Q for _ in range(cnt)]
        qr = [rb(254) % Q for _ in range(cnt)]
        qf = [rb(254) % Q for _ in range(cnt)]
        qc = [0] * cnt
        for i in range(cnt):
            lhs = (qm[i] * a[i] % Q * b[i] % Q * RINV + ql[i] * a[i] + qr[i] * b[i] + qf[i] * d[i]) % Q * RINV % Q
            qc[i] = (c[i] - lhs) % Q
    else:
        raise ValueError(profile)
    return [_bytes(col) for col in (a, b, c, d, qm, ql, qr, qf, qc)]


def _cached(kind: str, log_n: int, profile: str, make):
    """PLONK_CIRCUIT_CACHE=<directory> (the test suite sets it): the generated circuit of (kind, size, profile) is written
    once and re-read by every later caller — the same deterministic bytes either way.  A 2^20-gate circuit is ~12 s of
    pure-Python big-integer work and a GPU test session used to regenerate it a dozen times (every rank of every
    multi-rank child included); generation is serialised by a file lock so that eight ranks starting together produce it
    once.  Nothing is cached below 2^15 gates or without the variable (bench.py as the driver runs it)."""
    import os
    root = os.environ.get("PLONK_CIRCUIT_CACHE")
    if not root or log_n < 15:
        return make()
    import fcntl
    import pickle
    path = os.path.join(root, f"{kind}_{profile}_2p{log_n}_v1.pkl")
    try:
        os.makedirs(root, exist_ok=True)
        lock = open(path + ".lock", "w")
    except OSError:                      # cache directory unusable: generate as if there were none
        return make()
    with lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            try:
                if os.path.exists(path):
                    with open(path, "rb") as f:
                        return pickle.load(f)
            except (OSError, EOFError, pickle.UnpicklingError):
                pass                     # unreadable entry: regenerate (and overwrite it below)
            out = make()
            tmp = path + f".tmp{os.getpid()}"
            try:
                with open(tmp, "wb") as f:
                    pickle.dump(out, f, protocol=pickle.HIGHEST_PROTOCOL)
                os.replace(tmp, path)
            except OSError:              # disk full / read-only: the caller still gets its circuit
                try:
                    os.unlink(tmp)
                except OSError:
                    pass
            return out
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def arithmetic_circuit(log_n: int, profile: str = "dense", workers: int = 0):
    return _cached("arith", log_n, profile, lambda: _arithmetic_circuit(log_n, profile, workers))


def _arithmetic_circuit(log_n: int, profile: str = "dense", workers: int = 0):
    """Chains of arithmetic gates: the output of gate i is wired to input a of gate i + 1 (sigma_1, sigma_3
    non-trivial), q_O = -1, q_arith = 1.  `dense`: random selectors, the output follows.  `bench-like`: the
    wire values are drawn first (half of them < 4), q_C is solved for so that each gate holds.
    Up to 2^20 gates it is ONE chain from one RNG stream (the headline workload, stable proof digest); above,
    the rows are 2^20-gate chains with their own streams, generated in `workers` processes when asked.
    Returns (wires[4] bytes, key columns {name: bytes} in evaluation form, trivial polys {name: [ints]})."""
    n = 1 << log_n
    seed0 = {"dense": 0x5EED0001, "bench-like": 0x5EED0002}[profile]
    seg = min(n, 1 << 20)
    jobs = [(profile, seed0 + 0x10000 * k, seg, log_n, k * seg) for k in range(n // seg)]
    if workers > 1 and len(jobs) > 1:
        # plain subprocesses of this file (no multiprocessing: nothing depends on the caller's __main__, no fork of a
        # process that holds a GPU context); each writes its eleven columns (nine gate columns + its rows of sigma_1 / sigma_3:
        # round 5 — the permutation columns used to be 15 s of serial work in the parent at 2^22 gates) to a file
        import os
        import subprocess
        import sys
        import tempfile
        parts = [None] * len(jobs)
        with tempfile.TemporaryDirectory() as tmp:
            pending = list(enumerate(jobs))
            running = []
            while pending or running:
                while pending and len(running) < workers:
                    k, (prof, seed, cnt, lg, first) = pending.pop(0)
                    path = os.path.join(tmp, f"chunk{k}.bin")
                    running.append((k, path, subprocess.Popen([sys.executable, os.path.abspath(__file__), prof, str(seed), str(cnt), path,
                                                               str(lg), str(first)])))
                k, path, proc = running.pop(0)
                if proc.wait(timeout=900) != 0:
                    raise RuntimeError("circuit chunk generator failed")
                raw = open(path, "rb").read()
                step = len(raw) // 11
                parts[k] = [raw[i * step:(i + 1) * step] for i in range(11)]
    else:
        parts = [_arith_chunk(j[:3]) + _sigma_chunk(j[3], j[4], j[2]) for j in jobs]
    col = [b"".join(p[k] for p in parts) for k in range(11)]
    wires = col[:4]
    cols = {"s_sigma_1": col[9], "s_sigma_3": col[10]}
    for name, k in (("q_m", 4), ("q_l", 5), ("q_r", 6), ("q_f", 7), ("q_c", 8)):
        cols[name] = col[k]
    trivial = {"q_o": [Q - 1], "q_arith": [1], "s_sigma_2": [0, K1], "s_sigma_4": [0, K3]}
    return wires, cols, trivial


def _sigma_chunk(log_n: int, first: int, cnt: int):
    """sigma_1 / sigma_3 (evaluation form, Montgomery bytes) of the gate chain that occupies rows [first, first + cnt):
    inside a chain sigma_1[i] = K2 w^(i-1) (Output(i-1)), sigma_3[i] = w^(i+1) (Left(i+1)); the chain's ends map to themselves."""
    omega = pow(ROOT_OF_UNITY, 1 << (32 - log_n), Q)
    t, T = R * pow(omega, first, Q) % Q, [0] * cnt          # T[j] = w^(first + j) * R
    for j in range(cnt):
        T[j] = t
        t = t * omega % Q
    s1 = [T[j] if j == 0 else K2 * T[j - 1] % Q for j in range(cnt)]
    s3 = [K2 * T[j] % Q if j == cnt - 1 else T[j + 1] for j in range(cnt)]
    return [_bytes(s1), _bytes(s3)]


# ---- JubJub (the widgets only need the curve equation, not the subgroup) -----------------------------
def _fr_sqrt(a):
    a %= Q
    if a == 0:
        return 0
    if pow(a, (Q - 1) // 2, Q) != 1:
        return None
    s, t = 32, (Q - 1) >> 32
    z = pow(7, t, Q)
    m, c, r, b = s, z, pow(a, (t + 1) // 2, Q), pow(a, t, Q)
    while b != 1:
        i, x = 0, b
        while x != 1:
            x = x * x % Q
            i += 1
        f = pow(c, 1 << (m - i - 1), Q)
        m, c, r, b = i, f * f % Q, r * f % Q, b * f * f % Q
    return r


def jj_add(p1, p2):
    (x1, y1), (x2, y2) = p1, p2
    k = EDWARDS_D * x1 % Q * x2 % Q * y1 % Q * y2 % Q
    return ((x1 * y2 + y1 * x2) * pow(1 + k, -1, Q) % Q, (y1 * y2 + x1 * x2) * pow(1 - k, -1, Q) % Q)


def jj_point():
    y = 2
    while True:
        x = _fr_sqrt((y * y - 1) * pow(EDWARDS_D * y * y + 1, -1, Q))
        if x:
            return (x, y)
        y += 1


class _Rows:
    """Row-by-row layout: wire VALUES (a, b, c, d) and selector values per gate."""

    def __init__(self):
        self.rows = []

    def gate(self, a=0, b=0, c=0, d=0, **sel):
        self.rows.append(((a % Q, b % Q, c % Q, d % Q), {k: v % Q for k, v in sel.items()}))

    def arith(self, a=0, b=0, c=0, d=0, **sel):
        sel["q_arith"] = 1
        self.gate(a, b, c, d, **sel)

    # range gadget: accumulators of 2-bit quads, (d, c, b, a) per row, the next row's d continues
    def range(self, value, quads):
        digits = [(value >> (2 * (quads - 1 - i))) & 3 for i in range(quads)]
        accs, acc = [0], 0
        for q in digits:
            acc = 4 * acc + q
            accs.append(acc)
        assert len(accs) % 4 == 1
        for r in range((len(accs) - 1) // 4):
            self.gate(a=accs[4 * r + 3], b=accs[4 * r + 2], c=accs[4 * r + 1], d=accs[4 * r], q_range=1)
        self.arith(d=accs[-1])                                         # carries the final accumulator as d_next

    # logic gadget: rows (a_i, b_i, w_{i+1}, d_i), q_c = q_logic = -1 (XOR) / +1 (AND); closing row
    def logic(self, x, y, quads, xor):
        sel = Q - 1 if xor else 1
        a = b = d = 0
        for i in range(quads):
            qa = (x >> (2 * (quads - 1 - i))) & 3
            qb = (y >> (2 * (quads - 1 - i))) & 3
            self.gate(a=a, b=b, c=qa * qb, d=d, q_c=sel, q_logic=sel)
            a, b = 4 * a + qa, 4 * b + qb
            d = 4 * d + ((qa ^ qb) if xor else (qa & qb))
        self.gate(a=a, b=b, d=d)

    # fixed-base rounds: (acc_x, acc_y, xy_alpha, scalar_acc), q_l = x_beta, q_r = y_beta, q_c = xy_beta
    def fixed_base(self, base, digits):
        mult = [base]
        for _ in range(1, len(digits)):
            mult.append(jj_add(mult[-1], mult[-1]))
        mult.reverse()
        acc_pt, acc_sc = (0, 1), 0
        for i, dg in enumerate(digits):
            pt = (0, 1) if dg == 0 else (mult[i] if dg == 1 else ((-mult[i][0]) % Q, mult[i][1]))
            xb, yb = mult[i]
            self.gate(a=acc_pt[0], b=acc_pt[1], c=pt[0] * pt[1], d=acc_sc, q_l=xb, q_r=yb, q_c=xb * yb,
                      q_fixed_group_add=1)
            acc_pt, acc_sc = jj_add(acc_pt, pt), (2 * acc_sc + dg) % Q
        self.arith(a=acc_pt[0], b=acc_pt[1], d=acc_sc)
        return acc_pt

    # curve addition: (x1, y1, x2, y2) with the selector, then (x3, y3, 0, x1 y2)
    def curve_add(self, p1, p2):
        p3 = jj_add(p1, p2)
        self.gate(a=p1[0], b=p1[1], c=p2[0], d=p2[1], q_variable_group_add=1)
        self.gate(a=p3[0], b=p3[1], d=p1[0] * p2[1])
        return p3

    def mul_gate(self, r, a, b, d):
        qm, qf, qc = r.randrange(1, Q), r.randrange(Q), r.randrange(Q)
        self.arith(a=a, b=b, c=qm * a * b + qf * d + qc, d=d, q_m=qm, q_f=qf, q_c=qc, q_o=Q - 1)


def widget_block(blk: int, seed: int = 0x5EED0003) -> _Rows:
    """`blk` rows with every widget family, ending on a plain arithmetic row (so a tile never looks into
    the next one through the X -> omega X rotation with anything but closing-row zeros)."""
    r = random.Random(seed)
    rows = _Rows()
    base = jj_point()
    pts = []
    order, step = (2, 0, 1, 2, 3, 4, 3), 0      # fixed base twice before the first curve addition
    while len(rows.rows) < blk - 48:
        kind = order[step % len(order)]
        step += 1
        if kind == 0:
            rows.range(r.getrandbits(32), 16)
        elif kind == 1:
            rows.logic(r.getrandbits(20), r.getrandbits(20), 10, xor=bool(r.getrandbits(1)))
        elif kind == 2:
            pts.append(rows.fixed_base(base, [r.choice((-1, 0, 1)) for _ in range(12)]))
        elif kind == 3:
            pts.append(rows.curve_add(r.choice(pts), r.choice(pts)))
            pts = pts[-6:]
        else:
            for _ in range(6):
                rows.mul_gate(r, r.randrange(Q), r.randrange(Q), r.randrange(Q))
    while len(rows.rows) < blk:
        rows.mul_gate(r, r.randrange(Q), r.randrange(Q), r.randrange(Q))
    assert len(rows.rows) == blk
    return rows


def _widget_layout(log_n, blk_log, pool):
    """(rows per tile, tiles, blocks in the pool, block id of every tile, the blocks' rows)"""
    n = 1 << log_n
    blk = 1 << min(blk_log, log_n)
    tiles = n // blk
    pool = max(1, min(pool, tiles))
    rnd = random.Random(0x5EED0003)
    blocks = [widget_block(blk, 0x5EED0003 + 7919 * b).rows for b in range(pool)]
    ids = [t % pool for t in range(tiles)]
    rnd.shuffle(ids)
    return blk, tiles, pool, ids, blocks


def widget_circuit(log_n: int, blk_log: int = 8, pool: int = 64):
    if blk_log == 8 and pool == 64:
        return _cached("widget", log_n, "widgets", lambda: _widget_circuit(log_n, blk_log, pool))
    return _widget_circuit(log_n, blk_log, pool)


def _widget_circuit(log_n: int, blk_log: int = 8, pool: int = 64):
    """n = 2^log_n rows: tiles of 2^blk_log rows drawn at random from a pool of `pool` different blocks (different
    witnesses), so no column is periodic and every polynomial is dense.  Position (column, row) of a tile is
    copy-constrained to the same position of the next tile built from the same block (equal values by
    construction): the permutation is a product of long cycles; two public inputs sit on the last two rows
    of tile 0.
    Returns (wires[4] bytes, key columns {name: bytes} in evaluation form, public inputs {row: value})."""
    n = 1 << log_n
    blk, tiles, pool, ids, blocks = _widget_layout(log_n, blk_log, pool)
    mont = lambda v: v % Q * R % Q   # noqa: E731
    wcols = [[_bytes([mont(rw[0][col]) for rw in rows]) for rows in blocks] for col in range(4)]
    wires = [b"".join(wcols[col][b] for b in ids) for col in range(4)]
    cols = {}
    for name in SELECTORS:
        per_block = [[mont(rw[1].get(name, 0)) for rw in rows] for rows in blocks]
        if any(any(v) for v in per_block):
            raw = [_bytes(v) for v in per_block]
            cols[name] = bytearray(b"".join(raw[b] for b in ids))
    # public inputs: rows blk-2, blk-1 of tile 0 are `-a + PI = 0` (q_l = -1, q_arith = 1, PI = a); the tiles that
    # repeat this block keep their multiplication gate on those rows — the wire values are identical either way
    pi = {}
    for row in (blk - 2, blk - 1):
        a_val = blocks[ids[0]][row][0][0]
        for name in cols:
            v = {"q_l": Q - 1, "q_arith": 1}.get(name, 0)
            cols[name][32 * row:32 * row + 32] = int.to_bytes(mont(v), 32, "little")
        pi[row] = a_val
    # sigma: tile t -> the next tile with the same block id (cyclically)
    nxt = [0] * tiles
    last, first = {}, {}
    for t, b in enumerate(ids):
        if b in last:
            nxt[last[b]] = t
        else:
            first[b] = t
        last[b] = t
    for b, t in last.items():
        nxt[t] = first[b]
    T = _omega_table(log_n)
    ks = (1, K1, K2, K3)
    for k in range(4):   # sigma_k[t * blk + r] = K_k * w^(nxt[t] * blk + r)
        out = []
        for t in range(tiles):
            seg = T[nxt[t] * blk:(nxt[t] + 1) * blk]
            out.append(_bytes(seg if k == 0 else [ks[k] * v % Q for v in seg]))
        cols[f"s_sigma_{k + 1}"] = b"".join(out)
    return wires, {k: bytes(v) for k, v in cols.items()}, pi


# ---- the same circuits as gate columns: what Compiler::preprocess starts from (reference src/compiler.rs:145-175,
# src/composer.rs:119-167) — per-gate selector values, the witness index on every wire, the witness values ---------
def arithmetic_columns(log_n: int, profile: str = "dense", workers: int = 0) -> dict:
    """arithmetic_circuit(log_n, profile) with its copy constraints expressed through witnesses: gate i owns the
    witnesses 3i (output c), 3i + 1 (b), 3i + 2 (d); its input a is the output witness of gate i - 1, or an own
    witness 3n + k at the head of chain k.  Returns {selectors: {name: bytes}, wires: [4 x uint32 bytes],
    witnesses: count, values: bytes (Montgomery)} plus the padded wire columns under `columns`."""
    import numpy as np
    n = 1 << log_n
    seg = min(n, 1 << 20)
    wires, cols, _trivial = arithmetic_circuit(log_n, profile, workers)
    i = np.arange(n, dtype=np.int64)
    head = i % seg == 0
    ids = [np.where(head, 3 * n + i // seg, 3 * (i - 1)), 3 * i + 1, 3 * i, 3 * i + 2]   # a, b, c, d
    count = 3 * n + n // seg
    vals = np.zeros((count, 32), np.uint8)
    for col in (0, 1, 3, 2):   # c last: inside a chain a[i] and c[i - 1] are the same witness with the same value
        vals[ids[col]] = np.frombuffer(wires[col], np.uint8).reshape(n, 32)
    one = int.to_bytes(R, 32, "little")
    minus_one = int.to_bytes((Q - 1) * R % Q, 32, "little")
    selectors = {k: v for k, v in cols.items() if k.startswith("q_")}
    selectors["q_o"] = minus_one * n
    selectors["q_arith"] = one * n
    return dict(selectors=selectors, wires=[x.astype(np.uint32).tobytes() for x in ids], witnesses=count,
                values=vals.tobytes(), columns=wires, public_inputs={})


def widget_columns(log_n: int, blk_log: int = 8, pool: int = 64) -> dict:
    """widget_circuit(log_n) the same way: position (column, row) of pool block b is ONE witness shared by every tile
    drawn from b — the copy constraints widget_circuit writes into the sigma columns directly."""
    import numpy as np
    n = 1 << log_n
    blk, tiles, pool, ids, blocks = _widget_layout(log_n, blk_log, pool)
    wires, cols, pi = widget_circuit(log_n, blk_log, pool)
    tile_ids = np.repeat(np.asarray(ids, dtype=np.int64), blk)
    row = np.tile(np.arange(blk, dtype=np.int64), tiles)
    idx = [((tile_ids * 4 + col) * blk + row).astype(np.uint32).tobytes() for col in range(4)]
    mont = lambda v: v % Q * R % Q   # noqa: E731
    vals = b"".join(_bytes([mont(rw[0][col]) for rw in blocks[b]]) for b in range(pool) for col in range(4))
    return dict(selectors={k: v for k, v in cols.items() if k.startswith("q_")}, wires=idx, witnesses=pool * 4 * blk,
                values=vals, columns=wires, public_inputs=pi)


if __name__ == "__main__":   # chunk worker of arithmetic_circuit: <profile> <seed> <count> <output file> <log_n> <first row>
    import sys
    _prof, _seed, _cnt, _out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    with open(_out, "wb") as _f:
        for _col in _arith_chunk((_prof, _seed, _cnt)) + _sigma_chunk(int(sys.argv[5]), int(sys.argv[6]), _cnt):
            _f.write(_col)
