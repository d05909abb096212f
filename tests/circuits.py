"""Shared helpers of the parity tests: circuits laid out with the oracle Composer, turned into the
byte arrays both provers take (the C restatement oracle/c/oracle_prove.c and the HIP prover through
the C-ABI), at sizes where the big-int oracle's own compile step would be too slow.

Column construction follows Compiler::preprocess (reference src/compiler.rs:116-232) and
Permutation::compute_sigma_polynomials (src/composer/permutation.rs:150-211); interpolation uses
the C restatement of EvaluationDomain::ifft (oracle/c/oracle.c)."""
from __future__ import annotations

import functools
import random

from oracle import bls12_381 as E
from oracle import cbind
from oracle import plonk as O
from tests import widget_circuits as WC

Q = E.Q
SIGMA = ["s_sigma_1", "s_sigma_2", "s_sigma_3", "s_sigma_4"]


def fr_bytes(vals) -> bytes:
    R = E.FR_R
    return b"".join((v % Q * R % Q).to_bytes(32, "little") for v in vals)


def fr_vals(buf: bytes):
    return [int.from_bytes(buf[i:i + 32], "little") * E.FR_RINV % Q for i in range(0, len(buf), 32)]


def next_pow2(n: int) -> int:
    p = 1
    while p < n:
        p *= 2
    return p


def wires_of(composer, size):
    W = composer.witnesses
    cols = [[0] * size for _ in range(4)]
    for i, g in enumerate(composer.constraints):
        cols[0][i], cols[1][i], cols[2][i], cols[3][i] = W[g.a], W[g.b], W[g.c], W[g.d]
    return cols


def compile_fast(composer, label: bytes) -> dict:
    """Everything a prover needs from a composed circuit, as Montgomery byte strings.
    polys: {name: bytes} coefficient form over the size-n domain (untrimmed)."""
    constraints = len(composer.constraints)
    size = next_pow2(constraints)
    log_n = size.bit_length() - 1
    polys = {}
    for name in O.SELECTORS:
        col = [getattr(g, name) % Q for g in composer.constraints] + [0] * (size - constraints)
        polys[name] = cbind.ntt_bytes(fr_bytes(col), log_n, True, False, size) if any(col) else b""
    omega = pow(E.ROOT_OF_UNITY, 1 << (32 - log_n), Q)
    roots, cur = [], 1
    for _ in range(size):
        roots.append(cur)
        cur = cur * omega % Q
    ks = [1, E.K1, E.K2, E.K3]
    for i, mapping in enumerate(composer.sigma_mappings(size)):
        lag = [ks[col] * roots[idx] % Q for col, idx in mapping]
        polys[SIGMA[i]] = cbind.ntt_bytes(fr_bytes(lag), log_n, True, False, size)
    pi = dict(composer.public_inputs)
    idx = sorted(pi)
    return dict(constraints=constraints, size=size, log_n=log_n, label=label, polys=polys,
                wires=[fr_bytes(c) for c in wires_of(composer, size)],
                pi=pi, pi_idx=idx, pi_val=fr_bytes([pi[i] for i in idx]))


def circuit_columns(composer) -> dict:
    """A composed circuit as the columns Compiler::preprocess starts from (compiler.rs:145-175, composer.rs:119-167):
    per-gate selector values (Montgomery bytes, all-zero selectors omitted), the witness index on every wire,
    the witness values."""
    sel = {}
    for name in O.SELECTORS:
        col = [getattr(g, name) % Q for g in composer.constraints]
        if any(col):
            sel[name] = fr_bytes(col)
    wires = [[getattr(g, w) for g in composer.constraints] for w in "abcd"]
    return dict(selectors=sel, wires=wires, witnesses=len(composer.witnesses), values=fr_bytes(composer.witnesses))


def big_widget_circuit(ngates: int, seed: int = 1):
    """>= ngates - 8 and <= ngates gates: every widget family with honest non-trivial witnesses
    (tests/widget_circuits.py gadgets), random arithmetic gates in between, public inputs."""
    def build():
        r = random.Random(seed)
        c = O.Composer()
        base = WC.jj_base()
        pts = []
        ws = [c.append_witness(r.randrange(Q)) for _ in range(4)]
        npi = 0
        while len(c.constraints) < ngates - 64:
            kind = r.randrange(6)
            if kind == 0:
                WC.add_range(c, r.getrandbits(32), 16)
            elif kind == 1:
                WC.add_logic(c, r.getrandbits(20), r.getrandbits(20), 10, xor=bool(r.getrandbits(1)))
            elif kind == 2:
                pts.append(WC.add_fixed_base(c, base, [r.choice((-1, 0, 1)) for _ in range(12)]))
            elif kind == 3 and len(pts) >= 2:
                pts.append(WC.add_curve_addition(c, r.choice(pts), r.choice(pts)))
                pts = pts[-8:]
            elif kind == 4:
                for _ in range(8):
                    ws.append(c.gate_mul(r.choice(ws), r.choice(ws), r.choice(ws), q_m=r.randrange(1, Q), q_f=1, q_c=r.randrange(Q)))
                ws = ws[-16:]
            elif npi < 5:
                v = r.randrange(Q)
                c.append_gate(O.Gate(a=c.append_witness(v), q_l=Q - 1, pi=v))   # append_public (composer.rs:377-389)
                npi += 1
        while len(c.constraints) < ngates - 1:
            ws.append(c.gate_add(r.choice(ws), r.choice(ws), r.choice(ws), q_l=r.randrange(Q), q_r=r.randrange(Q), q_f=1, q_c=r.randrange(Q)))
        v = r.randrange(Q)
        c.append_gate(O.Gate(a=c.append_witness(v), q_l=Q - 1, pi=v))           # a public input on the last row
        return c
    return build


@functools.lru_cache(maxsize=2)   # the 2^20-gate tests of one module ask for the same 100 MB key several times in a row
def synthetic_srs(n: int, tau: int = 0x5EED0000 * 0x9E3779B97F4A7C15 % Q, g: int = 0xA5A5A5A5DEADBEEF) -> bytes:
    """[g tau^i] G1, i < n, as raw 96-byte points (C oracle; PublicParameters::setup semantics)."""
    return cbind.srs_generate(fr_bytes([tau]), fr_bytes([g]), n)


def blinders(seed: int) -> bytes:
    r = random.Random(seed)
    return fr_bytes([r.randrange(Q) for _ in range(14)])
