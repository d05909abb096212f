"""Prover::prove restatement around the NTT/MSM hot path (oracle; test
infrastructure only — big ints, small sizes).

Follows the reference, file:line cited per function:
  Composer (minimal: witnesses, raw gates, arithmetic helpers)   src/composer.rs:105-240,264-440
  Permutation                                                    src/composer/permutation.rs:70-294
  PublicParameters::setup / trim, CommitKey::commit              src/commitment_scheme/kzg10/srs.rs:61-100,188-196; key.rs:336-417
  Compiler::preprocess                                           src/compiler.rs:116-461
  Prover::new / prove_inner                                      src/compiler/prover.rs:53-115,415-761
  quotient_poly::compute                                         src/proof_system/quotient_poly.rs:20-310
  widget compute_quotient_i / compute_linearization              src/proof_system/widget/**/proverkey.rs
  linearization_poly::compute                                    src/proof_system/linearization_poly.rs:168-264
  Proof::to_bytes                                                src/proof_system/proof.rs:137-162
"""
from __future__ import annotations

from dataclasses import dataclass, field

from . import bls12_381 as E
from .bls12_381 import EDWARDS_D, K1, K2, K3, Q, fr_inv, fr_to_bytes, g1_compress
from .fft import EvaluationDomain, next_pow2
from .merlin import Transcript

ADDED_BLINDING_DEGREE = 6          # srs.rs:54
CIRCUIT_SIZE_PADDING = 6           # compiler.rs:48
SELECTORS = ["q_m", "q_l", "q_r", "q_o", "q_f", "q_c", "q_arith", "q_range",
             "q_logic", "q_fixed_group_add", "q_variable_group_add"]


# ---------------------------------------------------------------------------
# polynomial helpers (src/fft/polynomial.rs)
# ---------------------------------------------------------------------------
def poly_trim(c):                                   # from_coefficients_vec :79-93
    c = [x % Q for x in c]
    while c and c[-1] == 0:
        c.pop()
    return c


def poly_eval(c, x):                                # evaluate :120-137
    acc, p = 0, 1
    for v in c:
        acc = (acc + v * p) % Q
        p = p * x % Q
    return acc


def poly_scale(c, s):                               # Mul<&BlsScalar> :409-421
    return poly_trim([v * s for v in c])


def poly_add(a, b):                                 # Add :178-203
    n = max(len(a), len(b))
    a = a + [0] * (n - len(a))
    b = b + [0] * (n - len(b))
    return poly_trim([x + y for x, y in zip(a, b)])


def poly_ruffini(c, z):                             # ruffini :345-367
    quotient, k = [], 0
    for coeff in reversed(c):
        t = (coeff + k) % Q
        quotient.append(t)
        k = z * t % Q
    quotient.pop()
    quotient.reverse()
    return poly_trim(quotient)


def batch_inversion(v):                             # util.rs:87-117 (zeros skipped)
    return [fr_inv(x) if x % Q else 0 for x in v]


# ---------------------------------------------------------------------------
# Composer (minimal)
# ---------------------------------------------------------------------------
@dataclass
class Gate:
    a: int = 0
    b: int = 0
    c: int = 0
    d: int = 0
    q_m: int = 0
    q_l: int = 0
    q_r: int = 0
    q_o: int = 0
    q_f: int = 0
    q_c: int = 0
    q_arith: int = 0
    q_range: int = 0
    q_logic: int = 0
    q_fixed_group_add: int = 0
    q_variable_group_add: int = 0
    pi: int | None = None


class Composer:
    """Composer::initialized + append_witness / append_gate / assert_equal_constant
    (composer.rs:105-113,118-167,177-240,402-418). Witness index 0 is ZERO."""

    def __init__(self):
        self.witnesses: list[int] = []
        self.constraints: list[Gate] = []
        self.public_inputs: dict[int, int] = {}
        self.witness_map: dict[int, list[tuple[int, int]]] = {}   # permutation.rs:70-102
        zero = self.append_witness(0)
        one = self.append_witness(1)
        self.assert_equal_constant(zero, 0)
        self.assert_equal_constant(one, 1)
        self._append_dummy_gates()

    def append_witness(self, v: int) -> int:
        self.witnesses.append(v % Q)
        self.witness_map[len(self.witnesses) - 1] = []
        return len(self.witnesses) - 1

    def append_custom_gate(self, g: Gate):
        n = len(self.constraints)
        self.constraints.append(g)
        if g.pi is not None:
            self.public_inputs[n] = g.pi % Q
        for col, w in enumerate((g.a, g.b, g.c, g.d)):
            self.witness_map[w].append((col, n))

    def append_gate(self, g: Gate):                 # Constraint::arithmetic :203-205
        g.q_arith = 1
        self.append_custom_gate(g)

    def assert_equal_constant(self, w: int, constant: int, public=None):
        self.append_gate(Gate(a=w, q_l=Q - 1, q_c=constant % Q, pi=public))

    def _append_dummy_gates(self):                  # composer.rs:204-240
        six = self.append_witness(6)
        one = self.append_witness(1)
        seven = self.append_witness(7)
        min_twenty = self.append_witness(Q - 20)
        self.append_gate(Gate(q_m=1, q_l=2, q_r=3, q_f=1, q_c=4, q_o=4,
                              a=six, b=seven, d=one, c=min_twenty))
        self.append_gate(Gate(q_m=1, q_l=1, q_r=1, q_c=127, q_o=1,
                              a=min_twenty, b=six, c=seven))

    def gate_add(self, a, b, d=0, q_l=1, q_r=1, q_f=0, q_c=0):   # :430-440
        av, bv, dv = self.witnesses[a], self.witnesses[b], self.witnesses[d]
        out = self.append_witness(q_l * av + q_r * bv + q_f * dv + q_c)
        self.append_gate(Gate(a=a, b=b, c=out, d=d, q_l=q_l % Q, q_r=q_r % Q,
                              q_f=q_f % Q, q_c=q_c % Q, q_o=Q - 1))
        return out

    def gate_mul(self, a, b, d=0, q_m=1, q_f=0, q_c=0):
        av, bv, dv = self.witnesses[a], self.witnesses[b], self.witnesses[d]
        out = self.append_witness(q_m * av * bv + q_f * dv + q_c)
        self.append_gate(Gate(a=a, b=b, c=out, d=d, q_m=q_m % Q, q_f=q_f % Q,
                              q_c=q_c % Q, q_o=Q - 1))
        return out

    # --- permutation.rs:106-141 / 150-170 ---
    def sigma_mappings(self, n):
        sig = [[(col, i) for i in range(n)] for col in range(4)]
        for wires in self.witness_map.values():
            for k, (col, idx) in enumerate(wires):
                sig[col][idx] = wires[(k + 1) % len(wires)]
        return sig

    def public_input_indexes(self):
        return sorted(self.public_inputs)


# ---------------------------------------------------------------------------
# KZG10 (srs.rs / key.rs)
# ---------------------------------------------------------------------------
def srs_setup(max_degree: int, rng, keep: int | None = None):
    """PublicParameters::setup (srs.rs:61-100).  RNG order: tau (:74), g scalar
    (:80), h scalar (:91, G2 — drawn but unused here).  `keep` computes only the
    first `keep` powers (the KAT trims to 23 of 1031)."""
    max_degree += ADDED_BLINDING_DEGREE
    x = rng.random_nonzero_scalar()
    g = E.g1_mul(E.G1_GEN, rng.random_nonzero_scalar())
    _h = rng.random_nonzero_scalar()
    npts = max_degree + 1 if keep is None else min(keep, max_degree + 1)
    pts, p = [], 1
    for _ in range(npts):
        pts.append(E.g1_mul(g, p))
        p = p * x % Q
    return pts


def srs_trim(powers_of_g, truncated_degree: int):   # srs.rs:188-196 + key.rs:336-354
    t = truncated_degree + ADDED_BLINDING_DEGREE
    assert 0 < t <= len(powers_of_g) - 1
    if t == 1:
        t += 1
    return powers_of_g[: t + 1]


def commit(ck, poly, msm=E.msm_naive):              # key.rs:376-388
    deg = len(poly) - 1 if poly else 0
    if deg > len(ck) - 1:
        raise ValueError("PolynomialDegreeTooLarge")
    return msm(ck, poly)


def compute_aggregate_witness(polys, point, v):     # key.rs:394-417
    n = max(len(p) for p in polys)
    acc, power = [0] * n, 1
    for p in polys:
        for i, t in enumerate(p):
            acc[i] = (acc[i] + t * power) % Q
        power = power * v % Q
    return poly_ruffini(poly_trim(acc), point)


# ---------------------------------------------------------------------------
# Compiler::preprocess + Prover::new
# ---------------------------------------------------------------------------
@dataclass
class ProverKey:
    n: int
    polys: dict = field(default_factory=dict)        # 11 selectors + s_sigma_1..4
    evals8: dict = field(default_factory=dict)       # same keys + "linear"
    v_h_coset_8n: list = field(default_factory=list)


@dataclass
class Prover:
    label: bytes
    pk: ProverKey
    ck: list
    vk: dict
    size: int
    constraints: int
    sigma_evaluations: list
    vanishing_coset_inverses: list


def compile_circuit(pp, label: bytes, composer: Composer, msm=E.msm_naive) -> Prover:
    """Compiler::compile_with_composer + preprocess (compiler.rs:116-461) and
    Prover::new (prover.rs:53-115)."""
    constraints = len(composer.constraints)
    ck = srs_trim(pp, next_pow2(constraints + CIRCUIT_SIZE_PADDING))
    size = next_pow2(constraints)
    domain = EvaluationDomain(size - 1 if size > 1 else 1)
    assert domain.size == size
    pk = ProverKey(n=size)
    for name in SELECTORS:
        col = [getattr(g, name) % Q for g in composer.constraints]
        pk.polys[name] = poly_trim(domain.ifft(col + [0] * (size - constraints)))
    roots = domain.elements()
    ks = [1, K1, K2, K3]
    for i, mapping in enumerate(composer.sigma_mappings(size)):   # permutation.rs:177-211
        lag = [ks[col] * roots[idx] % Q for col, idx in mapping]
        pk.polys[f"s_sigma_{i + 1}"] = poly_trim(domain.ifft(lag))
    vk = {"n": constraints}
    for name, poly in pk.polys.items():
        vk[name] = commit(ck, poly, msm)            # zero poly -> identity (unwrap_or_default)
    d8 = EvaluationDomain(8 * size)
    for name, poly in pk.polys.items():
        pk.evals8[name] = d8.coset_fft(poly)
    pk.evals8["linear"] = d8.coset_fft([0, 1])
    pk.v_h_coset_8n = d8.vanishing_poly_over_coset(size)
    vinv = batch_inversion(pk.v_h_coset_8n[:8])
    dom_c = EvaluationDomain(constraints)
    sig_ev = [dom_c.fft(pk.polys[f"s_sigma_{i}"]) for i in range(1, 5)]
    return Prover(label, pk, ck, vk, size, constraints, sig_ev, vinv)


def seed_transcript_v3(label: bytes, vk: dict, constraints: int, version: int = 3) -> Transcript:
    """Transcript::base_v3 (transcript.rs:131-145) + VerifierKey::seed_transcript
    (widget.rs:218-258).  version = 2: Transcript::base (transcript.rs:110-129) + seed_transcript_legacy
    (widget.rs:260-265) — the legacy seeding binds the LABEL s_sigma_4 to the commitment of s_sigma_1
    (seed_transcript_inner, widget.rs:224-228); everything else is the same."""
    t = Transcript(label)
    t.circuit_domain_sep(constraints)
    for lab in ["q_m", "q_l", "q_r", "q_o", "q_c", "q_f", "q_arith", "q_range", "q_logic",
                "q_variable_group_add", "q_fixed_group_add",
                "s_sigma_1", "s_sigma_2", "s_sigma_3", "s_sigma_4"]:
        t.append_commitment(lab.encode(), vk["s_sigma_1" if (version == 2 and lab == "s_sigma_4") else lab])
    t.circuit_domain_sep(vk["n"])
    return t


# ---------------------------------------------------------------------------
# Widgets: compute_quotient_i
# ---------------------------------------------------------------------------
def delta(f):                                       # range/proverkey.rs:88-93
    return f * (f - 1) * (f - 2) * (f - 3) % Q


def delta_xor_and(a, b, w, c, q_c):                 # logic/proverkey.rs:108-144
    F = w * (w * (4 * w - 18 * (a + b) + 81) + 18 * (a * a + b * b) - 81 * (a + b) + 83)
    Ee = 3 * (a + b + c) - 2 * F
    B = q_c * (9 * c - 3 * (a + b))
    return (B + Ee) % Q


def quotient_arith(e, i, a, b, c, d):               # arithmetic/proverkey.rs:44-71
    return ((a * b * e["q_m"][i] + a * e["q_l"][i] + b * e["q_r"][i] + c * e["q_o"][i]
             + d * e["q_f"][i] + e["q_c"][i]) * e["q_arith"][i]) % Q


def range_identity(ch, a, b, c, d, d_w):
    kappa = ch * ch % Q
    k2 = kappa * kappa % Q
    k3 = k2 * kappa % Q
    return (delta(c - 4 * d) + delta(b - 4 * c) * kappa + delta(a - 4 * b) * k2
            + delta(d_w - 4 * a) * k3) % Q


def quotient_range(e, i, ch, a, b, c, d, d_w):      # range/proverkey.rs:32-58
    return range_identity(ch, a, b, c, d, d_w) * e["q_range"][i] * ch % Q


def logic_identity(ch, a_i, a_w, b_i, b_w, c_i, d_i, d_w, q_c):
    kappa = ch * ch % Q
    k2 = kappa * kappa % Q
    k3 = k2 * kappa % Q
    k4 = k3 * kappa % Q
    a = (a_w - 4 * a_i) % Q
    b = (b_w - 4 * b_i) % Q
    d = (d_w - 4 * d_i) % Q
    w = c_i
    return (delta(a) + delta(b) * kappa + delta(d) * k2 + (w - a * b) * k3
            + delta_xor_and(a, b, w, d, q_c) * k4) % Q


def quotient_logic(e, i, ch, a, a_w, b, b_w, c, d, d_w):       # logic/proverkey.rs:34-70
    return e["q_logic"][i] * logic_identity(ch, a, a_w, b, b_w, c, d, d_w, e["q_c"][i]) * ch % Q


def fixed_identity(ch, a, a_w, b, b_w, c, d, d_w, q_l, q_r, q_c):
    kappa = ch * ch % Q
    k2 = kappa * kappa % Q
    k3 = k2 * kappa % Q
    x_beta, y_beta = q_l, q_r
    bit = (d_w - d - d) % Q
    bit_consistency = bit * (bit - 1) * (bit + 1) % Q
    y_alpha = (bit * bit * (y_beta - 1) + 1) % Q
    x_alpha = bit * x_beta % Q
    xy_consistency = (bit * q_c - c) * kappa % Q
    x_acc = ((a_w + a_w * c * a * b * EDWARDS_D) - (a * y_alpha + b * x_alpha)) * k2 % Q
    y_acc = ((b_w - b_w * c * a * b * EDWARDS_D) - (b * y_alpha + a * x_alpha)) * k3 % Q
    return (bit_consistency + x_acc + y_acc + xy_consistency) % Q


def quotient_fixed(e, i, ch, a, a_w, b, b_w, c, d, d_w):       # fixed_base/proverkey.rs:39-101
    return fixed_identity(ch, a, a_w, b, b_w, c, d, d_w, e["q_l"][i], e["q_r"][i],
                          e["q_c"][i]) * e["q_fixed_group_add"][i] * ch % Q


def var_identity(ch, a, a_w, b, b_w, c, d, d_w):
    kappa = ch * ch % Q
    x_1, x_3, y_1, y_3, x_2, y_2, x1_y2 = a, a_w, b, b_w, c, d, d_w
    xy_consistency = (x_1 * y_2 - x1_y2) % Q
    y1_x2, y1_y2, x1_x2 = y_1 * x_2 % Q, y_1 * y_2 % Q, x_1 * x_2 % Q
    x3c = ((x1_y2 + y1_x2) - (x_3 + x_3 * EDWARDS_D * x1_y2 * y1_x2)) * kappa % Q
    y3c = ((y1_y2 + x1_x2) - (y_3 - y_3 * EDWARDS_D * x1_y2 * y1_x2)) * kappa * kappa % Q
    return (xy_consistency + x3c + y3c) % Q


def quotient_var(e, i, ch, a, a_w, b, b_w, c, d, d_w):         # curve_addition/proverkey.rs:33-77
    return var_identity(ch, a, a_w, b, b_w, c, d, d_w) * e["q_variable_group_add"][i] * ch % Q


def quotient_perm(e, i, a, b, c, d, z, z_w, alpha, l1_alpha_sq, beta, gamma):
    """permutation/proverkey.rs:40-125."""
    x = e["linear"][i]
    ident = ((a + beta * x + gamma) * (b + beta * K1 * x + gamma) % Q
             * (c + beta * K2 * x + gamma) % Q * (d + beta * K3 * x + gamma) % Q * z * alpha) % Q
    copy = ((a + beta * e["s_sigma_1"][i] + gamma) * (b + beta * e["s_sigma_2"][i] + gamma) % Q
            * (c + beta * e["s_sigma_3"][i] + gamma) % Q
            * (d + beta * e["s_sigma_4"][i] + gamma) % Q * z_w * alpha) % Q
    one = (z - 1) * l1_alpha_sq % Q
    return (ident - copy + one) % Q


def quotient_compute(prover: Prover, z_poly, wires, pi_poly, ch):
    """quotient_poly::compute (quotient_poly.rs:20-137)."""
    alpha, beta, gamma, range_ch, logic_ch, fixed_ch, var_ch = ch
    pk = prover.pk
    n8 = 8 * pk.n
    d8 = EvaluationDomain(n8)
    e = pk.evals8
    z8, a8, b8, c8, d8e = [d8.coset_fft(p) for p in (z_poly, *wires)]   # :139-157
    for v in (z8, a8, b8, d8e):                                        # :61-67
        v.extend(v[:8])
    pi8 = d8.coset_fft(pi_poly)                                        # :177
    # compute_permutation_checks prelude (:265-284)
    l1_alpha_sq = alpha * alpha % Q
    lag = batch_inversion([(x - 1) % Q for x in e["linear"]])
    n_inv = d8.size_inv * 8 % Q
    lag = [li * vh % Q * n_inv % Q for li, vh in zip(lag, pk.v_h_coset_8n)]
    quotient = []
    for i in range(n8):
        a, b, c, d = a8[i], b8[i], c8[i], d8e[i]
        a_w, b_w, d_w = a8[i + 8], b8[i + 8], d8e[i + 8]
        t1 = (quotient_arith(e, i, a, b, c, d)
              + quotient_range(e, i, range_ch, a, b, c, d, d_w)
              + quotient_logic(e, i, logic_ch, a, a_w, b, b_w, c, d, d_w)
              + quotient_fixed(e, i, fixed_ch, a, a_w, b, b_w, c, d, d_w)
              + quotient_var(e, i, var_ch, a, a_w, b, b_w, c, d, d_w) + pi8[i])
        t2 = quotient_perm(e, i, a, b, c, d, z8[i], z8[i + 8], alpha,
                           lag[i] * l1_alpha_sq % Q, beta, gamma)
        quotient.append((t1 + t2) * prover.vanishing_coset_inverses[i & 7] % Q)   # :96-101
    t = poly_trim(d8.coset_ifft(quotient))                                       # :103-104
    if len(t) > 7 * pk.n:                                                        # :132
        raise ValueError("CircuitUnsatisfied")
    return t


# ---------------------------------------------------------------------------
# linearization_poly::compute
# ---------------------------------------------------------------------------
def barycentric_eval(evals, point, domain):         # proof.rs:1041-1088
    numerator = (pow(point, domain.size, Q) - 1) * domain.size_inv % Q
    acc = 0
    for i, ev in enumerate(evals):
        if ev % Q:
            den = (pow(domain.group_gen_inv, i, Q) * point - 1) % Q
            acc = (acc + fr_inv(den) * ev) % Q
    return acc * numerator % Q


def linearization_compute(prover, ch, z_ch, z_poly, ev, domain, t_polys, pub_inputs_dense):
    """linearization_poly.rs:168-264."""
    alpha, beta, gamma, range_ch, logic_ch, fixed_ch, var_ch = ch
    p = prover.pk.polys
    # arithmetic/proverkey.rs:73-111
    f1 = poly_scale(p["q_m"], ev["a"] * ev["b"] % Q)
    for poly, s in ((p["q_l"], ev["a"]), (p["q_r"], ev["b"]), (p["q_o"], ev["c"]),
                    (p["q_f"], ev["d"]), (p["q_c"], 1)):
        f1 = poly_add(f1, poly_scale(poly, s))
    f1 = poly_scale(f1, ev["q_arith"])
    # range/proverkey.rs:60-85
    f1 = poly_add(f1, poly_scale(p["q_range"], range_identity(
        range_ch, ev["a"], ev["b"], ev["c"], ev["d"], ev["d_w"]) * range_ch % Q))
    # logic/proverkey.rs:72-105
    f1 = poly_add(f1, poly_scale(p["q_logic"], logic_identity(
        logic_ch, ev["a"], ev["a_w"], ev["b"], ev["b_w"], ev["c"], ev["d"], ev["d_w"],
        ev["q_c"]) * logic_ch % Q))
    # fixed_base/proverkey.rs:103-159
    f1 = poly_add(f1, poly_scale(p["q_fixed_group_add"], fixed_identity(
        fixed_ch, ev["a"], ev["a_w"], ev["b"], ev["b_w"], ev["c"], ev["d"], ev["d_w"],
        ev["q_l"], ev["q_r"], ev["q_c"]) * fixed_ch % Q))
    # curve_addition/proverkey.rs:79-120
    f1 = poly_add(f1, poly_scale(p["q_variable_group_add"], var_identity(
        var_ch, ev["a"], ev["a_w"], ev["b"], ev["b_w"], ev["c"], ev["d"], ev["d_w"]) * var_ch % Q))
    pi_eval = barycentric_eval(pub_inputs_dense, z_ch, domain)     # :183-189
    f1 = poly_add(f1, [pi_eval])
    # permutation/proverkey.rs:127-269
    bz = beta * z_ch % Q
    a_ = ((ev["a"] + bz + gamma) * (ev["b"] + K1 * bz + gamma) % Q
          * (ev["c"] + K2 * bz + gamma) % Q * (ev["d"] + K3 * bz + gamma) % Q * alpha) % Q
    b_ = ((ev["a"] + beta * ev["s_sigma_1"] + gamma) * (ev["b"] + beta * ev["s_sigma_2"] + gamma) % Q
          * (ev["c"] + beta * ev["s_sigma_3"] + gamma) % Q * (beta * ev["z"] % Q) * alpha) % Q
    dom_z = EvaluationDomain(len(z_poly) - 1 - 2)                  # :162 new(z_poly.degree() - 2)
    l1 = dom_z.first_lagrange_at(z_ch)
    f2 = poly_add(poly_add(poly_scale(z_poly, a_), poly_scale(p["s_sigma_4"], (-b_) % Q)),
                  poly_scale(z_poly, l1 * alpha % Q * alpha % Q))
    n = domain.size
    z_n = pow(z_ch, n, Q)
    quot = poly_add(poly_add(poly_add(t_polys[0], poly_scale(t_polys[1], z_n)),
                             poly_scale(t_polys[2], pow(z_ch, 2 * n, Q))),
                    poly_scale(t_polys[3], pow(z_ch, 3 * n, Q)))
    quot = poly_scale(quot, (-(z_n - 1)) % Q)
    return poly_add(poly_add(f1, f2), quot)


# ---------------------------------------------------------------------------
# Prover::prove_inner (V3)
# ---------------------------------------------------------------------------
EVAL_ORDER = ["a", "b", "c", "d", "a_w", "b_w", "d_w", "q_arith", "q_c", "q_l", "q_r",
              "s_sigma_1", "s_sigma_2", "s_sigma_3", "z"]          # linearization_poly.rs:98-124
COMM_ORDER = ["a", "b", "c", "d", "z", "t_low", "t_mid", "t_high", "t_fourth",
              "w_z", "w_zw"]                                       # proof.rs:137-162


def blind_poly(witnesses, blinders, domain):        # prover.rs:139-152
    c = domain.ifft(witnesses)
    for i, b in enumerate(blinders):
        c[i] = (c[i] - b) % Q
        c.append(b)
    return poly_trim(c)


def permutation_vec(domain, wires, beta, gamma, sigma_ev):   # permutation.rs:213-294
    n = domain.size
    roots = domain.elements()
    ks = [1, K1, K2, K3]
    out, product = [], 1
    for i in range(n):
        out.append(product)
        if i + 1 < n:
            num = den = 1
            for k in range(4):
                num = num * (wires[k][i] + beta * roots[i] * ks[k] + gamma) % Q
                den = den * (wires[k][i] + beta * sigma_ev[k][i] + gamma) % Q
            assert den != 0
            product = product * num % Q * fr_inv(den) % Q
    return out


def prove(prover: Prover, rng, composer: Composer, msm=E.msm_naive, trace: dict | None = None, version: int = 3):
    """prove_inner (prover.rs:415-761).  Returns (proof_bytes, public_inputs)."""
    assert len(composer.constraints) == prover.constraints         # composer.rs:452-459
    size = prover.size
    domain = EvaluationDomain(prover.constraints)
    pk, ck = prover.pk, prover.ck
    tr = seed_transcript_v3(prover.label, prover.vk, prover.constraints, version)   # prove_with_version(V2 | V3), prover.rs:365-413
    pi_idx = composer.public_input_indexes()
    public_inputs = [composer.public_inputs[i] for i in pi_idx]
    dense_pi = [0] * size
    for i in pi_idx:
        dense_pi[i] = composer.public_inputs[i]
    for pi in public_inputs:
        tr.append_scalar(b"pi", pi)
    # round 1 (:444-479)
    W = composer.witnesses
    cols = [[0] * size for _ in range(4)]
    for i, g in enumerate(composer.constraints):
        cols[0][i], cols[1][i], cols[2][i], cols[3][i] = W[g.a], W[g.b], W[g.c], W[g.d]
    wire_blinders = [[rng.random_scalar(), rng.random_scalar()] for _ in range(4)]   # :154-161
    wire_polys = [blind_poly(cols[k], wire_blinders[k], domain) for k in range(4)]
    comm = {}
    for k, name in enumerate("abcd"):
        comm[name] = commit(ck, wire_polys[k], msm)
    for name in "abcd":
        tr.append_commitment(f"{name}_comm".encode(), comm[name])
    # round 2 (:481-505)
    beta = tr.challenge_scalar(b"beta")
    tr.append_scalar(b"beta", beta)
    gamma = tr.challenge_scalar(b"gamma")
    perm = permutation_vec(domain, cols, beta, gamma, prover.sigma_evaluations)
    z_poly = blind_poly(perm, [rng.random_scalar() for _ in range(3)], domain)      # :133-135,503
    comm["z"] = commit(ck, z_poly, msm)
    tr.append_commitment(b"z_comm", comm["z"])
    # round 3 (:507-589)
    alpha = tr.challenge_scalar(b"alpha")
    range_ch = tr.challenge_scalar(b"range separation challenge")
    logic_ch = tr.challenge_scalar(b"logic separation challenge")
    fixed_ch = tr.challenge_scalar(b"fixed base separation challenge")
    var_ch = tr.challenge_scalar(b"variable base separation challenge")
    ch = (alpha, beta, gamma, range_ch, logic_ch, fixed_ch, var_ch)
    pi_poly = poly_trim(domain.ifft(dense_pi))
    t_poly = quotient_compute(prover, z_poly, wire_polys, pi_poly, ch)
    t_poly = t_poly + [0] * max(0, 3 * size + 1 - len(t_poly))
    n = size
    t_low, t_mid, t_high, t_4 = (list(t_poly[0:n]), list(t_poly[n:2 * n]),
                                 list(t_poly[2 * n:3 * n]), list(t_poly[3 * n:]))
    b12, b13, b14 = rng.random_scalar(), rng.random_scalar(), rng.random_scalar()     # :553-555
    t_low.append(b12)
    t_mid[0] = (t_mid[0] - b12) % Q
    t_mid.append(b13)
    t_high[0] = (t_high[0] - b13) % Q
    t_high.append(b14)
    t_4[0] = (t_4[0] - b14) % Q
    t_polys = [poly_trim(x) for x in (t_low, t_mid, t_high, t_4)]
    for name, p in zip(("t_low", "t_mid", "t_high", "t_fourth"), t_polys):
        comm[name] = commit(ck, p, msm)
    for name in ("t_low", "t_mid", "t_high", "t_fourth"):
        tr.append_commitment(f"{name}_comm".encode(), comm[name])
    # round 4 (:591-676)
    z_ch = tr.challenge_scalar(b"z_challenge")
    zw = z_ch * domain.group_gen % Q
    P = pk.polys
    ev = {
        "a": poly_eval(wire_polys[0], z_ch), "b": poly_eval(wire_polys[1], z_ch),
        "c": poly_eval(wire_polys[2], z_ch), "d": poly_eval(wire_polys[3], z_ch),
        "s_sigma_1": poly_eval(P["s_sigma_1"], z_ch), "s_sigma_2": poly_eval(P["s_sigma_2"], z_ch),
        "s_sigma_3": poly_eval(P["s_sigma_3"], z_ch), "z": poly_eval(z_poly, zw),
        "a_w": poly_eval(wire_polys[0], zw), "b_w": poly_eval(wire_polys[1], zw),
        "d_w": poly_eval(wire_polys[3], zw),
        "q_arith": poly_eval(P["q_arith"], z_ch), "q_c": poly_eval(P["q_c"], z_ch),
        "q_l": poly_eval(P["q_l"], z_ch), "q_r": poly_eval(P["q_r"], z_ch),
    }
    for lab in ("a", "b", "c", "d", "s_sigma_1", "s_sigma_2", "s_sigma_3", "z"):
        tr.append_scalar(f"{lab}_eval".encode(), ev[lab])
    for lab in ("a_w", "b_w", "d_w", "q_arith", "q_c", "q_l", "q_r"):
        tr.append_scalar(f"{lab}_eval".encode(), ev[lab])
    # round 5 (:678-739)
    v_ch = tr.challenge_scalar(b"v_challenge")
    r_poly = linearization_compute(prover, ch, z_ch, z_poly, ev, domain, t_polys, dense_pi)
    agg = compute_aggregate_witness(
        [r_poly, *wire_polys, P["s_sigma_1"], P["s_sigma_2"], P["s_sigma_3"],
         P["q_arith"], P["q_c"], P["q_l"], P["q_r"]], z_ch, v_ch)
    comm["w_z"] = commit(ck, agg, msm)
    v_w = tr.challenge_scalar(b"v_w_challenge")
    sh = compute_aggregate_witness([z_poly, wire_polys[0], wire_polys[1], wire_polys[3]], zw, v_w)
    comm["w_zw"] = commit(ck, sh, msm)
    if trace is not None:
        trace.update(dict(wire_polys=wire_polys, z_poly=z_poly, t_polys=t_polys, ev=ev, comm=comm,
                          challenges=dict(alpha=alpha, beta=beta, gamma=gamma, range=range_ch,
                                          logic=logic_ch, fixed=fixed_ch, var=var_ch, z=z_ch,
                                          v=v_ch, v_w=v_w),
                          r_poly=r_poly, agg=agg, shifted_agg=sh, perm=perm, pi_poly=pi_poly,
                          blinders=dict(wires=wire_blinders, t=(b12, b13, b14))))
    out = b"".join(g1_compress(comm[k]) for k in COMM_ORDER)
    out += b"".join(fr_to_bytes(ev[k]) for k in EVAL_ORDER)
    assert len(out) == 1008
    return out, public_inputs
