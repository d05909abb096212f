"""GPU parity: device-resident Prover::prove (plonk_prover_prove through the C-ABI) vs the
oracle restatement of reference src/compiler/prover.rs:415-761 — bit-identical 1008-byte
Proof on the same SRS, witness and blinders; and against the reference's own KAT digest."""
import hashlib
import random

import pytest

from oracle import bls12_381 as E
from oracle import plonk as O
from oracle.rng import StdRng
from tests.test_oracle_kat import KAT_DIGEST

pytestmark = pytest.mark.gpu
Q = E.Q


class FixedBlinders:
    """RNG stand-in that records / replays BlsScalar::random draws."""

    def __init__(self, rng):
        self.rng, self.drawn = rng, []

    def random_scalar(self):
        s = self.rng.random_scalar()
        self.drawn.append(s)
        return s


def gpu_prover(ctx, oprover, vk=False):
    import plonk_amd
    ctx.srs_load(oprover.ck)
    vkb = None
    if vk:
        vkb = b"".join(E.g1_compress(oprover.vk[name]) for name in plonk_amd.POLY_ORDER)
    p = plonk_amd.Prover(ctx, oprover.constraints, oprover.label, oprover.pk.polys, vkb)
    assert p.describe()["quotient_domain"] == (ctx.get_config().quotient_domain if p.size >= 8 else 8)   # the switch was honoured
    return p


def wires_of(composer, size):
    W = composer.witnesses
    cols = [[0] * size for _ in range(4)]
    for i, g in enumerate(composer.constraints):
        cols[0][i], cols[1][i], cols[2][i], cols[3][i] = W[g.a], W[g.b], W[g.c], W[g.d]
    return cols


def both_prove(ctx, pp, label, build_circuit, seed):
    oprover = O.compile_circuit(pp, label, build_circuit(), msm=E.msm_pippenger)
    rec = FixedBlinders(StdRng.seed_from_u64(seed))
    comp = build_circuit()
    expected, pis = O.prove(oprover, rec, comp, msm=E.msm_pippenger)
    assert len(rec.drawn) == 14
    gp = gpu_prover(ctx, oprover)
    got = gp.prove(wires_of(comp, oprover.size), dict(comp.public_inputs), rec.drawn)
    gp.close()
    return got, expected, oprover


@pytest.fixture(scope="module")
def ctx():
    import plonk_amd
    c = plonk_amd.Context(0)
    yield c
    c.close()


@pytest.fixture(autouse=True, params=["quotient-4n", "quotient-8n"])
def quotient_domain(request, ctx):
    """Every test runs on both quotient domains: the default 4n (+ de-aliasing by the low
    coefficients, prover.hip quotient_low) and the reference-shaped 8n (quotient_poly.rs:96-137) —
    plonk_gpu_config.quotient_domain of the module's context (provers built afterwards follow it;
    gpu_prover asserts that they did)."""
    from conftest import configure
    configure(ctx, quotient_domain=8 if request.param == "quotient-8n" else 4)
    return request.param


def test_deterministic_v3_proof_matches_base_digest(ctx, kat_setup):
    """reference prover.rs:1132-1162 through the HIP prover: SRS seed 0x9235e700, proving
    RNG seed 0x9235e701, MinimalCircuit -> blake2b(proof bytes) == literal at :1151-1158."""
    import plonk_amd
    _, oprover, circuit = kat_setup
    gp = gpu_prover(ctx, oprover)
    # Compiler::preprocess commitments computed by the GPU MSM (compiler.rs:213-232)
    assert gp.vk_commitments() == b"".join(E.g1_compress(oprover.vk[n]) for n in plonk_amd.POLY_ORDER)
    rng = StdRng.seed_from_u64(0x9235E701)
    blinders = [rng.random_scalar() for _ in range(14)]
    comp = circuit()
    proof = gp.prove(wires_of(comp, oprover.size), {}, blinders)
    assert len(proof) == 1008
    assert hashlib.blake2b(proof).digest() == KAT_DIGEST
    # and with the verifier key passed in instead of recomputed
    gp2 = gpu_prover(ctx, oprover, vk=True)
    assert gp2.prove(wires_of(comp, oprover.size), {}, blinders) == proof
    gp.close()
    gp2.close()


def test_legacy_v2_transcript_matches_the_oracle(ctx, kat_setup):
    """plonk_prover_set_version(2) = Prover::prove_with_version(V2) (prover.rs:365-413): only the transcript seeding differs
    (the label s_sigma_4 carries s_sigma_1's commitment, widget.rs:224-228); bytes equal the C oracle's V2 proof, setting the
    version back restores the KAT digest, other values are refused."""
    import plonk_amd
    from oracle import cbind
    _, oprover, circuit = kat_setup
    gp = gpu_prover(ctx, oprover)
    rng = StdRng.seed_from_u64(0x9235E701)
    blinders = [rng.random_scalar() for _ in range(14)]
    comp = circuit()
    wires = wires_of(comp, oprover.size)
    from tests import circuits as C
    cp = cbind.CProver(oprover.constraints, oprover.label, {k: C.fr_bytes(v) for k, v in oprover.pk.polys.items()},
                       b"".join(E.g1_to_raw96(p) for p in oprover.ck))
    cp.set_version(2)
    want_v2 = cp.prove([C.fr_bytes(w) for w in C.wires_of(comp, oprover.size)], [], b"", C.fr_bytes(blinders))
    gp.set_version(2)
    assert gp.prove(wires, {}, blinders) == want_v2
    gp.set_version(3)
    assert hashlib.blake2b(gp.prove(wires, {}, blinders)).digest() == KAT_DIGEST
    with pytest.raises(plonk_amd.PlonkError):
        gp.set_version(1)
    gp.close()


def arithmetic_circuit(ngates, seed, with_pi=True):
    def build():
        r = random.Random(seed)
        c = O.Composer()
        ws = [c.append_witness(r.randrange(Q)) for _ in range(8)]
        while len(c.constraints) < ngates - (2 if with_pi else 0):
            a, b, d = r.choice(ws), r.choice(ws), r.choice(ws)
            if r.random() < 0.5:
                ws.append(c.gate_add(a, b, d, q_l=r.randrange(Q), q_r=r.randrange(1, 50), q_f=r.randrange(3),
                                     q_c=r.randrange(Q)))
            else:
                ws.append(c.gate_mul(a, b, d, q_m=r.randrange(1, Q), q_f=r.randrange(2), q_c=r.randrange(100)))
        if with_pi:
            # append_public (composer.rs:377-389): -w + PI = 0
            for _ in range(2):
                v = r.randrange(Q)
                w = c.append_witness(v)
                c.append_gate(O.Gate(a=w, q_l=Q - 1, pi=v))
        return c
    return build


@pytest.mark.parametrize("ngates,seed", [(16, 1), (50, 2), (64, 3), (200, 4)])
def test_random_arithmetic_circuits_bit_exact(ctx, ngates, seed):
    pp = O.srs_setup(300, StdRng.seed_from_u64(77), keep=300)
    got, expected, _ = both_prove(ctx, pp, b"gpu-parity", arithmetic_circuit(ngates, seed), 1000 + seed)
    assert got == expected


def widget_circuit():
    """Rows activating every selector family on satisfying assignments: all-zero rows satisfy
    the range/logic/fixed-base/variable-base identities; one non-trivial range row (quads)."""
    c = O.Composer()
    z = 0
    one = 1
    # non-trivial range gate: d=1, c=4d+2, b=4c+3, a=4b+0, next d = 4a+1
    d0 = c.append_witness(1)
    c0 = c.append_witness(6)
    b0 = c.append_witness(27)
    a0 = c.append_witness(108)
    dn = c.append_witness(433)
    c.append_custom_gate(O.Gate(a=a0, b=b0, c=c0, d=d0, q_range=1))
    c.append_gate(O.Gate(a=z, b=z, c=z, d=dn))                       # next row carries d_next (0 = 0 gate)
    for sel in ("q_range", "q_logic", "q_fixed_group_add", "q_variable_group_add"):
        g = O.Gate(a=z, b=z, c=z, d=z)
        setattr(g, sel, 1 if sel != "q_logic" else Q - 1)
        if sel == "q_logic":
            g.q_c = Q - 1                                            # Constraint::logic_xor (constraint.rs:213-217)
        if sel == "q_fixed_group_add":
            g.q_l, g.q_r, g.q_c = 5, 9, 45                           # x_beta, y_beta, xy_beta
        c.append_custom_gate(g)
        c.append_gate(O.Gate(a=z, b=z, c=z, d=z))
    w = c.append_witness(11)
    c.assert_equal_constant(w, 11)
    _ = one
    return c


def test_all_widget_selectors_bit_exact(ctx):
    pp = O.srs_setup(64, StdRng.seed_from_u64(5), keep=64)
    got, expected, oprover = both_prove(ctx, pp, b"widgets", widget_circuit, 4242)
    assert all(oprover.pk.polys[k] for k in ("q_range", "q_logic", "q_fixed_group_add", "q_variable_group_add"))
    assert got == expected


def test_unsatisfied_circuit_is_rejected(ctx):
    """reference tests/common/mod.rs:60-80: forged witnesses -> Error::CircuitUnsatisfied."""
    import plonk_amd
    pp = O.srs_setup(64, StdRng.seed_from_u64(6), keep=64)
    build = arithmetic_circuit(20, 9, with_pi=False)
    oprover = O.compile_circuit(pp, b"unsat", build(), msm=E.msm_pippenger)
    comp = build()
    cols = wires_of(comp, oprover.size)
    cols[2][7] = (cols[2][7] + 1) % Q            # break one output wire
    gp = gpu_prover(ctx, oprover)
    with pytest.raises(plonk_amd.CircuitUnsatisfied):   # exactly Error::CircuitUnsatisfied (quotient_poly.rs:132)
        gp.prove(cols, {}, list(range(1, 15)))
    # the prover stays usable and still proves the honest witness bit-exactly afterwards
    rec = FixedBlinders(StdRng.seed_from_u64(31))
    honest = build()
    expected, _ = O.prove(oprover, rec, honest, msm=E.msm_pippenger)
    assert gp.prove(wires_of(honest, oprover.size), {}, rec.drawn) == expected
    gp.close()
    comp.witnesses[comp.constraints[7].c] = cols[2][7]
    with pytest.raises((ValueError, AssertionError)):
        O.prove(oprover, FixedBlinders(StdRng.seed_from_u64(1)), comp, msm=E.msm_pippenger)


def test_prover_refuses_to_prove_after_its_srs_was_replaced(ctx):
    """A prover is bound to the commit key its context held when it was built: replacing the SRS
    (plonk_srs_load) makes older provers return PLONK_ERR_STATE instead of a silently invalid proof."""
    import plonk_amd
    pp = O.srs_setup(64, StdRng.seed_from_u64(6), keep=64)
    build = arithmetic_circuit(20, 9, with_pi=False)
    oprover = O.compile_circuit(pp, b"stale", build(), msm=E.msm_pippenger)
    gp = gpu_prover(ctx, oprover)
    cols = wires_of(build(), oprover.size)
    first = gp.prove(cols, {}, list(range(1, 15)))
    other = O.srs_setup(64, StdRng.seed_from_u64(7), keep=64)
    ctx.srs_load(other)
    with pytest.raises(plonk_amd.PlonkError) as ei:
        gp.prove(cols, {}, list(range(1, 15)))
    assert ei.value.code == -7
    gp.close()
    gp2 = gpu_prover(ctx, oprover)          # rebuilt on the original key: same proof again
    assert gp2.prove(cols, {}, list(range(1, 15))) == first
    gp2.close()


def test_host_wire_schedule_is_the_one_asked_for(ctx, kat_setup):
    """Run by the variant children of tests/test_gpu_msm_variants.py (PLONK_TEST_EXPECT_WIRE_LAUNCHES): with PLONK_WIRE_BY_COLUMN=1 / 2
    a proof from HOST wire columns commits to them in 3 (a, b, c + d) / 4 launches of msm_batch_device's phase 1 at EVERY size
    (the default takes that schedule from 2^19 gates on), and plonk_prover_describe reports what the last proof did — a switch
    the library ignores fails here instead of re-testing the grouped launch.  Same bytes as the resident-column proof."""
    import os
    want = os.environ.get("PLONK_TEST_EXPECT_WIRE_LAUNCHES")
    if want is None:
        pytest.skip("only meaningful under a forced schedule")
    import plonk_amd
    from oracle import cbind
    from tests import circuits as C
    comp = C.big_widget_circuit(1 << 12, seed=77)()
    case = C.compile_fast(comp, b"wire-schedule")
    srs = C.synthetic_srs(case["size"] + 7)
    ctx.srs_load_bytes(srs, len(srs) // 96)
    gp = plonk_amd.Prover(ctx, case["constraints"], case["label"], case["polys"])
    bl = C.blinders(random.Random(5).randrange(1 << 30))
    assert gp.describe()["wire_group_launches"] == 0
    n = case["size"]
    wbuf = ctx.alloc(4 * 32 * n)
    for k in range(4):
        wbuf.upload(case["wires"][k], 32 * n * k)
    resident = gp.prove_dev(wbuf.ptr, case["pi"], bl)
    assert gp.describe()["wire_group_launches"] == 1
    host = gp.prove_host_bytes(case["wires"], case["pi"], bl)
    assert gp.describe()["wire_group_launches"] == int(want)
    assert host == resident
    cp = cbind.CProver(case["constraints"], case["label"], case["polys"], srs)
    assert resident == cp.prove(case["wires"], case["pi_idx"], case["pi_val"], bl)
    cp.close()
    wbuf.free()
    gp.close()


def test_host_time_slots_count_the_five_synchronisations_of_a_proof(ctx, kat_setup):
    """plonk_profile_read slots 8-10 (round 6): per proof four commitment groups are finished on the host (slot 8), five
    synchronisations return and are followed by a launch (slot 9), five waits are timed (slot 10); the times are host wall
    time inside prove() and cannot exceed it."""
    import time
    import plonk_amd
    from tests import circuits as C
    comp = C.big_widget_circuit(1 << 11, seed=78)()
    case = C.compile_fast(comp, b"host-slots")
    srs = C.synthetic_srs(case["size"] + 7)
    ctx.srs_load_bytes(srs, len(srs) // 96)
    gp = plonk_amd.Prover(ctx, case["constraints"], case["label"], case["polys"])
    bl = C.blinders(11)
    n = case["size"]
    wbuf = ctx.alloc(4 * 32 * n)
    for k in range(4):
        wbuf.upload(case["wires"][k], 32 * n * k)
    first = gp.prove_dev(wbuf.ptr, case["pi"], bl)
    # round 6: the Lagrange-basis table carries a third blinding point and z is committed from its evaluations, unless
    # PLONK_Z_COMMIT=coeff (a variant child) keeps the coefficient form — the library reports which
    import os
    if os.environ.get("PLONK_WIRE_COMMIT") != "coeff":
        assert gp.describe()["lagrange_points"] == n + (2 if os.environ.get("PLONK_Z_COMMIT") == "coeff" else 3)
    ctx.profile(True)
    ctx.profile_reset()
    t0 = time.perf_counter()
    for _ in range(3):
        assert gp.prove_dev(wbuf.ptr, case["pi"], bl) == first
    ctx.sync()
    wall_ms = (time.perf_counter() - t0) * 1e3
    got = {slot: ctx.profile_read(slot) for slot in (8, 9, 10, 11)}
    ctx.profile(False)
    assert [got[s][1] for s in (8, 9, 10, 11)] == [12, 15, 15, 12], got
    # slot 11 sums the helper threads of every group: groups of 4, 1, 4, 2 commitments — the group of one has none.  A
    # PLONK_HOST_THREADS the library ignored fails here (tests/test_gpu_msm_variants.py runs this file with 0 and 7).
    import os
    env = os.environ.get("PLONK_HOST_THREADS")
    workers = int(env) if env is not None else (3 if len(os.sched_getaffinity(0)) >= 8 else 0)
    assert got[11][0] == 3 * 3 * workers, (got[11], workers)
    assert 0 < got[8][0] <= got[9][0] < wall_ms and 0 < got[10][0] < wall_ms, (got, wall_ms)
    gp.close()
    wbuf.free()
