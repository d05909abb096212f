"""Widget identities pinned to the gates' semantics (tests/widget_circuits.py).
CPU: the oracle proves the honest semantic witnesses and the proof verifies; corrupting one
quad / one curve coordinate is rejected.  GPU: bit-exact against the oracle on both quotient
domains (the 4n path evaluates the same widget formulas in F[X]/(X^7) on the host)."""
import pytest

from oracle import bls12_381 as E
from oracle import plonk as O
from oracle.rng import StdRng
from oracle.verifier import verify_with_tau
from widget_circuits import jj_add, jj_base, on_curve, semantic_widget_circuit

Q = E.Q


def test_jubjub_helpers():
    b = jj_base()
    assert on_curve(b) and on_curve(jj_add(b, b)) and jj_add(b, (0, 1)) == b
    assert jj_add(jj_add(b, b), b) == jj_add(b, jj_add(b, b))


def _setup(seed):
    build = semantic_widget_circuit(seed)
    comp = build()
    n = len(comp.constraints)
    rng = StdRng.seed_from_u64(0xABCD)
    tau = rng.random_scalar()                       # srs_setup draws tau first (srs.rs:61-100)
    pp = O.srs_setup(2 * n + 16, StdRng.seed_from_u64(0xABCD), keep=2 * n + 16)
    prover = O.compile_circuit(pp, b"widgets-semantic", build(), msm=E.msm_pippenger)
    return build, prover, pp, tau


def test_oracle_accepts_semantic_witnesses_and_rejects_corrupted_ones():
    build, prover, pp, tau = _setup(3)
    assert all(prover.pk.polys[k] for k in ("q_range", "q_logic", "q_fixed_group_add", "q_variable_group_add"))
    comp = build()
    proof, pis = O.prove(prover, StdRng.seed_from_u64(5), comp, msm=E.msm_pippenger)
    assert verify_with_tau(proof, prover.vk, b"widgets-semantic", prover.constraints, dict(comp.public_inputs), tau, pp[0])
    # corrupt one witness inside each widget family: no longer a polynomial quotient
    for sel in ("q_range", "q_logic", "q_fixed_group_add", "q_variable_group_add"):
        bad = build()
        row = next(i for i, g in enumerate(bad.constraints) if getattr(g, sel))
        g = bad.constraints[row + 1]                # the row that carries the "next" values
        bad.witnesses[g.d] = (bad.witnesses[g.d] + 1) % Q
        with pytest.raises((ValueError, AssertionError)):
            O.prove(prover, StdRng.seed_from_u64(5), bad, msm=E.msm_pippenger)


@pytest.mark.gpu
@pytest.mark.parametrize("domain", ["quotient-4n", "quotient-8n"])
@pytest.mark.parametrize("seed", [3, 4])
def test_hip_prover_bit_exact_on_semantic_widget_circuits(monkeypatch, domain, seed):
    import plonk_amd
    from test_gpu_prover import FixedBlinders, wires_of
    build, prover, pp, tau = _setup(seed)
    rec = FixedBlinders(StdRng.seed_from_u64(100 + seed))
    comp = build()
    expected, _ = O.prove(prover, rec, comp, msm=E.msm_pippenger)
    ctx = plonk_amd.Context(0, plonk_amd.GpuConfig(quotient_domain=8 if domain == "quotient-8n" else 4))
    ctx.srs_load(prover.ck)
    gp = plonk_amd.Prover(ctx, prover.constraints, prover.label, prover.pk.polys)
    assert gp.describe()["quotient_domain"] == (8 if domain == "quotient-8n" else 4)
    got = gp.prove(wires_of(comp, prover.size), dict(comp.public_inputs), rec.drawn)
    assert got == expected
    # a corrupted quad inside the logic gadget must be refused by the device prover as well
    bad = build()
    row = next(i for i, g in enumerate(bad.constraints) if g.q_logic)
    bad.witnesses[bad.constraints[row + 1].d] += 1
    with pytest.raises(plonk_amd.CircuitUnsatisfied):   # exactly Error::CircuitUnsatisfied (quotient_poly.rs:132)
        gp.prove(wires_of(bad, prover.size), dict(bad.public_inputs), rec.drawn)
    gp.close()
    ctx.close()


def test_host_quotient_low_matches_oracle_quotient():
    """widgets.hpp `quotient_low` (the de-aliasing input of the 4n quotient domain): evaluating the whole
    numerator formula — every widget included — in F[X]/(X^7) on the lowest 7 coefficients of each
    polynomial must give the 7 lowest coefficients of the oracle's quotient t (computed the reference's
    way, on the 8n coset).  Runs on the CPU through the host harness."""
    import ctypes

    import plonk_amd
    from test_field_host import build_host_lib
    lib = build_host_lib()
    from oracle.fft import EvaluationDomain
    from widget_circuits import EDWARDS_D
    build, prover, pp, tau = _setup(3)
    tr = {}
    O.prove(prover, StdRng.seed_from_u64(77), build(), msm=E.msm_pippenger, trace=tr)
    n = prover.size
    ch = tr["challenges"]
    dom = EvaluationDomain(n)

    def low7(poly):
        return (list(poly) + [0] * 7)[:7]

    mont = plonk_amd.fr_to_bytes_mont
    key_low = b"".join(mont(low7(prover.pk.polys[name])) for name in plonk_amd.POLY_ORDER)
    has = bytes(1 if prover.pk.polys[name] else 0 for name in plonk_amd.POLY_ORDER[:11])
    lows = b"".join(mont(low7(p)) for p in (*tr["wire_polys"], tr["z_poly"], tr["pi_poly"]))
    chs = mont([ch["alpha"], ch["beta"], ch["gamma"], ch["range"], ch["logic"], ch["fixed"], ch["var"], EDWARDS_D,
                dom.group_gen, pow(n, -1, Q)])
    out = ctypes.create_string_buffer(7 * 32)
    lib.h_quotient_low(key_low, has, lows, chs, out)
    got = plonk_amd.fr_from_bytes_mont(out.raw)
    assert got == low7(tr["t_polys"][0])
    assert any(got)
