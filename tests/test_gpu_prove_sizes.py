"""GPU parity at the BASELINE sizes: the 1008 Proof bytes of the HIP prover (through the C-ABI)
must equal, byte for byte, those of the C restatement of the reference's prove_inner
(oracle/c/oracle_prove.c — pinned to the reference KAT digest and to the big-int oracle by
tests/test_oracle_c_prove.py) on the same SRS, circuit, witness and blinders.

Circuits: every widget family (range, logic XOR/AND, fixed-base, curve addition) with honest
non-trivial witnesses, random arithmetic gates and public inputs (tests/circuits.py), on both
quotient domains.  Sizes 2^12 / 2^13 / 2^16 exercise the code paths a 256-gate circuit never reaches:
multi-workgroup scans (scan_block / scan_totals / scan_apply), batch inversion over more than one
4096-element workgroup, the tree combine of eval_kernel, 2- and 3-pass NTTs inside the prover
(8n = 2^19 at 2^16 gates) and MSM buckets with more than one slice.  BASELINE config 2:
"2^16 gates, 1xMI355X, HIP NTT + HIP Pippenger MSM, bit-exact Proof vs CPU"."""
import hashlib

import pytest

from conftest import configure
from oracle import cbind
from tests import circuits as C

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import plonk_amd
    c = plonk_amd.Context(0)
    yield c
    c.close()


_cases = {}


def case_for(log_n):
    """circuit + SRS + C-oracle prover, shared by the two quotient domains"""
    if log_n not in _cases:
        _cases.clear()   # one size resident at a time (the 8n key evaluations of the oracle are large)
        case = C.compile_fast(C.big_widget_circuit(1 << log_n, seed=100 + log_n)(), b"size-parity")
        assert case["size"] == 1 << log_n and len(case["pi_idx"]) >= 2
        srs = C.synthetic_srs(case["size"] + 7)
        cp = cbind.CProver(case["constraints"], case["label"], case["polys"], srs)
        _cases[log_n] = (case, srs, cp)
    return _cases[log_n]


def gpu_proof(ctx, case, srs, blinders_mont, vk=None, expect=None, host_too=False):
    """expect: fields of plonk_prover_describe the prover must report (the switch under test was honoured);
    host_too: also prove from HOST wire columns (plonk_prover_prove) and require the same bytes — above 2^18 gates that path
    commits column by column as the copies land (round 6: msm_batch_device phases, prover.hip by_column)"""
    import plonk_amd
    ctx.srs_load_bytes(srs, len(srs) // 96)
    gp = plonk_amd.Prover(ctx, case["constraints"], case["label"], case["polys"], vk)
    try:
        info = gp.describe()
        assert info["quotient_domain"] == ctx.get_config().quotient_domain
        for k, v in (expect or {}).items():
            assert info[k] == v, (k, info)
        got_vk = gp.vk_commitments()
        wbuf = ctx.alloc(4 * 32 * case["size"])
        for k in range(4):
            wbuf.upload(case["wires"][k], 32 * case["size"] * k)
        proof = gp.prove_dev(wbuf.ptr, case["pi"], blinders_mont)
        wbuf.free()
        if host_too:
            assert gp.prove_host_bytes(case["wires"], case["pi"], blinders_mont) == proof
        return proof, got_vk
    finally:
        gp.close()


@pytest.mark.parametrize("domain", ["quotient-4n", "quotient-8n"])
@pytest.mark.parametrize("log_n", [12, 13, 16])
def test_proof_bytes_equal_c_oracle(ctx, monkeypatch, log_n, domain):
    configure(ctx, quotient_domain=8 if domain == "quotient-8n" else 4, wire_commit=0)
    case, srs, cp = case_for(log_n)
    for name in ("q_range", "q_logic", "q_fixed_group_add", "q_variable_group_add"):
        assert case["polys"][name], name          # every selector family is active
    bl = C.blinders(7000 + log_n)
    expected = cp.prove(case["wires"], case["pi_idx"], case["pi_val"], bl)
    got, vk = gpu_proof(ctx, case, srs, bl)
    assert vk == cp.vk()                           # Compiler::preprocess commitments (compiler.rs:213-232)
    assert got == expected, (hashlib.blake2b(got).hexdigest()[:16], hashlib.blake2b(expected).hexdigest()[:16])


@pytest.mark.parametrize("domain", ["quotient-4n", "quotient-8n"])
def test_unsatisfied_witness_is_circuit_unsatisfied_exactly(ctx, monkeypatch, domain):
    """reference quotient_poly.rs:132 returns Error::CircuitUnsatisfied and nothing else; the shim maps
    PLONK_ERR_UNSAT to it.  One corrupted wire value in a widget row and one in an arithmetic row."""
    import plonk_amd
    configure(ctx, quotient_domain=8 if domain == "quotient-8n" else 4, wire_commit=0)
    case, srs, cp = case_for(12)
    ctx.srs_load_bytes(srs, len(srs) // 96)
    gp = plonk_amd.Prover(ctx, case["constraints"], case["label"], case["polys"], cp.vk())
    n = case["size"]
    wbuf = ctx.alloc(4 * 32 * n)
    for col, row in ((2, 9), (0, n - 3), (3, 1000)):
        for k in range(4):
            w = bytearray(case["wires"][k])
            if k == col:
                w[32 * row] ^= 1
            wbuf.upload(bytes(w), 32 * n * k)
        with pytest.raises(plonk_amd.CircuitUnsatisfied):
            gp.prove_dev(wbuf.ptr, case["pi"], C.blinders(3))
    # a wrong public input is an unsatisfied circuit too
    for k in range(4):
        wbuf.upload(case["wires"][k], 32 * n * k)
    bad_pi = dict(case["pi"])
    bad_pi[case["pi_idx"][0]] = (bad_pi[case["pi_idx"][0]] + 1) % C.Q
    with pytest.raises(plonk_amd.CircuitUnsatisfied):
        gp.prove_dev(wbuf.ptr, bad_pi, C.blinders(3))
    # and the prover still proves the honest witness bit-exactly afterwards
    bl = C.blinders(4)
    assert gp.prove_dev(wbuf.ptr, case["pi"], bl) == cp.prove(case["wires"], case["pi_idx"], case["pi_val"], bl)
    wbuf.free()
    gp.close()


@pytest.mark.slow
@pytest.mark.parametrize("profile", ["bench-like", "widgets"])
def test_side_workloads_equal_c_oracle_2p20(ctx, monkeypatch, profile):
    """The two other workloads bench.py times at 2^20 gates (`prove_ms_bench_like`: half of the wire values < 4, the skewed
    digits of a real witness; `prove_ms_all_widgets_pi`: every widget family + public inputs): the whole proof against the
    C oracle at the size that is timed, not only at 2^13 / 2^17."""
    import bench_circuits as BC
    configure(ctx, quotient_domain=4, wire_commit=0)
    log_n = 20
    n = 1 << log_n
    _cases.clear()
    if profile == "widgets":
        wires, cols, pi = BC.widget_circuit(log_n)
        trivial = {}
    else:
        wires, cols, trivial = BC.arithmetic_circuit(log_n, profile)
        pi = {}
    threads = cbind.max_threads()
    polys = {k: C.fr_bytes(v) for k, v in trivial.items()}
    for name, raw in cols.items():
        polys[name] = cbind.ntt_bytes(raw, log_n, True, False, n, threads)
    srs = C.synthetic_srs(n + 7)
    idx = sorted(pi)
    case = dict(constraints=n, size=n, label=b"bench", polys=polys, wires=wires, pi=pi, pi_idx=idx,
                pi_val=C.fr_bytes([pi[i] for i in idx]))
    bl = C.blinders(2021)
    got, vk = gpu_proof(ctx, case, srs, bl, host_too=True)   # bench-like: heavy buckets in every column's own launch
    cp = cbind.CProver(n, b"bench", polys, srs, vk48=vk, threads=threads)
    # the oracle takes its eleven commitments from the key's trapdoor — [g p(tau)] G is the group element the MSM over
    # [g tau^i] G returns (tests/test_oracle_c_prove.py::test_trapdoor_commitments_are_the_msm_commitments), as the 2^22 test
    # does: transforms, quotient and every O(n) pass remain the oracle's own.  The CPU MSMs (14 of its 22 s) run in full in
    # test_proof_bytes_equal_c_oracle_2p20 on the dense workload and at 2^12 … 2^19 above.
    tau, g = 0x5EED0000 * 0x9E3779B97F4A7C15 % C.Q, 0xA5A5A5A5DEADBEEF      # = tests/circuits.py synthetic_srs
    cp.set_trapdoor(C.fr_bytes([tau]), C.fr_bytes([g]))
    expected = cp.prove(wires, idx, case["pi_val"], bl)
    cp.close()
    assert got == expected


@pytest.mark.slow
@pytest.mark.parametrize("domain", ["quotient-4n"])   # the 8n domain is compared at 2^12 .. 2^16 above
def test_proof_bytes_equal_c_oracle_2p20(ctx, monkeypatch, domain):
    """BASELINE config 3 (2^20 gates): the bench circuit of bench.py (dense arithmetic profile) and the
    whole 1008-byte proof against the C oracle run on the host cores (about a minute)."""
    import bench
    import plonk_amd
    configure(ctx, quotient_domain=8 if domain == "quotient-8n" else 4, wire_commit=0)
    log_n = 20
    n = 1 << log_n
    key = ("bench", log_n)
    if key not in _cases:
        _cases.clear()
        wires, cols, trivial = bench.synth_circuit(log_n)
        polys = {k: C.fr_bytes(v) for k, v in trivial.items()}
        for name, raw in cols.items():
            polys[name] = cbind.ntt_bytes(raw, log_n, True, False, n)
        srs = C.synthetic_srs(n + 7)
        _cases[key] = (wires, polys, srs)
    wires, polys, srs = _cases[key]
    case = dict(constraints=n, size=n, label=b"bench", polys=polys, wires=wires, pi={}, pi_idx=[], pi_val=b"")
    bl = C.blinders(2020)
    got, vk = gpu_proof(ctx, case, srs, bl, host_too=True)
    cp = cbind.CProver(n, b"bench", polys, srs, vk48=vk)   # VK commitments are compared at 2^12..2^16; here they seed both transcripts
    expected = cp.prove(wires, [], b"", bl)
    cp.close()
    assert got == expected


@pytest.mark.slow
@pytest.mark.parametrize("log_n", [18, 19])
def test_proof_bytes_equal_c_oracle_at_the_layout_crossover(ctx, monkeypatch, log_n):
    """The two sizes either side of the table / bucket crossover (round 4: 2^18 + 64 terms): 2^18 gates run window rows, 2^15
    buckets, 32-entry slices in order of length with single-slice buckets written by their lane and the rest summed by a quad
    per bucket; 2^19 gates run bit-position rows and 2^19 buckets with ~12 entries per bucket.  bench.py's dense circuit, the
    whole proof against the C oracle."""
    import bench
    configure(ctx, quotient_domain=4, wire_commit=0)
    n = 1 << log_n
    _cases.clear()
    wires, cols, trivial = bench.synth_circuit(log_n)
    polys = {k: C.fr_bytes(v) for k, v in trivial.items()}
    for name, raw in cols.items():
        polys[name] = cbind.ntt_bytes(raw, log_n, True, False, n)
    srs = C.synthetic_srs(n + 7)
    case = dict(constraints=n, size=n, label=b"bench", polys=polys, wires=wires, pi={}, pi_idx=[], pi_val=b"")
    bl = C.blinders(1800 + log_n)
    got, vk = gpu_proof(ctx, case, srs, bl, host_too=True)
    cp = cbind.CProver(n, b"bench", polys, srs, vk48=vk)
    expected = cp.prove(wires, [], b"", bl)
    cp.close()
    assert got == expected


@pytest.mark.parametrize("profile,log_n", [("bench-like", 13), ("widgets", 13), ("dense", 12),
                                            pytest.param("bench-like", 17, marks=pytest.mark.slow)])   # 2^17: chunked coarse bins inside prove()
def test_wire_commitment_modes_agree_with_each_other_and_the_oracle(ctx, monkeypatch, profile, log_n):
    """The wire commitments taken from the wire VALUES over the Lagrange-basis key (default on one GPU) and from the
    blinded coefficient forms (PLONK_WIRE_COMMIT=coeff, the reference's way) are the same group elements: both modes
    must reproduce the C oracle's proof bytes on bench.py's workloads, including the skewed `bench-like` witness."""
    import bench_circuits as BC
    import plonk_amd
    configure(ctx, quotient_domain=4, wire_commit=0)
    n = 1 << log_n
    if profile == "widgets":
        wires, cols, pi = BC.widget_circuit(log_n)
        trivial = {}
    else:
        wires, cols, trivial = BC.arithmetic_circuit(log_n, profile)
        pi = {}
    polys = {k: C.fr_bytes(v) for k, v in trivial.items()}
    for name, raw in cols.items():
        polys[name] = cbind.ntt_bytes(raw, log_n, True, False, n)
    srs = C.synthetic_srs(n + 7)
    idx = sorted(pi)
    case = dict(constraints=n, size=n, label=b"modes", polys=polys, wires=wires, pi=pi, pi_idx=idx,
                pi_val=C.fr_bytes([pi[i] for i in idx]))
    cp = cbind.CProver(n, b"modes", polys, srs)
    bl = C.blinders(99)
    expected = cp.prove(wires, idx, case["pi_val"], bl)
    want_vk = cp.vk()
    cp.close()
    lag, vk = gpu_proof(ctx, case, srs, bl, expect={"wire_commit_values": 1})
    configure(ctx, wire_commit=1)
    coeff, vk2 = gpu_proof(ctx, case, srs, bl, expect={"wire_commit_values": 0})
    configure(ctx, wire_commit=0)
    assert vk == vk2 == want_vk
    assert lag == expected and coeff == expected
