"""CPU: the synthetic workloads of bench.py (bench_circuits.py) are honest circuits — the C restatement of the
reference's prove() (oracle/c/oracle_prove.c) accepts their witnesses (a wrong gadget layout or permutation
would be Error::CircuitUnsatisfied, quotient_poly.rs:132), every selector family of the `widgets` profile is
active, its columns are not periodic, and the chunked generator used above 2^20 gates equals the
single-process one."""
import hashlib

import pytest

import bench_circuits as BC
from oracle import cbind
from tests import circuits as C


def prove_with_oracle(log_n, wires, cols, trivial, pi):
    n = 1 << log_n
    polys = {k: C.fr_bytes(v) for k, v in trivial.items()}
    for name, raw in cols.items():
        polys[name] = cbind.ntt_bytes(raw, log_n, True, False, n)
    cp = cbind.CProver(n, b"bench", polys, C.synthetic_srs(n + 7))
    idx = sorted(pi)
    proof = cp.prove(wires, idx, C.fr_bytes([pi[i] for i in idx]), C.blinders(5))
    cp.close()
    return proof, polys


@pytest.mark.parametrize("profile", ["dense", "bench-like"])
def test_arithmetic_profiles_are_satisfied(profile):
    wires, cols, trivial = BC.arithmetic_circuit(10, profile)
    proof, _ = prove_with_oracle(10, wires, cols, trivial, {})
    assert len(proof) == 1008
    if profile == "bench-like":     # about half of the wire values are < 4
        vals = C.fr_vals(wires[1]) + C.fr_vals(wires[3])
        small = sum(v < 4 for v in vals) / len(vals)
        assert 0.4 < small < 0.6


@pytest.mark.parametrize("log_n", [9, 12])
def test_widget_profile_is_satisfied_dense_and_complete(log_n):
    wires, cols, pi = BC.widget_circuit(log_n)
    assert len(pi) == 2
    proof, polys = prove_with_oracle(log_n, wires, cols, {}, pi)
    assert len(proof) == 1008
    for name in ("q_range", "q_logic", "q_fixed_group_add", "q_variable_group_add", "q_arith", "q_m"):
        assert any(polys[name]), name
    if log_n == 12:   # 16 tiles from a pool of different blocks: no periodic column, dense wire polynomials
        n = 1 << log_n
        a_poly = C.fr_vals(cbind.ntt_bytes(wires[0], log_n, True, False, n))
        assert sum(1 for v in a_poly if v) > n - 8


def test_chunk_workers_equal_the_in_process_generator():
    args = ("dense", 0x5EED0001 + 0x10000, 64)
    direct = BC._arith_chunk(args)
    import os
    import subprocess
    import sys
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "c.bin")
        subprocess.check_call([sys.executable, BC.__file__, args[0], str(args[1]), str(args[2]), path, "8", "64"])   # rows 64 .. 127 of a 2^8 domain
        raw = open(path, "rb").read()
    assert raw == b"".join(direct + BC._sigma_chunk(8, 64, 64))      # nine gate columns + the chunk's rows of sigma_1 / sigma_3 (round 5)
    # a chain of its own: sigma_1 maps the chain's first row to itself, sigma_3 its last row
    T = BC._omega_table(8)
    s1, s3 = BC._sigma_chunk(8, 64, 64)
    assert s1[:32] == BC._bytes([T[64]]) and s1[32:64] == BC._bytes([BC.K2 * T[64] % BC.Q])
    assert s3[-32:] == BC._bytes([BC.K2 * T[127] % BC.Q]) and s3[:32] == BC._bytes([T[65]])
    # the headline circuit is one chain from one stream: its digest is part of the bench line's stability
    w, c, t = BC.arithmetic_circuit(8, "dense")
    assert hashlib.blake2b(b"".join(w)).hexdigest()[:16] == hashlib.blake2b(b"".join(BC.arithmetic_circuit(8, "dense")[0])).hexdigest()[:16]


def _sigma_columns_from_witnesses(lib, cc, log_n):
    """Permutation::compute_sigma_polynomials' evaluation columns (permutation.rs:141-175) from the gate-column
    form, through the product's host pass (plonk_amd/csrc/permutation.hpp, compiled for the host)."""
    import ctypes
    n = 1 << log_n
    out = (ctypes.c_uint32 * (4 * n))()
    bufs = [ctypes.create_string_buffer(w, len(w)) for w in cc["wires"]]
    args = [ctypes.cast(b, ctypes.POINTER(ctypes.c_uint32)) for b in bufs]
    assert lib.h_sigma_mappings(*args, ctypes.c_uint64(n), ctypes.c_uint64(n), ctypes.c_uint64(cc["witnesses"]), out) == 0
    T = BC._omega_table(log_n)
    ks = (1, BC.K1, BC.K2, BC.K3)
    return [BC._bytes([ks[out[col * n + i] >> 30] * T[out[col * n + i] & 0x3FFFFFFF] % BC.Q for i in range(n)]) for col in range(4)]


def test_gate_column_forms_describe_the_same_circuits():
    """arithmetic_columns / widget_columns (what plonk_compile takes) against arithmetic_circuit / widget_circuit
    (what the coefficient-form path takes): same wire columns from the witness table, same sigma columns from the
    witness indices, same selectors."""
    from tests.test_field_host import build_host_lib
    lib = build_host_lib()
    log_n = 10
    for profile in ("dense", "bench-like"):
        wires, cols, trivial = BC.arithmetic_circuit(log_n, profile)
        cc = BC.arithmetic_columns(log_n, profile)
        sig = _sigma_columns_from_witnesses(lib, cc, log_n)
        T = BC._omega_table(log_n)
        assert sig[0] == cols["s_sigma_1"] and sig[2] == cols["s_sigma_3"]
        assert sig[1] == BC._bytes([BC.K1 * t % BC.Q for t in T]) and sig[3] == BC._bytes([BC.K3 * t % BC.Q for t in T])
        assert trivial["s_sigma_2"] == [0, BC.K1] and trivial["s_sigma_4"] == [0, BC.K3]
        for name in ("q_m", "q_l", "q_r", "q_f", "q_c"):
            assert cc["selectors"][name] == cols[name]
        assert C.fr_vals(cc["selectors"]["q_o"]) == [BC.Q - 1] * (1 << log_n) and trivial["q_o"] == [BC.Q - 1]
        assert C.fr_vals(cc["selectors"]["q_arith"]) == [1] * (1 << log_n) and trivial["q_arith"] == [1]
        assert cc["columns"] == wires
    wires, cols, pi = BC.widget_circuit(11)
    cc = BC.widget_columns(11)
    sig = _sigma_columns_from_witnesses(lib, cc, 11)
    for k in range(4):
        assert sig[k] == cols[f"s_sigma_{k + 1}"]
    assert cc["columns"] == wires and cc["public_inputs"] == pi
    assert {k: v for k, v in cols.items() if k.startswith("q_")} == cc["selectors"]
