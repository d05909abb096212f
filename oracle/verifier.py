"""Proof::verify restatement with a KNOWN-tau shortcut (oracle; test infrastructure only).

Follows reference src/proof_system/proof.rs:218-507 (V3 verification): transcript replay,
r_0, [D] (append_linearization_commitment_terms :808-889 and each widget's verifier-key
compute_linearization_commitment), [F], [E] and the final check

    e(-([W_z] + u [W_zw]), [tau]H) * e(z [W_z] + u z w [W_zw] + [F] - [E] + [D], H) == 1 .

The pairing itself is outside this build's scope (verification is O(1), CPU).  For
the synthetic SRS used by tests and bench.py the trapdoor tau is known, so the pairing equation
is equivalent to the G1 identity   right == tau * ([W_z] + u [W_zw])   which needs no G2/Gt.
This gives a SIZE-INDEPENDENT acceptance test for proofs produced at BASELINE sizes (2^16 ...
2^22 gates) where no CPU prover can serve as oracle.
"""
from __future__ import annotations

from . import bls12_381 as E
from .bls12_381 import K1, K2, K3, Q, fr_inv
from .fft import EvaluationDomain
from .plonk import (COMM_ORDER, EVAL_ORDER, fixed_identity, logic_identity, range_identity, seed_transcript_v3,
                    var_identity)

V_MAX_DEGREE = 11   # proof.rs:23


def parse_proof(proof: bytes):
    assert len(proof) == 1008
    comm = {k: E.g1_decompress(proof[48 * i:48 * i + 48]) for i, k in enumerate(COMM_ORDER)}
    ev = {}
    for i, k in enumerate(EVAL_ORDER):
        v = int.from_bytes(proof[528 + 32 * i:528 + 32 * i + 32], "little")
        assert v < Q, "non-canonical scalar"
        ev[k] = v
    return comm, ev


def verify_with_tau(proof: bytes, vk: dict, label: bytes, constraints: int, public_inputs: dict,
                    tau: int, srs_g) -> bool:
    """vk: name -> affine commitment (or None) for the 15 key polynomials; public_inputs:
    gate row -> value; srs_g = powers_of_g[0] (OpeningKey.g).  Returns True iff the proof
    satisfies the verification equation."""
    comm, ev = parse_proof(proof)
    domain = EvaluationDomain(constraints)
    n = domain.size
    tr = seed_transcript_v3(label, dict(vk, n=constraints), constraints)
    pis = sorted(public_inputs.items())
    for _, v in pis:
        tr.append_scalar(b"pi", v)
    for name in "abcd":
        tr.append_commitment(f"{name}_comm".encode(), comm[name])
    beta = tr.challenge_scalar(b"beta")
    tr.append_scalar(b"beta", beta)
    gamma = tr.challenge_scalar(b"gamma")
    tr.append_commitment(b"z_comm", comm["z"])
    alpha = tr.challenge_scalar(b"alpha")
    range_ch = tr.challenge_scalar(b"range separation challenge")
    logic_ch = tr.challenge_scalar(b"logic separation challenge")
    fixed_ch = tr.challenge_scalar(b"fixed base separation challenge")
    var_ch = tr.challenge_scalar(b"variable base separation challenge")
    for name in ("t_low", "t_mid", "t_high", "t_fourth"):
        tr.append_commitment(f"{name}_comm".encode(), comm[name])
    z = tr.challenge_scalar(b"z_challenge")
    for lab in ("a", "b", "c", "d", "s_sigma_1", "s_sigma_2", "s_sigma_3", "z"):
        tr.append_scalar(f"{lab}_eval".encode(), ev[lab])
    for lab in ("a_w", "b_w", "d_w", "q_arith", "q_c", "q_l", "q_r"):
        tr.append_scalar(f"{lab}_eval".encode(), ev[lab])
    v = tr.challenge_scalar(b"v_challenge")
    v_w = tr.challenge_scalar(b"v_w_challenge")
    tr.append_commitment(b"w_z_chall_comm", comm["w_z"])
    tr.append_commitment(b"w_z_chall_w_comm", comm["w_zw"])
    u = tr.challenge_scalar(b"u_challenge")

    z_n = pow(z, n, Q)
    z_h = (z_n - 1) % Q
    # compute_lagrange_and_barycentric_evaluations (proof.rs:997-1039)
    if (z - 1) % Q == 0:
        return False
    l1 = z_h * fr_inv(domain.size_as_field_element * (z - 1) % Q) % Q
    acc = 0
    for idx, val in pis:
        if val % Q:
            den = (pow(domain.group_gen_inv, idx, Q) * z - 1) % Q
            if den == 0:
                return False
            acc = (acc + fr_inv(den) * val) % Q
    pi_eval = acc * z_h % Q * domain.size_inv % Q
    r0 = (pi_eval - l1 * alpha * alpha
          - alpha * (ev["a"] + beta * ev["s_sigma_1"] + gamma) * (ev["b"] + beta * ev["s_sigma_2"] + gamma)
          * (ev["c"] + beta * ev["s_sigma_3"] + gamma) * (ev["d"] + gamma) * ev["z"]) % Q
    vc = [0] * (V_MAX_DEGREE + 3)
    vc[0] = v
    for i in range(1, V_MAX_DEGREE):
        vc[i] = vc[i - 1] * v % Q
    vc[V_MAX_DEGREE] = v_w * u % Q
    vc[V_MAX_DEGREE + 1] = vc[V_MAX_DEGREE] * v_w % Q
    vc[V_MAX_DEGREE + 2] = vc[V_MAX_DEGREE + 1] * v_w % Q
    e_evals = [ev[k] for k in ("a", "b", "c", "d", "s_sigma_1", "s_sigma_2", "s_sigma_3", "q_arith", "q_c", "q_l",
                               "q_r", "a_w", "b_w", "d_w")]
    e_scalar = (sum(x * c for x, c in zip(e_evals, vc)) - r0 + u * ev["z"]) % Q

    terms = []   # (scalar, point)
    qa = ev["q_arith"]
    terms += [(ev["a"] * ev["b"] * qa, vk["q_m"]), (ev["a"] * qa, vk["q_l"]), (ev["b"] * qa, vk["q_r"]),
              (ev["c"] * qa, vk["q_o"]), (ev["d"] * qa, vk["q_f"]), (qa, vk["q_c"])]
    terms.append((range_identity(range_ch, ev["a"], ev["b"], ev["c"], ev["d"], ev["d_w"]) * range_ch, vk["q_range"]))
    terms.append((logic_identity(logic_ch, ev["a"], ev["a_w"], ev["b"], ev["b_w"], ev["c"], ev["d"], ev["d_w"],
                                 ev["q_c"]) * logic_ch, vk["q_logic"]))
    terms.append((fixed_identity(fixed_ch, ev["a"], ev["a_w"], ev["b"], ev["b_w"], ev["c"], ev["d"], ev["d_w"],
                                 ev["q_l"], ev["q_r"], ev["q_c"]) * fixed_ch, vk["q_fixed_group_add"]))
    terms.append((var_identity(var_ch, ev["a"], ev["a_w"], ev["b"], ev["b_w"], ev["c"], ev["d"], ev["d_w"]) * var_ch,
                  vk["q_variable_group_add"]))
    # permutation/verifierkey.rs:46-104
    x = ((ev["a"] + beta * z + gamma) * (ev["b"] + beta * K1 * z + gamma) % Q * (ev["c"] + beta * K2 * z + gamma) % Q
         * ((ev["d"] + beta * K3 * z + gamma) * alpha % Q)) % Q
    terms.append(((x + l1 * alpha * alpha + u) % Q, comm["z"]))
    y = -((ev["a"] + beta * ev["s_sigma_1"] + gamma) * (ev["b"] + beta * ev["s_sigma_2"] + gamma) % Q
          * (ev["c"] + beta * ev["s_sigma_3"] + gamma) % Q * (beta * ev["z"] % Q * alpha % Q)) % Q
    terms.append((y, vk["s_sigma_4"]))
    nzh = (-z_h) % Q
    terms += [(nzh, comm["t_low"]), (z_n * nzh, comm["t_mid"]), (z_n * z_n * nzh, comm["t_high"]),
              (z_n * z_n * z_n * nzh, comm["t_fourth"])]
    f = vc[:V_MAX_DEGREE]
    f[0] = (f[0] + vc[V_MAX_DEGREE]) % Q
    f[1] = (f[1] + vc[V_MAX_DEGREE + 1]) % Q
    f[3] = (f[3] + vc[V_MAX_DEGREE + 2]) % Q
    f_points = [comm["a"], comm["b"], comm["c"], comm["d"], vk["s_sigma_1"], vk["s_sigma_2"], vk["s_sigma_3"],
                vk["q_arith"], vk["q_c"], vk["q_l"], vk["q_r"]]
    terms += list(zip(f, f_points))
    terms.append(((-e_scalar) % Q, srs_g))
    terms.append((z, comm["w_z"]))
    terms.append((u * z % Q * domain.group_gen % Q, comm["w_zw"]))
    right = E.msm_naive([p for _, p in terms], [s % Q for s, _ in terms])
    left = E.g1_add(comm["w_z"], E.g1_mul(comm["w_zw"], u) if comm["w_zw"] is not None else None)
    return right == (E.g1_mul(left, tau) if left is not None else None)
