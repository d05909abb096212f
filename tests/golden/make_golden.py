"""Generates tests/golden/vectors.json — small committed known-answer vectors for the hot path.

The reference (Rust) cannot run in this environment, so the vectors come from the oracle
restatement AFTER it has been pinned to the reference's own literal
(`deterministic_v3_proof_matches_base_digest`, prover.rs:1151-1158: this script refuses to write
anything unless the oracle reproduces that digest).  Inputs are the ones the reference's tests
use wherever it has any (domain.rs:570-651, quotient_poly.rs:315-349, kzg10/proof.rs:120-158).
The product-side tests (tests/test_golden.py) compare the HIP path with these bytes without
calling the oracle; a CPU test checks that the oracle still reproduces them.

    python tests/golden/make_golden.py        # rewrites vectors.json
"""
import hashlib
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import bls12_381 as E                      # noqa: E402
from oracle.bls12_381 import Q                         # noqa: E402
from oracle.fft import EvaluationDomain                # noqa: E402
from oracle.plonk import Composer, compile_circuit, prove, srs_setup  # noqa: E402
from oracle.rng import StdRng                          # noqa: E402
from oracle.serialize import prover_to_bytes           # noqa: E402

KAT_DIGEST = ("e8564ec22d8cc0ba", "1d333237")            # first / last bytes of the 64-byte literal; full value in tests/test_oracle_kat.py


def fr_digest(vals):
    return hashlib.blake2b(b"".join(v.to_bytes(32, "little") for v in vals)).hexdigest()


def hexs(vals):
    return ["%064x" % v for v in vals]


def main():
    out = {"_generator": "tests/golden/make_golden.py (oracle pinned to prover.rs:1151-1158)"}

    # ---- whole chain: the reference's KAT
    pp = srs_setup(1 << 10, StdRng.seed_from_u64(0x9235E700), keep=23)

    def circuit():
        c = Composer()
        w = c.append_witness(7)
        c.assert_equal_constant(w, 7)
        return c

    prover = compile_circuit(pp, b"proof-compatibility", circuit())
    proof, _ = prove(prover, StdRng.seed_from_u64(0x9235E701), circuit())
    digest = hashlib.blake2b(proof).hexdigest()
    assert digest.startswith(KAT_DIGEST[0]) and digest.endswith(KAT_DIGEST[1]), "oracle does not reproduce the reference KAT"
    rng = StdRng.seed_from_u64(0x9235E701)
    comp = circuit()
    W = comp.witnesses
    wires = [[W[getattr(g, k)] for g in comp.constraints] for k in "abcd"]
    out["kat"] = {
        "source": "prover.rs:1132-1162",
        "proof_hex": proof.hex(),
        "proof_blake2b": digest,
        "blinders": hexs([rng.random_scalar() for _ in range(14)]),
        "wires": [hexs(col) for col in wires],
        "size": prover.size, "constraints": prover.constraints,
        "srs_raw96_hex": [E.g1_to_raw96(p).hex() for p in prover.ck],
        "polys": {k: hexs(v) for k, v in prover.pk.polys.items()},
        "vk_compressed_hex": {k: E.g1_compress(v).hex() for k, v in prover.vk.items() if k != "n"},
        "prover_blob_blake2b": hashlib.blake2b(prover_to_bytes(prover)).hexdigest(),
        "prover_blob_len": len(prover_to_bytes(prover)),
    }

    # ---- NTT: the reference's own test inputs
    n = 4096
    a = [i + 1 for i in range(n)]                      # domain.rs:575
    d = EvaluationDomain(n)
    out["ntt_4096_i_plus_1"] = {
        "source": "domain.rs:570-618 (input), transforms domain.rs:166-232",
        "fft": {"blake2b": fr_digest(d.fft(a)), "head": hexs(d.fft(a)[:4])},
        "ifft": {"blake2b": fr_digest(d.ifft(a)), "head": hexs(d.ifft(a)[:4])},
        "coset_fft_515": {"blake2b": fr_digest(d.coset_fft(a[:515])), "head": hexs(d.coset_fft(a[:515])[:4])},
        "coset_ifft": {"blake2b": fr_digest(d.coset_ifft(a)), "head": hexs(d.coset_ifft(a)[:4])},
    }
    d8 = EvaluationDomain(256)
    lin = d8.coset_fft([0, 1])                         # domain.rs:620-636: 7 * w^i
    out["coset_linear_256"] = {"source": "domain.rs:620-636", "values": hexs(lin[:8]), "blake2b": fr_digest(lin)}
    batch = []
    for idx in range(5):                               # quotient_poly.rs:315-349: idx * n + i + 1 on 2^12
        v = [idx * n + i + 1 for i in range(n)]
        batch.append(fr_digest(d.coset_fft(v)))
    out["coset_batch_of_five_4096"] = {"source": "quotient_poly.rs:315-349", "blake2b": batch}

    # ---- MSM / commit
    r = random.Random(0x6D736D)
    pts = [E.g1_mul(E.G1_GEN, r.randrange(1, Q)) for _ in range(96)]
    cases = {}
    for name, sc in {
        "uniform_96": [r.randrange(Q) for _ in range(96)],
        "edge_scalars": [0, 1, 2, Q - 1, Q - 2, (1 << 255) % Q, (1 << 16) - 1, 1 << 16, (1 << 15), (1 << 15) + 1] + [r.randrange(Q) for _ in range(20)],
        "single_term": [r.randrange(Q)],
        "all_zero": [0] * 17,
    }.items():
        cases[name] = {"scalars": hexs(sc), "result_compressed_hex": E.g1_compress(E.msm_naive(pts[:len(sc)], sc)).hex()}
    out["msm_96"] = {"source": "key.rs:376-388 (commit = msm_variable_base over the first m bases)",
                     "points_raw96_hex": [E.g1_to_raw96(p).hex() for p in pts], "cases": cases}
    two, three, five = (E.g1_mul(E.G1_GEN, k) for k in (2, 3, 5))   # kzg10/proof.rs:120-158
    out["small_linear_combination"] = {
        "source": "kzg10/proof.rs:120-158 (2G, 3G, 5G; v = 7)",
        "result_compressed_hex": E.g1_compress(E.msm_naive([two, three, five], [1, 7, 49])).hex(),
        "equals_generator_times": 2 + 21 + 245,
    }
    with open(os.path.join(HERE, "vectors.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", os.path.join(HERE, "vectors.json"), os.path.getsize(os.path.join(HERE, "vectors.json")), "bytes")


if __name__ == "__main__":
    main()
