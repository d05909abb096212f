#!/usr/bin/env python
"""Seed corpus of tests/fuzz/fuzz_serial.cpp (committed under tests/fuzz/corpus/; this script regenerates it).

Seeds = the files the CPU tests already use — the reference KAT circuit's `Prover::to_bytes()` blob and the PublicParameters
file of its commit key in all three encodings, produced by oracle/serialize.py's restatement of the reference writers
(prover.rs:238-263, widget.rs:347-447, key.rs:215-229,303-308, srs.rs:114-153) — plus the malformed cases the reference's own
decoder tests walk through (prover.rs:796-1107 blob sections, key.rs:1020-1135 raw commit keys): patched lengths, non-canonical
scalars, off-curve points, wrong domains.  Each seed starts with the harness's 3-byte prefix (parser, truncated degree).

    python tests/fuzz/make_corpus.py            # rewrites tests/fuzz/corpus/
"""
import hashlib
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import bls12_381 as E  # noqa: E402
from oracle.bls12_381 import Q  # noqa: E402
from oracle.plonk import Composer, compile_circuit, srs_setup  # noqa: E402
from oracle.rng import StdRng  # noqa: E402
from oracle.serialize import (DOMAIN_SIZE, prover_to_bytes, public_parameters_to_raw_var_bytes,  # noqa: E402
                              public_parameters_to_var_bytes)


def kat_prover():
    pp = srs_setup(1 << 10, StdRng.seed_from_u64(0x9235E700), keep=23)
    c = Composer()
    w = c.append_witness(7)
    c.assert_equal_constant(w, 7)
    return compile_circuit(pp, b"proof-compatibility", c)


def patch(b, off, data):
    b = bytearray(b)
    b[off:off + len(data)] = data
    return bytes(b)


def main():
    import g2_ref
    out = os.path.join(HERE, "corpus")
    os.makedirs(out, exist_ok=True)
    for f in os.listdir(out):
        os.unlink(os.path.join(out, f))
    op = kat_prover()
    blob = prover_to_bytes(op)
    label_len, pk_len, ck_len, vk_len = (int.from_bytes(blob[8 * i:8 * i + 8], "big") for i in range(4))
    pk = 48 + label_len
    ck = pk + pk_len
    vk = ck + ck_len
    eval_size = 8 * op.size * 32 + DOMAIN_SIZE
    seeds = {"blob_kat": (0, 0, blob)}
    muts = {
        "size16": patch(blob, 32, (16).to_bytes(8, "big")),
        "constraints7": patch(blob, 40, (7).to_bytes(8, "big")),
        "pk_len_huge": patch(blob, 8, (1 << 40).to_bytes(8, "big")),
        "lens_overflow": patch(blob, 0, b"\xff" * 32),
        "pk_n16": patch(blob, pk, (16).to_bytes(8, "little")),
        "pk_n_huge": patch(blob, pk, (1 << 61).to_bytes(8, "little")),
        "eval_size_small": patch(blob, pk + 8, (100).to_bytes(8, "little")),
        "poly_len9": patch(blob, pk + 16, (9).to_bytes(8, "little")),
        "poly_len_huge": patch(blob, pk + 16, ((1 << 64) - 1).to_bytes(8, "little")),
        "coeff_q": patch(blob, pk + 24, Q.to_bytes(32, "little")),
        "linear_eval": patch(blob, ck - 2 * eval_size + DOMAIN_SIZE + 32 * 5, (1).to_bytes(32, "little")),
        "vanishing_eval": patch(blob, ck - eval_size + DOMAIN_SIZE, (0).to_bytes(32, "little")),
        "ck_count0": patch(blob, ck, (0).to_bytes(8, "little")),
        "ck_count22": patch(blob, ck, (22).to_bytes(8, "little")),
        "ck_count_huge": patch(blob, ck, ((1 << 64) // 97 + 5).to_bytes(8, "little")),
        "ck_identity": patch(blob, ck + 8 + 96, b"\x01"),
        "ck_off_curve": patch(blob, ck + 8 + 48, bytes([blob[ck + 8 + 48] ^ 1])),
        "vk_n": patch(blob, vk, (6).to_bytes(8, "little")),
        "vk_commitment": patch(blob, vk + 8, b"\x9f" + blob[vk + 9:vk + 56]),
        "vk_identity": patch(blob, vk + 8, b"\xc0" + bytes(47)),
    }
    # only the header and the first sections: small seeds the mutators can grow
    for name, m in muts.items():
        seeds["blob_" + name] = (0, 0, m)
    seeds["blob_header_only"] = (0, 0, blob[:48])
    seeds["blob_cut_pk"] = (0, 0, blob[:pk + 200])
    okey = E.g1_compress(E.G1_GEN) + g2_ref.g2_compress(g2_ref.G2_GEN) + g2_ref.g2_compress(g2_ref.g2_mul(g2_ref.G2_GEN, 0x1234567))
    raw = public_parameters_to_raw_var_bytes(okey, op.ck)
    comp = public_parameters_to_var_bytes(okey, op.ck)
    seeds["pp_raw_unchecked"] = (1, 0, raw)
    seeds["pp_raw"] = (2, 0, raw)
    seeds["pp_raw_trim8"] = (2, 8, raw)
    seeds["pp_raw_trim_too_large"] = (2, 30, raw)
    seeds["pp_compressed"] = (3, 0, comp)
    seeds["pp_compressed_trim4"] = (3, 4, comp)
    seeds["pp_raw_count_huge"] = (2, 0, patch(raw, 240, ((1 << 64) - 1).to_bytes(8, "little")))
    seeds["pp_raw_unchecked_count_huge"] = (1, 3, patch(raw, 240, ((1 << 64) - 1).to_bytes(8, "little")))
    seeds["pp_raw_identity"] = (2, 0, patch(raw, 240 + 8 + 96, b"\x01"))
    seeds["pp_raw_not_reduced"] = (2, 0, patch(raw, 240 + 8, b"\xff" * 48))
    seeds["pp_opening_identity"] = (2, 0, patch(raw, 0, b"\xc0" + bytes(47)))
    seeds["pp_opening_h_bad"] = (2, 0, patch(raw, 48, b"\x80" + bytes(95)))
    seeds["pp_compressed_short_chunk"] = (3, 0, comp[:-7])
    seeds["pp_compressed_identity"] = (3, 0, patch(comp, 240, b"\xc0" + bytes(47)))
    seeds["pp_compressed_no_flag"] = (3, 0, patch(comp, 240, bytes([comp[240] & 0x7F])))
    seeds["pp_only_opening_key"] = (2, 0, raw[:240])
    for name, (which, degree, data) in sorted(seeds.items()):
        with open(os.path.join(out, name), "wb") as f:
            f.write(bytes([which, degree & 0xFF, degree >> 8]) + data)
    total = sum(os.path.getsize(os.path.join(out, f)) for f in os.listdir(out))
    print(f"{len(seeds)} seeds, {total} bytes, blob sha256 {hashlib.sha256(blob).hexdigest()[:16]}")


if __name__ == "__main__":
    main()
