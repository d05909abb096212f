"""CPU oracle for the dusk-plonk prover hot path (NTT + KZG MSM inside Prover::prove).

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it.
The product (`plonk_amd/`) never imports, links or executes anything in here.

Pinning: `oracle.prover.prove` reproduces the reference's only end-to-end
known-answer test, `deterministic_v3_proof_matches_base_digest`
(reference src/compiler/prover.rs:1132-1162): blake2b-512 of the 1008 proof
bytes equals the literal at prover.rs:1151-1158.  See tests/test_oracle_kat.py.
That single digest pins Montgomery/limb conventions, the ChaCha12 `StdRng`
stream, `BlsScalar::random`/`from_bytes_wide`, Merlin/STROBE framing, transcript
label order, G1 compressed encoding, blinding layout and every NTT/MSM result on
the path.

The arithmetic itself lives in third-party crates that are NOT vendored under
/root/reference (dusk-bls12_381 "0.14", merlin "3.0", rand "0.8" -> rand_chacha
0.3 / rand_core 0.6, blake2b_simd =1.0.3; reference Cargo.toml:19-42).  Their
published algorithms are restated here from the BLS12-381 / STROBE / ChaCha
specifications and anchored by the digest above.
"""
