// Pins the on-disk format of `Prover::to_bytes()` with ONE external run.
//
// This container has no cargo, so `plonk_prover_from_bytes` (plonk_amd/csrc/serial.hip) is tested
// against oracle/serialize.py's restatement of the layout.  Whoever has a Rust toolchain can close
// the loop: drop this file into the reference tree as `tests/dump_kat_blob.rs`, run
//
//     cargo test --release --test dump_kat_blob -- --nocapture
//
// and compare the printed digest with the one this repository predicts for the very same prover
// (tests/test_prover_blob.py::test_kat_blob_digest_prediction):
//
//     len    = 43966 bytes
//     blake2b-512 = 959ac0e3ee3d8f14695fccf849c92c9e2292e979b6de720d9a6d15e279b88e98
//                   da02c27c8e99f76ec8453c0b7813f7d11e68baf5cf67795ec985771bcca3ed23
//
// The prover is the one of the reference's own KAT (src/compiler/prover.rs:1132-1147): SRS seed
// 0x9235_e700, `setup(1 << 10)`, MinimalCircuit, label b"proof-compatibility".  Equal digests mean the
// loader has been reading the reference's real bytes all along; a mismatch localises to
// oracle/serialize.py (field order / integer endianness / padding), not to the GPU path.
use dusk_plonk::prelude::*;
use rand::rngs::StdRng;
use rand::SeedableRng;

#[derive(Default)]
struct MinimalCircuit;

impl Circuit for MinimalCircuit {
    fn circuit(&self, composer: &mut Composer) -> Result<(), Error> {
        let w = composer.append_witness(BlsScalar::from(7u64));
        composer.assert_equal_constant(w, BlsScalar::from(7u64), None);
        Ok(())
    }
}

#[test]
fn dump_kat_prover_blob_digest() {
    let mut setup_rng = StdRng::seed_from_u64(0x9235_e700);
    let pp = PublicParameters::setup(1 << 10, &mut setup_rng).expect("public parameters");
    let (prover, _) = Compiler::compile::<MinimalCircuit>(&pp, b"proof-compatibility").expect("compile");
    let blob = prover.to_bytes();
    let digest = blake2b_simd::blake2b(&blob);
    println!("len = {}", blob.len());
    println!("blake2b-512 = {}", digest.to_hex());
    // optional: write the blob so that tests/test_prover_blob.py can load the real thing
    std::fs::write("kat_prover.blob", &blob).expect("write blob");
}
