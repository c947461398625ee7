//! Run (on a machine with cargo and the ark-* 0.3 crates):
//!
//! ```sh
//! MANTA_CAPTURE_DIR=$PWD/../../tests/golden/arkworks cargo test --release -- --test-threads 1
//! ```
//!
//! Each test proves one transfer shape through the reference's own sampling helpers
//! (`manta-pay/src/test/payment.rs:52-83,222-273,364-413`, the functions `manta-benchmark/benches/*.rs` time) with a seeded
//! RNG; the capture hook in `Groth16::prove` writes `tests/golden/arkworks/<shape>-0000.bin`. Commit the files:
//! `tests/test_gpu_reference_fixture.py` then checks the GPU prover against them byte for byte.

use manta_pay::{
    parameters,
    test::payment::{private_transfer, to_private, to_public, UtxoAccumulator},
};
use rand_chacha::{rand_core::SeedableRng, ChaCha20Rng};

fn rng(tag: u64) -> ChaCha20Rng {
    ChaCha20Rng::seed_from_u64(0x4D41_4E54_4100_0000 | tag)
}

#[test]
fn capture_to_private() {
    std::env::set_var("MANTA_CAPTURE_NAME", "to-private");
    let (proving_context, _, parameters, utxo_accumulator_model) = parameters::generate().expect("parameters");
    to_private::prove(&proving_context.to_private, &parameters, &utxo_accumulator_model, &mut rng(1));
}

#[test]
fn capture_private_transfer() {
    let (proving_context, _, parameters, utxo_accumulator_model) = parameters::generate().expect("parameters");
    let mut accumulator = UtxoAccumulator::new(utxo_accumulator_model);
    // the two ToPrivate proofs inside `prove` are captured too (names to-private-for-pt-0000 / -0001)
    std::env::set_var("MANTA_CAPTURE_NAME", "private-transfer");
    private_transfer::prove(&proving_context, &parameters, &mut accumulator, &mut rng(2));
}

#[test]
fn capture_to_public() {
    let (proving_context, _, parameters, utxo_accumulator_model) = parameters::generate().expect("parameters");
    let mut accumulator = UtxoAccumulator::new(utxo_accumulator_model);
    std::env::set_var("MANTA_CAPTURE_NAME", "to-public");
    to_public::prove(&proving_context, &parameters, &mut accumulator, &mut rng(3));
}
