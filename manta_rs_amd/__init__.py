"""manta_rs_amd -- MI355X-native Groth16 prove hot path for manta-rs (host-side mirror over the C ABI).

The product is `lib/libmantagpu.so` (hand-written HIP for gfx950 + a C++ host runtime, C ABI in
`include/mantagpu.h`). This package is the thin Python mirror of the reference interface for this path:

    reference (Rust)                                            here
    ----------------------------------------------------------  ---------------------------------
    manta_crypto::constraint::ProofSystem::prove                 Groth16.prove(context, compiler, rng)
      (manta-crypto/src/constraint.rs:47-103, impl at
       manta-crypto/src/arkworks/groth16.rs:589-600)
    groth16::ProvingContext<E>{proving_key}  (groth16.rs:216)    ProvingContext
    constraint::R1CS<F> (arkworks/constraint/mod.rs:94-135)      R1CS (matrices + assignment)
    ark_ec::msm::VariableBaseMSM::multi_scalar_mul               VariableBaseMSM.multi_scalar_mul
    ark_poly::Radix2EvaluationDomain::{fft,ifft,coset_*}         Radix2EvaluationDomain

There is NO CPU fallback: importing `manta_rs_amd.api` fails loudly if the HIP library is missing.
"""
from . import synth  # noqa: F401

BN254 = 0
BLS12_381 = 1
