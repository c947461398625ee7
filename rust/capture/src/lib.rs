//! Fixture writer: one file per captured proof, `tests/golden/arkworks/<name>.bin`.
//!
//! What is captured is exactly what crosses the C ABI of `libmantagpu.so` on the prove path (`include/mantagpu.h`):
//! the three R1CS matrices as `cs.to_matrices()` gives them, the full assignment `z = instance || witness`, the two
//! blinding scalars in the order `create_random_proof` draws them, the proving key as `ProvingKey::serialize_unchecked`
//! writes it (`ProvingContext::encode`, `manta-crypto/src/arkworks/groth16.rs:290-303`) and the proof as
//! `proof_as_bytes` gives it (`groth16.rs:186-195`) -- produced by arkworks on the CPU, so a test that feeds the first
//! six to the GPU library and compares with the seventh pins the HIP path to the reference bit for bit.
//!
//! Container (little-endian; field elements are their in-memory Montgomery limbs, `Fp256.0.0`, 4 x u64):
//!
//! ```text
//! SECTIONS = magic curve m num_instance num_variables a b c z r s proving_key proof
//!   magic          8 bytes  "MGFX0001"
//!   curve          u32      0 = BN254, 1 = BLS12-381;  u32 reserved (0)
//!   m              u64      constraints
//!   num_instance   u64      P (incl. the constant one)
//!   num_variables  u64      V = instance + witness
//!   a, b, c        each: nnz u64 | row_ptr (m + 1) x u32 | col nnz x u32 | val nnz x 32 B
//!   z              V x 32 B
//!   r, s           32 B each
//!   proving_key    len u64 | bytes   (serialize_unchecked)
//!   proof          len u64 | bytes   (canonical compressed a | b | c)
//! ```
#![forbid(unsafe_code)]

use ark_ec::PairingEngine;
use ark_ff::{BigInteger256, FpParameters, PrimeField};
use ark_groth16::{Proof, ProvingKey};
use ark_relations::r1cs::ConstraintMatrices;
use ark_serialize::CanonicalSerialize;
use std::{fs, io::Write, path::Path};

/// Field order of the container (the Python mirror checks its own list against this line).
pub const SECTIONS: &str = "magic curve m num_instance num_variables a b c z r s proving_key proof";

/// Curves the library knows (`mg_curve_t`).
pub trait CaptureCurve: PairingEngine {
    const CURVE: u32;
    /// In-memory Montgomery limbs of a scalar-field element.
    fn limbs(x: &Self::Fr) -> [u64; 4];
}

impl CaptureCurve for ark_bn254::Bn254 {
    const CURVE: u32 = 0;
    #[inline]
    fn limbs(x: &Self::Fr) -> [u64; 4] {
        (x.0).0
    }
}

impl CaptureCurve for ark_bls12_381::Bls12_381 {
    const CURVE: u32 = 1;
    #[inline]
    fn limbs(x: &Self::Fr) -> [u64; 4] {
        (x.0).0
    }
}

fn put_fr<E: CaptureCurve>(out: &mut Vec<u8>, x: &E::Fr) {
    debug_assert_eq!(<<E::Fr as PrimeField>::Params as FpParameters>::MODULUS_BITS <= 256, true);
    let _: Option<BigInteger256> = None; // (the limb layout assumed above)
    for limb in E::limbs(x) {
        out.extend_from_slice(&limb.to_le_bytes());
    }
}

fn put_matrix<E: CaptureCurve>(out: &mut Vec<u8>, rows: &[Vec<(E::Fr, usize)>]) {
    let nnz: usize = rows.iter().map(Vec::len).sum();
    out.extend_from_slice(&(nnz as u64).to_le_bytes());
    let mut at = 0u32;
    out.extend_from_slice(&at.to_le_bytes());
    for row in rows {
        at += row.len() as u32;
        out.extend_from_slice(&at.to_le_bytes());
    }
    for row in rows {
        for (_, col) in row {
            out.extend_from_slice(&(*col as u32).to_le_bytes());
        }
    }
    for row in rows {
        for (coeff, _) in row {
            put_fr::<E>(out, coeff);
        }
    }
}

/// Serialises one captured proof; `z` = instance assignment followed by witness assignment.
pub fn encode<E: CaptureCurve>(
    matrices: &ConstraintMatrices<E::Fr>,
    z: &[E::Fr],
    r: &E::Fr,
    s: &E::Fr,
    proving_key: &ProvingKey<E>,
    proof: &Proof<E>,
) -> Vec<u8> {
    let mut out = Vec::new();
    out.extend_from_slice(b"MGFX0001");
    out.extend_from_slice(&E::CURVE.to_le_bytes());
    out.extend_from_slice(&0u32.to_le_bytes());
    out.extend_from_slice(&(matrices.num_constraints as u64).to_le_bytes());
    out.extend_from_slice(&(matrices.num_instance_variables as u64).to_le_bytes());
    out.extend_from_slice(&(z.len() as u64).to_le_bytes());
    put_matrix::<E>(&mut out, &matrices.a);
    put_matrix::<E>(&mut out, &matrices.b);
    put_matrix::<E>(&mut out, &matrices.c);
    for x in z {
        put_fr::<E>(&mut out, x);
    }
    put_fr::<E>(&mut out, r);
    put_fr::<E>(&mut out, s);
    let mut pk = Vec::new();
    proving_key
        .serialize_unchecked(&mut pk)
        .expect("serialising into a Vec cannot fail");
    out.extend_from_slice(&(pk.len() as u64).to_le_bytes());
    out.extend_from_slice(&pk);
    let mut pr = Vec::new();
    proof.serialize(&mut pr).expect("serialising into a Vec cannot fail");
    out.extend_from_slice(&(pr.len() as u64).to_le_bytes());
    out.extend_from_slice(&pr);
    out
}

/// The four classes of SURVEY.md 8(d)'s witness-like distribution, counted on the assignment a real transfer proves: how many
/// entries of `z` are 0, are 1, are another value below 2^64, are anything else (canonical values, `into_repr`). Zero scalars
/// cost a prover nothing, ones one addition, the last class a full set of windows -- arkworks and the GPU library alike -- so
/// these four numbers decide what a proofs/s figure on a SYNTHETIC witness is worth (`manta_rs_amd/synth.py` profiles
/// `sparse` / `W` / `dense`; `bench.py` prints the same histogram for what it proves).
pub fn histogram<E: CaptureCurve>(z: &[E::Fr]) -> [u64; 4] {
    let mut h = [0u64; 4];
    for x in z {
        let limbs: [u64; 4] = x.into_repr().0;
        let class = if limbs == [0, 0, 0, 0] {
            0
        } else if limbs == [1, 0, 0, 0] {
            1
        } else if limbs[1] == 0 && limbs[2] == 0 && limbs[3] == 0 {
            2
        } else {
            3
        };
        h[class] += 1;
    }
    h
}

/// How many entries of each query are the point at infinity (a variable absent from A or B): those scalars are skipped by the
/// MSMs whatever their value, so the effective density of the a / b MSMs is the histogram above restricted to the others.
pub fn infinity_counts<E: CaptureCurve>(proving_key: &ProvingKey<E>) -> [u64; 4] {
    use ark_ec::AffineCurve;
    [
        proving_key.a_query.iter().filter(|p| p.is_zero()).count() as u64,
        proving_key.b_g1_query.iter().filter(|p| p.is_zero()).count() as u64,
        proving_key.b_g2_query.iter().filter(|p| p.is_zero()).count() as u64,
        proving_key.l_query.iter().filter(|p| p.is_zero()).count() as u64,
    ]
}

/// `name.hist.json` next to the fixture: `{"n": V, "zero": .., "one": .., "small": .., "dense": .., "infinity": {...}}`.
pub fn histogram_json<E: CaptureCurve>(z: &[E::Fr], proving_key: &ProvingKey<E>) -> String {
    let h = histogram::<E>(z);
    let inf = infinity_counts::<E>(proving_key);
    format!(
        "{{\"n\": {}, \"zero\": {}, \"one\": {}, \"small\": {}, \"dense\": {}, \"infinity\": {{\"a_query\": {}, \"b_g1_query\": {}, \"b_g2_query\": {}, \"l_query\": {}}}}}\n",
        z.len(), h[0], h[1], h[2], h[3], inf[0], inf[1], inf[2], inf[3]
    )
}

/// Writes `dir/name.bin` and `dir/name.hist.json`.
pub fn write<E: CaptureCurve>(
    dir: &Path,
    name: &str,
    matrices: &ConstraintMatrices<E::Fr>,
    z: &[E::Fr],
    r: &E::Fr,
    s: &E::Fr,
    proving_key: &ProvingKey<E>,
    proof: &Proof<E>,
) -> std::io::Result<()> {
    fs::create_dir_all(dir)?;
    let mut f = fs::File::create(dir.join(format!("{name}.bin")))?;
    f.write_all(&encode::<E>(matrices, z, r, s, proving_key, proof))?;
    fs::write(dir.join(format!("{name}.hist.json")), histogram_json::<E>(z, proving_key))
}
