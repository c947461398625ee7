//! MI355X backend of `manta_crypto::arkworks::groth16` — the safe layer over `mantagpu-sys`.
//!
//! What the reference does with arkworks and what this crate does instead:
//!
//! | reference (manta-crypto/src/arkworks/groth16.rs)                   | here                                   |
//! |--------------------------------------------------------------------|----------------------------------------|
//! | `ProvingContext::new(proving_key)` `:216-245`                      | [`GpuProvingContext::new`]             |
//! | `impl Decode for ProvingContext` `:268-288`                        | [`GpuProvingContext::from_bytes`]      |
//! | `ArkGroth16::prove(&pk, compiler, &mut SizedRng(rng))` `:589-600`  | [`GpuProvingContext::prove`]           |
//! | k calls of the above from a signer / ledger simulation             | [`GpuProvingContext::prove_many`]      |
//! | `VerifyingContext::new(&vk)` / `decode` `:305-517`                 | [`GpuVerifyingContext`]                |
//! | `ArkGroth16::verify_with_processed_vk` `:603-609`                  | [`GpuVerifyingContext::verify`]        |
//!
//! Source only: the build image of this repository has no Rust toolchain, so this file has never been compiled.

use ark_ec::{AffineCurve, PairingEngine};
use ark_ff::{PrimeField, UniformRand, Zero};
use ark_groth16::{Proof, ProvingKey, VerifyingKey};
use ark_relations::r1cs::{
    ConstraintMatrices, ConstraintSynthesizer, ConstraintSystem, OptimizationGoal, SynthesisError,
};
use ark_serialize::CanonicalDeserialize;
use core::{marker::PhantomData, ptr};
use mantagpu_sys as sys;
use rand_core::{CryptoRng, RngCore};
use std::sync::Mutex;

/// Mirror of the reference's intentionally opaque unit error (`groth16.rs:50-60`): every non-zero status of the C
/// ABI, every arkworks error, maps to it.
#[derive(Clone, Copy, Debug, Default, Eq, Hash, PartialEq)]
pub struct Error;

impl From<SynthesisError> for Error {
    #[inline]
    fn from(_: SynthesisError) -> Self {
        Self
    }
}

#[inline]
fn check(status: core::ffi::c_int) -> Result<(), Error> {
    if status == sys::MG_SUCCESS {
        Ok(())
    } else {
        Err(Error)
    }
}

/// A pairing engine the library has kernels for. The two impls below read the arkworks structs field by field
/// (`GroupAffine { x, y, infinity }`, `Fp256(BigInteger256([u64; 4]))`): Montgomery limbs leave Rust untouched.
pub trait Mi355xCurve: PairingEngine {
    /// `MG_BN254` or `MG_BLS12_381`
    const CURVE: sys::mg_curve_t;
    /// compressed proof: 128 B (BN254) / 192 B (BLS12-381)
    const PROOF_BYTES: usize;
    /// `u64` limbs of one G1 / G2 affine point in the C ABI
    const G1_LIMBS: usize;
    const G2_LIMBS: usize;
    /// appends `x || y` (infinity: zeros) of `p`
    fn push_g1(p: &Self::G1Affine, out: &mut Vec<u64>);
    /// appends `x.c0 x.c1 y.c0 y.c1`
    fn push_g2(p: &Self::G2Affine, out: &mut Vec<u64>);
    /// the four Montgomery limbs of a scalar
    fn fr_limbs(f: &Self::Fr) -> [u64; 4];
}

impl Mi355xCurve for ark_bn254::Bn254 {
    const CURVE: sys::mg_curve_t = sys::MG_BN254;
    const PROOF_BYTES: usize = 128;
    const G1_LIMBS: usize = 8;
    const G2_LIMBS: usize = 16;
    #[inline]
    fn push_g1(p: &ark_bn254::G1Affine, out: &mut Vec<u64>) {
        if p.infinity {
            out.extend_from_slice(&[0; 8]);
        } else {
            out.extend_from_slice(&p.x.0 .0);
            out.extend_from_slice(&p.y.0 .0);
        }
    }
    #[inline]
    fn push_g2(p: &ark_bn254::G2Affine, out: &mut Vec<u64>) {
        if p.infinity {
            out.extend_from_slice(&[0; 16]);
        } else {
            out.extend_from_slice(&p.x.c0.0 .0);
            out.extend_from_slice(&p.x.c1.0 .0);
            out.extend_from_slice(&p.y.c0.0 .0);
            out.extend_from_slice(&p.y.c1.0 .0);
        }
    }
    #[inline]
    fn fr_limbs(f: &ark_bn254::Fr) -> [u64; 4] {
        f.0 .0
    }
}

impl Mi355xCurve for ark_bls12_381::Bls12_381 {
    const CURVE: sys::mg_curve_t = sys::MG_BLS12_381;
    const PROOF_BYTES: usize = 192;
    const G1_LIMBS: usize = 12;
    const G2_LIMBS: usize = 24;
    #[inline]
    fn push_g1(p: &ark_bls12_381::G1Affine, out: &mut Vec<u64>) {
        if p.infinity {
            out.extend_from_slice(&[0; 12]);
        } else {
            out.extend_from_slice(&p.x.0 .0);
            out.extend_from_slice(&p.y.0 .0);
        }
    }
    #[inline]
    fn push_g2(p: &ark_bls12_381::G2Affine, out: &mut Vec<u64>) {
        if p.infinity {
            out.extend_from_slice(&[0; 24]);
        } else {
            out.extend_from_slice(&p.x.c0.0 .0);
            out.extend_from_slice(&p.x.c1.0 .0);
            out.extend_from_slice(&p.y.c0.0 .0);
            out.extend_from_slice(&p.y.c1.0 .0);
        }
    }
    #[inline]
    fn fr_limbs(f: &ark_bls12_381::Fr) -> [u64; 4] {
        f.0 .0
    }
}

/// `x || y` Montgomery limbs of a slice of G1 points, infinity = zeros (the C ABI's point format).
pub fn flatten_g1<E: Mi355xCurve>(points: &[E::G1Affine]) -> Vec<u64> {
    let mut out = Vec::with_capacity(points.len() * E::G1_LIMBS);
    for p in points {
        E::push_g1(p, &mut out);
    }
    out
}

/// The same for G2.
pub fn flatten_g2<E: Mi355xCurve>(points: &[E::G2Affine]) -> Vec<u64> {
    let mut out = Vec::with_capacity(points.len() * E::G2_LIMBS);
    for p in points {
        E::push_g2(p, &mut out);
    }
    out
}

/// One matrix of `cs.to_matrices()` (`Vec<Vec<(F, usize)>>`, one inner vector per constraint) in CSR form.
pub struct Csr {
    row_ptr: Vec<u32>,
    col: Vec<u32>,
    val: Vec<u64>,
}

impl Csr {
    pub fn new<E: Mi355xCurve>(rows: &[Vec<(E::Fr, usize)>]) -> Self {
        let nnz = rows.iter().map(Vec::len).sum::<usize>();
        let mut row_ptr = Vec::with_capacity(rows.len() + 1);
        let mut col = Vec::with_capacity(nnz);
        let mut val = Vec::with_capacity(4 * nnz);
        row_ptr.push(0);
        for row in rows {
            for (coeff, index) in row {
                col.push(*index as u32);
                val.extend_from_slice(&E::fr_limbs(coeff));
            }
            row_ptr.push(col.len() as u32);
        }
        Self { row_ptr, col, val }
    }

    fn view(&self) -> sys::mg_csr {
        sys::mg_csr {
            row_ptr: self.row_ptr.as_ptr(),
            col: self.col.as_ptr(),
            val: self.val.as_ptr(),
            nnz: self.col.len() as u64,
        }
    }
}

/// Device-resident proving key (+ the circuit's matrices once the first proof has been asked for). Lifetime = the
/// reference's `ProvingContext<E>`; shared by reference across the signer's threads exactly like it
/// (`manta-pay/src/simulation/mod.rs:75-79`): `mg_groth16_prove` is re-entrant on one context.
pub struct GpuProvingContext<E: Mi355xCurve> {
    ctx: *mut sys::mg_ctx,
    /// `(num_constraints, num_instance, nnz(A), nnz(B), nnz(C))` of the matrices on the device
    shape: Mutex<Option<(usize, usize, usize, usize, usize)>>,
    _engine: PhantomData<E>,
}

// SAFETY: the handle is an opaque pointer to a thread-safe C++ object (per-call streams and workspaces from internal
// pools; `mg_ctx_set_r1cs` takes an internal exclusive lock against proofs in flight).
unsafe impl<E: Mi355xCurve> Send for GpuProvingContext<E> {}
unsafe impl<E: Mi355xCurve> Sync for GpuProvingContext<E> {}

impl<E: Mi355xCurve> Drop for GpuProvingContext<E> {
    #[inline]
    fn drop(&mut self) {
        // SAFETY: created by mg_ctx_create*, destroyed exactly once
        unsafe { sys::mg_ctx_destroy(self.ctx) }
    }
}

/// How the partial points of a context sharded over several GPUs meet (`mg_ctx_opts.exchange`).
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum Exchange {
    /// through pinned host memory, summed on the host
    Host,
    /// folded on every GPU, one grouped `ncclAllGather` over xGMI inside the library (BASELINE north_star)
    Rccl,
}

/// Per-context decisions (`mg_ctx_opts`). A signer holds three contexts at once (`MultiProvingContext`,
/// `manta-accounting/src/transfer/canonical.rs:561-588`): what each may spend on its full tables is stated here, not in
/// the environment.
#[derive(Clone, Debug, Default)]
pub struct ContextOptions<'d> {
    /// `None` = the current HIP device; `Some(list)` range-shards every query over the listed GPUs
    pub devices: Option<&'d [i32]>,
    /// exchange of a sharded context; `None` = `Exchange::Host`
    pub exchange: Option<Exchange>,
    /// HBM the context may spend on full tables (single proofs run on them); `None` = a tenth of the device's HBM,
    /// `Some(0)` = bucket tables only
    pub full_table_bytes: Option<u64>,
    /// this context's scheduling (`mg_ctx_opts.tuning`); `None` = the process-wide values ([`tuning`] / [`set_tuning`])
    pub tuning: Option<sys::mg_tuning>,
}

/// The process-wide tuning in force (`mg_get_tuning`): the compiled-in defaults, the `MANTA_*` variables of
/// `mg_tuning_env_names` applied once, then whatever [`set_tuning`] stated. No field changes a proof's bytes.
pub fn tuning() -> Result<sys::mg_tuning, Error> {
    let mut t = core::mem::MaybeUninit::<sys::mg_tuning>::zeroed();
    // SAFETY: mg_get_tuning writes every field
    check(unsafe { sys::mg_get_tuning(t.as_mut_ptr()) })?;
    Ok(unsafe { t.assume_init() })
}

/// States the process-wide tuning for contexts created from now on (`mg_set_tuning`): validated as a whole, an
/// out-of-range field is an `Err` and leaves everything as it was.
pub fn set_tuning(t: &sys::mg_tuning) -> Result<(), Error> {
    check(unsafe { sys::mg_set_tuning(t) })
}

impl<E: Mi355xCurve> GpuProvingContext<E> {
    /// Uploads `proving_key` (the library copies and re-lays everything; no pointer is retained). `devices`: `None`
    /// = the current HIP device; `Some(list)` range-shards every query over the listed GPUs (`mg_ctx_create_sharded`).
    pub fn new(proving_key: &ProvingKey<E>, devices: Option<&[i32]>) -> Result<Self, Error> {
        Self::with_options(proving_key, &ContextOptions { devices, ..Default::default() })
    }

    fn abi_options(options: &ContextOptions<'_>) -> Result<sys::mg_ctx_opts, Error> {
        let mut o = core::mem::MaybeUninit::<sys::mg_ctx_opts>::zeroed();
        // SAFETY: mg_ctx_opts_init writes every field
        check(unsafe { sys::mg_ctx_opts_init(o.as_mut_ptr()) })?;
        let mut o = unsafe { o.assume_init() };
        if let Some(d) = options.devices {
            o.devices = d.as_ptr();
            o.n_devices = d.len() as i32;
        }
        if let Some(x) = options.exchange {
            o.exchange = if x == Exchange::Rccl { sys::MG_EXCHANGE_RCCL } else { sys::MG_EXCHANGE_HOST };
        }
        if let Some(b) = options.full_table_bytes {
            o.full_table_bytes = b.min(i64::MAX as u64) as i64;
        }
        if let Some(t) = options.tuning.as_ref() {
            o.tuning = t; // (read during mg_ctx_create_ex: `options` outlives the call)
        }
        Ok(o)
    }

    /// [`new`](Self::new) with the placement, the exchange and the table budget stated (`mg_ctx_create_ex`).
    pub fn with_options(proving_key: &ProvingKey<E>, options: &ContextOptions<'_>) -> Result<Self, Error> {
        let pk = proving_key;
        let alpha_g1 = flatten_g1::<E>(&[pk.vk.alpha_g1]);
        let beta_g1 = flatten_g1::<E>(&[pk.beta_g1]);
        let delta_g1 = flatten_g1::<E>(&[pk.delta_g1]);
        let beta_g2 = flatten_g2::<E>(&[pk.vk.beta_g2]);
        let delta_g2 = flatten_g2::<E>(&[pk.vk.delta_g2]);
        let a_query = flatten_g1::<E>(&pk.a_query);
        let b_g1_query = flatten_g1::<E>(&pk.b_g1_query);
        let b_g2_query = flatten_g2::<E>(&pk.b_g2_query);
        let h_query = flatten_g1::<E>(&pk.h_query);
        let l_query = flatten_g1::<E>(&pk.l_query);
        let view = sys::mg_pk_view {
            n_vars: pk.a_query.len() as u64,
            n_inputs: pk.vk.gamma_abc_g1.len() as u64,
            h_len: pk.h_query.len() as u64,
            alpha_g1: alpha_g1.as_ptr(),
            beta_g1: beta_g1.as_ptr(),
            delta_g1: delta_g1.as_ptr(),
            beta_g2: beta_g2.as_ptr(),
            delta_g2: delta_g2.as_ptr(),
            a_query: a_query.as_ptr(),
            b_g1_query: b_g1_query.as_ptr(),
            b_g2_query: b_g2_query.as_ptr(),
            h_query: h_query.as_ptr(),
            l_query: l_query.as_ptr(),
        };
        let mut ctx = ptr::null_mut();
        // SAFETY: every pointer of `view` outlives the call; the library copies before returning
        let opts = Self::abi_options(options)?;
        // (`options.devices`, which `opts` points into, outlives the call as well)
        check(unsafe { sys::mg_ctx_create_ex(E::CURVE, &view, &opts, &mut ctx) })?;
        Ok(Self { ctx, shape: Mutex::new(None), _engine: PhantomData })
    }

    /// [`from_bytes`](Self::from_bytes) with options and, when `checksum` is given, manta-parameters' integrity check in
    /// front of it whatever the placement (`mg_ctx_create_from_bytes_ex`).
    pub fn from_bytes_with_options(bytes: &[u8], checksum: Option<&[u8; 32]>, options: &ContextOptions<'_>) -> Result<Self, Error> {
        let opts = Self::abi_options(options)?;
        let mut ctx = ptr::null_mut();
        // SAFETY: `bytes`, `checksum` and the device list outlive the call
        check(unsafe {
            sys::mg_ctx_create_from_bytes_ex(
                E::CURVE,
                bytes.as_ptr(),
                bytes.len(),
                checksum.map_or(ptr::null(), |c| c.as_ptr()),
                &opts,
                &mut ctx,
            )
        })?;
        Ok(Self { ctx, shape: Mutex::new(None), _engine: PhantomData })
    }

    /// `impl Decode for ProvingContext` (`groth16.rs:268-288`): the arkworks `serialize_unchecked` bytes of the
    /// `ProvingKey` — the shipped `manta-parameters/data/pay/proving/*.lfs` files — handed over unparsed.
    pub fn from_bytes(bytes: &[u8]) -> Result<Self, Error> {
        let mut ctx = ptr::null_mut();
        // SAFETY: `bytes` outlives the call
        check(unsafe { sys::mg_ctx_create_from_bytes(E::CURVE, bytes.as_ptr(), bytes.len(), &mut ctx) })?;
        Ok(Self { ctx, shape: Mutex::new(None), _engine: PhantomData })
    }

    /// [`from_bytes`](Self::from_bytes) behind manta-parameters' integrity check: `checksum` is the BLAKE3 digest the caller
    /// expects (`HasChecksum::CHECKSUM`, `manta-parameters/src/lib.rs:188-212`); a key whose digest differs is refused
    /// before anything is uploaded (`manta_parameters::verify`, `lib.rs:173-177`).
    pub fn from_bytes_checked(bytes: &[u8], checksum: &[u8; 32]) -> Result<Self, Error> {
        let mut ctx = ptr::null_mut();
        // SAFETY: `bytes` and `checksum` outlive the call
        check(unsafe {
            sys::mg_ctx_create_from_bytes_checked(E::CURVE, bytes.as_ptr(), bytes.len(), checksum.as_ptr(), &mut ctx)
        })?;
        Ok(Self { ctx, shape: Mutex::new(None), _engine: PhantomData })
    }

    /// Makes sure the matrices of `matrices` are the ones on the device (once per circuit shape: they are identical
    /// for every proof of a shape, so the usual case is a lock, a comparison of five integers, and out).
    fn ensure_matrices(&self, matrices: &ConstraintMatrices<E::Fr>) -> Result<(), Error> {
        let key = (
            matrices.num_constraints,
            matrices.num_instance_variables,
            matrices.a_num_non_zero,
            matrices.b_num_non_zero,
            matrices.c_num_non_zero,
        );
        let mut shape = self.shape.lock().map_err(|_| Error)?;
        if *shape == Some(key) {
            return Ok(());
        }
        let (a, b, c) = (Csr::new::<E>(&matrices.a), Csr::new::<E>(&matrices.b), Csr::new::<E>(&matrices.c));
        // SAFETY: the three views point into `a`, `b`, `c`, alive across the call
        check(unsafe { sys::mg_ctx_set_r1cs(self.ctx, &a.view(), &b.view(), &c.view(), matrices.num_constraints as u64) })?;
        *shape = Some(key);
        Ok(())
    }

    /// Synthesizes `compiler` exactly like ark-groth16 `create_proof` and returns `instance || witness` as Montgomery limbs.
    fn assignment<C>(&self, compiler: C) -> Result<Vec<u64>, Error>
    where
        C: ConstraintSynthesizer<E::Fr>,
    {
        let cs = ConstraintSystem::new_ref();
        cs.set_optimization_goal(OptimizationGoal::Constraints);
        compiler.generate_constraints(cs.clone())?; // manta-crypto/src/arkworks/constraint/mod.rs:199-217
        cs.finalize();
        if self.shape.lock().map_err(|_| Error)?.is_none() {
            let matrices = cs.to_matrices().ok_or(Error)?;
            self.ensure_matrices(&matrices)?;
        }
        let cs = cs.borrow().ok_or(Error)?;
        let mut z = Vec::with_capacity(4 * (cs.instance_assignment.len() + cs.witness_assignment.len()));
        for f in cs.instance_assignment.iter().chain(cs.witness_assignment.iter()) {
            z.extend_from_slice(&E::fr_limbs(f));
        }
        // SAFETY: plain getter
        if z.len() as u64 != 4 * unsafe { sys::mg_ctx_num_variables(self.ctx) } {
            return Err(Error); // a different circuit than the key was made for
        }
        Ok(z)
    }

    /// The replacement of `ArkGroth16::prove(&context.proving_key, compiler, &mut SizedRng(rng))` (`groth16.rs:597`).
    /// `rng` must already be the reference's `SizedRng(rng)`: `r` and `s` are drawn here, first thing, in
    /// `create_random_proof`'s order, so a seeded RNG yields the byte-identical proof.
    pub fn prove<C, R>(&self, compiler: C, rng: &mut R) -> Result<Proof<E>, Error>
    where
        C: ConstraintSynthesizer<E::Fr>,
        R: CryptoRng + RngCore + ?Sized,
    {
        let r = E::Fr::rand(rng);
        let s = E::Fr::rand(rng);
        let z = self.assignment(compiler)?;
        let mut bytes = [0u8; 192];
        // SAFETY: z holds V x 4 limbs (checked in `assignment`), r / s four limbs each, `bytes` >= PROOF_BYTES
        check(unsafe {
            sys::mg_groth16_prove(self.ctx, z.as_ptr(), E::fr_limbs(&r).as_ptr(), E::fr_limbs(&s).as_ptr(), bytes.as_mut_ptr())
        })?;
        Proof::<E>::deserialize(&bytes[..E::PROOF_BYTES]).map_err(|_| Error)
    }

    /// Throughput mode: k proofs of this context's circuit in ONE pass of the GPU pipeline (`mg_groth16_prove_batch`).
    /// Randomness is drawn per proof, in order, each time before that proof's synthesis — the RNG stream is consumed as
    /// by k sequential `prove` calls, and proof q equals what the q-th call would have returned.
    pub fn prove_many<C, R>(&self, compilers: Vec<C>, rng: &mut R) -> Result<Vec<Proof<E>>, Error>
    where
        C: ConstraintSynthesizer<E::Fr>,
        R: CryptoRng + RngCore + ?Sized,
    {
        let k = compilers.len();
        if k == 0 {
            return Ok(Vec::new());
        }
        let (mut z, mut rs, mut ss) = (Vec::new(), Vec::with_capacity(4 * k), Vec::with_capacity(4 * k));
        for compiler in compilers {
            rs.extend_from_slice(&E::fr_limbs(&E::Fr::rand(rng)));
            ss.extend_from_slice(&E::fr_limbs(&E::Fr::rand(rng)));
            z.extend_from_slice(&self.assignment(compiler)?);
        }
        let mut bytes = vec![0u8; k * E::PROOF_BYTES];
        let mut proofs = Vec::with_capacity(k);
        for (chunk_index, chunk) in (0..k).collect::<Vec<_>>().chunks(1024).enumerate() {
            let (lo, n) = (chunk_index * 1024, chunk.len());
            let v = z.len() / k;
            // SAFETY: the slices passed hold n assignments / scalars / proofs starting at member `lo`
            check(unsafe {
                sys::mg_groth16_prove_batch(
                    self.ctx,
                    n as u64,
                    z[lo * v..].as_ptr(),
                    rs[4 * lo..].as_ptr(),
                    ss[4 * lo..].as_ptr(),
                    bytes[lo * E::PROOF_BYTES..].as_mut_ptr(),
                )
            })?;
        }
        for q in 0..k {
            proofs.push(Proof::<E>::deserialize(&bytes[q * E::PROOF_BYTES..(q + 1) * E::PROOF_BYTES]).map_err(|_| Error)?);
        }
        Ok(proofs)
    }
}

/// Device-resident prepared verifying key: `VerifyingContext<E>` (`groth16.rs:305-335`).
pub struct GpuVerifyingContext<E: Mi355xCurve> {
    vk: *mut sys::mg_vk,
    _engine: PhantomData<E>,
}

// SAFETY: as for GpuProvingContext; verification only reads the key
unsafe impl<E: Mi355xCurve> Send for GpuVerifyingContext<E> {}
unsafe impl<E: Mi355xCurve> Sync for GpuVerifyingContext<E> {}

impl<E: Mi355xCurve> Drop for GpuVerifyingContext<E> {
    #[inline]
    fn drop(&mut self) {
        // SAFETY: created by mg_vk_create*, destroyed exactly once
        unsafe { sys::mg_vk_destroy(self.vk) }
    }
}

impl<E: Mi355xCurve> GpuVerifyingContext<E> {
    /// `VerifyingContext::new(&vk)` = `ArkGroth16::process_vk` (`groth16.rs:323-327`), computed on the GPU.
    pub fn new(verifying_key: &VerifyingKey<E>) -> Result<Self, Error> {
        let vk = verifying_key;
        let alpha = flatten_g1::<E>(&[vk.alpha_g1]);
        let (beta, gamma, delta) = (flatten_g2::<E>(&[vk.beta_g2]), flatten_g2::<E>(&[vk.gamma_g2]), flatten_g2::<E>(&[vk.delta_g2]));
        let abc = flatten_g1::<E>(&vk.gamma_abc_g1);
        let mut out = ptr::null_mut();
        // SAFETY: all arrays outlive the call
        check(unsafe {
            sys::mg_vk_create(
                E::CURVE,
                alpha.as_ptr(),
                beta.as_ptr(),
                gamma.as_ptr(),
                delta.as_ptr(),
                abc.as_ptr(),
                vk.gamma_abc_g1.len() as u64,
                &mut out,
            )
        })?;
        Ok(Self { vk: out, _engine: PhantomData })
    }

    /// `impl Decode for VerifyingContext` (`groth16.rs:498-517`).
    pub fn from_bytes(bytes: &[u8]) -> Result<Self, Error> {
        let mut out = ptr::null_mut();
        // SAFETY: `bytes` outlives the call
        check(unsafe { sys::mg_vk_create_from_bytes(E::CURVE, bytes.as_ptr(), bytes.len(), &mut out) })?;
        Ok(Self { vk: out, _engine: PhantomData })
    }

    /// `impl Encode for VerifyingContext` (`groth16.rs:519-533`).
    pub fn to_bytes(&self) -> Result<Vec<u8>, Error> {
        // SAFETY: plain getter; then a buffer of exactly that size
        let mut out = vec![0u8; unsafe { sys::mg_vk_encoded_size(self.vk) }];
        check(unsafe { sys::mg_vk_encode(self.vk, out.as_mut_ptr()) })?;
        Ok(out)
    }

    fn proof_limbs(proof: &Proof<E>) -> Vec<u64> {
        let mut p = Vec::with_capacity(2 * E::G1_LIMBS + E::G2_LIMBS);
        E::push_g1(&proof.a, &mut p);
        E::push_g2(&proof.b, &mut p);
        E::push_g1(&proof.c, &mut p);
        p
    }

    /// `ArkGroth16::verify_with_processed_vk(&context.0, input, &proof.0)` (`groth16.rs:603-609`).
    pub fn verify(&self, input: &[E::Fr], proof: &Proof<E>) -> Result<bool, Error> {
        // SAFETY: plain getter
        if input.len() as u64 + 1 != unsafe { sys::mg_vk_num_inputs(self.vk) } {
            return Err(Error); // ark-groth16: SynthesisError::MalformedVerifyingKey
        }
        let inputs: Vec<u64> = input.iter().flat_map(|f| E::fr_limbs(f)).collect();
        let points = Self::proof_limbs(proof);
        let mut ok = 0;
        // SAFETY: sizes checked above
        check(unsafe { sys::mg_groth16_verify(self.vk, inputs.as_ptr(), points.as_ptr(), &mut ok) })?;
        Ok(ok == 1)
    }

    /// k proofs against this key in one pass (`mg_groth16_verify_batch`): `true` iff all verify. The 128-bit
    /// combination coefficients come from `rng`.
    pub fn verify_many<R>(&self, inputs: &[Vec<E::Fr>], proofs: &[Proof<E>], rng: &mut R) -> Result<bool, Error>
    where
        R: CryptoRng + RngCore + ?Sized,
    {
        let k = proofs.len();
        if k == 0 || inputs.len() != k {
            return Err(Error);
        }
        // SAFETY: plain getter
        let n = unsafe { sys::mg_vk_num_inputs(self.vk) } as usize;
        let (mut flat, mut points, mut rand) = (Vec::new(), Vec::new(), Vec::with_capacity(2 * k));
        for (input, proof) in inputs.iter().zip(proofs) {
            if input.len() + 1 != n {
                return Err(Error);
            }
            flat.extend(input.iter().flat_map(|f| E::fr_limbs(f)));
            points.extend_from_slice(&Self::proof_limbs(proof));
            let (lo, hi) = (rng.next_u64() | 1, rng.next_u64()); // never zero
            rand.extend_from_slice(&[lo, hi]);
        }
        let mut ok = 0;
        // SAFETY: k inputs / proofs / coefficient pairs as declared
        check(unsafe { sys::mg_groth16_verify_batch(self.vk, k as u64, flat.as_ptr(), points.as_ptr(), rand.as_ptr(), &mut ok) })?;
        Ok(ok == 1)
    }
}

/// `PairingEngineExt::has_same` / `same` (`manta-crypto/src/arkworks/pairing.rs:76-98`) on the GPU: `e(lhs.0, lhs.1) ==
/// e(rhs.0, rhs.1)`, evaluated as the single product `e(lhs.0, lhs.1) e(-rhs.0, rhs.1) == 1` (`mg_pairing_check`). The
/// trusted-setup verifier calls this on random linear combinations of whole queries (`mpc.rs:487-508`, `kzg.rs:503-521`).
pub fn same<E: Mi355xCurve>(lhs: (E::G1Affine, E::G2Affine), rhs: (E::G1Affine, E::G2Affine)) -> Result<bool, Error> {
    let g1 = flatten_g1::<E>(&[lhs.0, -rhs.0]);
    let g2 = flatten_g2::<E>(&[lhs.1, rhs.1]);
    let mut ok = 0;
    // SAFETY: two G1 and two G2 points in the C ABI's affine Montgomery layout
    check(unsafe { sys::mg_pairing_check(E::CURVE, g1.as_ptr(), g2.as_ptr(), 2, &mut ok) })?;
    Ok(ok == 1)
}

/// `PairingEngineExt::same_ratio` (`pairing.rs:101-109`): `e(lhs.0, rhs.1) == e(lhs.1, rhs.0)`.
pub fn same_ratio<E: Mi355xCurve>(lhs: (E::G1Affine, E::G1Affine), rhs: (E::G2Affine, E::G2Affine)) -> Result<bool, Error> {
    same::<E>((lhs.0, rhs.1), (lhs.1, rhs.0))
}

/// `mg_init(device)`: bind the calling thread (and contexts created from it) to one GPU. One process per GPU is the
/// deployment the benches assume; one process driving several GPUs uses the `devices` argument of
/// [`GpuProvingContext::new`] instead.
pub fn init(device: i32) -> Result<(), Error> {
    // SAFETY: no pointers involved
    check(unsafe { sys::mg_init(device) })
}

#[allow(dead_code)]
fn _assert_unused_imports<E: Mi355xCurve>(p: &E::G1Affine, f: &E::Fr) -> (bool, bool, u32) {
    (AffineCurve::is_zero(p), f.is_zero(), E::Fr::size_in_bits() as u32)
}
