//! Raw bindings to `include/mantagpu.h` — one declaration per exported symbol, same order as the header.
//! Conventions (see the header): field elements are little-endian `u64` limbs in Montgomery form — the in-memory
//! representation of `ark_ff::Fp256` / `Fp384` (`.0 .0`) — affine points are `x || y` with infinity = all zero, every
//! function returns 0 on success.
#![no_std]
#![allow(non_camel_case_types)]

use core::ffi::{c_char, c_float, c_int, c_uint, c_void};

pub type mg_curve_t = c_int;
pub const MG_BN254: mg_curve_t = 0;
pub const MG_BLS12_381: mg_curve_t = 1;

pub const MG_SUCCESS: c_int = 0;
pub const MG_ERROR_INVALID_ARGUMENT: c_int = 1;
pub const MG_ERROR_HIP: c_int = 2;
pub const MG_ERROR_OUT_OF_MEMORY: c_int = 3;
pub const MG_ERROR_DOMAIN_TOO_LARGE: c_int = 4;
pub const MG_ERROR_STATE: c_int = 5;
pub const MG_ERROR_CHECKSUM: c_int = 6;
pub const MG_TASK_A: c_uint = 1;
pub const MG_TASK_B_G1: c_uint = 2;
pub const MG_TASK_B_G2: c_uint = 4;
pub const MG_TASK_L: c_uint = 8;
pub const MG_TASK_H: c_uint = 16;

pub const MG_SCALARS_MONT: c_int = 1;
pub const MG_SCALARS_SPARSE: c_int = 2;

pub const MG_EC_ADD_MIXED: c_int = 0;
pub const MG_EC_ADD: c_int = 1;
pub const MG_EC_DOUBLE: c_int = 2;
pub const MG_EC_MUL: c_int = 3;
pub const MG_EC_SUB_MIXED: c_int = 4;
pub const MG_EC_MUL_FIXED: c_int = 5;

pub const MG_FIELD_ADD: c_int = 0;
pub const MG_FIELD_SUB: c_int = 1;
pub const MG_FIELD_MUL: c_int = 2;
pub const MG_FIELD_SQR: c_int = 3;
pub const MG_FIELD_NEG: c_int = 4;
pub const MG_FIELD_FROM_CANONICAL: c_int = 5;
pub const MG_FIELD_TO_CANONICAL: c_int = 6;
pub const MG_FIELD_INV: c_int = 7;

#[repr(C)]
pub struct mg_bases {
    _private: [u8; 0],
}
#[repr(C)]
pub struct mg_msm_job {
    _private: [u8; 0],
}
#[repr(C)]
pub struct mg_ctx {
    _private: [u8; 0],
}
#[repr(C)]
pub struct mg_vk {
    _private: [u8; 0],
}
#[repr(C)]
pub struct mg_partials_job {
    _private: [u8; 0],
}

/// `ark_groth16::ProvingKey<E>` as the library reads it (groth16.rs:216-245, field list :253-264).
#[repr(C)]
pub struct mg_pk_view {
    pub n_vars: u64,
    pub n_inputs: u64,
    pub h_len: u64,
    pub alpha_g1: *const u64,
    pub beta_g1: *const u64,
    pub delta_g1: *const u64,
    pub beta_g2: *const u64,
    pub delta_g2: *const u64,
    pub a_query: *const u64,
    pub b_g1_query: *const u64,
    pub b_g2_query: *const u64,
    pub h_query: *const u64,
    pub l_query: *const u64,
}

/// One matrix of `ConstraintSystemRef::to_matrices()` in CSR form.
/// `mg_tuning`: what a deployment decides about the library's scheduling (graph topology, streams per proof, coalescing window,
/// passes in flight, window widths of the key tables, table budget, queue placement). Fill it with `mg_tuning_init` (compiled-in
/// defaults) or `mg_get_tuning` (the process-wide values), change fields, hand it to `mg_set_tuning` or to one context through
/// `mg_ctx_opts::tuning`. No field changes a proof's bytes.
#[repr(C)]
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub struct mg_tuning {
    pub struct_size: u32,
    pub graph_mode: i32,
    pub graph_mode_batch: i32,
    pub prove_streams: i32,
    pub linear_chains: i32,
    pub coalesce_inflight: i32,
    pub coalesce_gather_us: i32,
    pub batch_inflight: i32,
    pub queue_aware: i32,
    pub msm_dedicated_queues: i32,
    pub window_bits_narrow: i32,
    pub window_bits_wide: i32,
    pub window_bits_h: i32,
    pub window_bits_g2: i32,
    pub full_table_bytes: i64,
}
pub const MG_GRAPH_OFF: i32 = 0;
pub const MG_GRAPH_SINGLE: i32 = 1;
pub const MG_GRAPH_SPLIT: i32 = 2;

/// `mg_ctx_opts`: what a deployment decides per context (placement, exchange, HBM budget of the full tables). Fill it with
/// `mg_ctx_opts_init`, then change fields; `struct_size` lets the C side accept an older, shorter struct.
#[repr(C)]
#[derive(Clone, Copy)]
pub struct mg_ctx_opts {
    pub struct_size: u32,
    pub exchange: u32,
    pub full_table_bytes: i64,
    pub devices: *const c_int,
    pub n_devices: i32,
    pub shard: i32,
    pub n_shards: i32,
    pub task_mask: u32,
    pub tuning: *const mg_tuning,
}
pub const MG_EXCHANGE_HOST: u32 = 0;
pub const MG_EXCHANGE_RCCL: u32 = 1;

#[repr(C)]
pub struct mg_csr {
    pub row_ptr: *const u32,
    pub col: *const u32,
    pub val: *const u64,
    pub nnz: u64,
}

#[repr(C)]
pub struct mg_pk_out {
    pub alpha_g1: *mut u64,
    pub beta_g1: *mut u64,
    pub delta_g1: *mut u64,
    pub beta_g2: *mut u64,
    pub gamma_g2: *mut u64,
    pub delta_g2: *mut u64,
    pub gamma_abc_g1: *mut u64,
    pub a_query: *mut u64,
    pub b_g1_query: *mut u64,
    pub b_g2_query: *mut u64,
    pub h_query: *mut u64,
    pub l_query: *mut u64,
}

extern "C" {
    // ---- runtime
    pub fn mg_init(device: c_int) -> c_int;
    pub fn mg_strerror(status: c_int) -> *const c_char;
    pub fn mg_last_error() -> *const c_char;
    pub fn mg_device_count(count: *mut c_int) -> c_int;
    pub fn mg_malloc(dptr: *mut *mut c_void, bytes: usize) -> c_int;
    pub fn mg_free(dptr: *mut c_void) -> c_int;
    pub fn mg_memcpy_h2d(dptr: *mut c_void, hptr: *const c_void, bytes: usize) -> c_int;
    pub fn mg_memcpy_d2h(hptr: *mut c_void, dptr: *const c_void, bytes: usize) -> c_int;
    pub fn mg_device_synchronize() -> c_int;
    pub fn mg_host_alloc(hptr: *mut *mut c_void, bytes: usize) -> c_int;
    pub fn mg_host_free(hptr: *mut c_void) -> c_int;
    pub fn mg_set_kernel_timing(on: c_int) -> c_int;
    pub fn mg_last_accumulate_ms() -> c_float;
    pub fn mg_last_accumulate_mhz() -> c_float;
    pub fn mg_clock_probe(iters: c_uint, memtime_mhz: *mut f64, mad_issue_per_us_per_simd: *mut f64, ms: *mut f64) -> c_int;
    pub fn mg_last_ntt_ms(out4: *mut c_float) -> c_int;
    pub fn mg_hw_queues(out2: *mut c_int) -> c_int;
    pub fn mg_last_prove_phases_ms(out10: *mut c_float) -> c_int;
    pub fn mg_last_pass_host_ms(out3: *mut c_float) -> c_int;

    // ---- variable-base MSM (ark_ec::msm::VariableBaseMSM::multi_scalar_mul)
    pub fn mg_bases_create(
        curve: mg_curve_t,
        group: c_int,
        affine_mont: *const u64,
        n: usize,
        on_device: c_int,
        precompute_window_bits: c_int,
        out: *mut *mut mg_bases,
    ) -> c_int;
    pub fn mg_bases_create_sharded(
        curve: mg_curve_t,
        group: c_int,
        affine_mont: *const u64,
        n: usize,
        devices: *const c_int,
        n_devices: c_int,
        precompute_window_bits: c_int,
        out: *mut *mut mg_bases,
    ) -> c_int;
    pub fn mg_bases_num_shards(bases: *const mg_bases) -> c_int;
    pub fn mg_bases_shard(bases: *const mg_bases, shard: c_int, device: *mut c_int, lo: *mut usize, hi: *mut usize) -> c_int;
    pub fn mg_bases_destroy(bases: *mut mg_bases);
    pub fn mg_bases_device_bytes(bases: *const mg_bases) -> usize;
    pub fn mg_msm(bases: *const mg_bases, scalars_canonical: *const u64, n: usize, out_affine_mont: *mut u64) -> c_int;
    pub fn mg_msm_launch(
        bases: *const mg_bases,
        d_scalars: *const u64,
        n: usize,
        scalar_flags: c_int,
        window_bits: c_int,
        job: *mut *mut mg_msm_job,
    ) -> c_int;
    pub fn mg_msm_launch_sharded(
        bases: *const mg_bases,
        d_scalars_per_shard: *const *const u64,
        scalar_flags: c_int,
        window_bits: c_int,
        job: *mut *mut mg_msm_job,
    ) -> c_int;
    pub fn mg_msm_finish(job: *mut mg_msm_job, out_affine_mont: *mut u64) -> c_int;
    pub fn mg_msm_result_to_device(job: *mut mg_msm_job, d_out_xyzz: *mut u64, stream: *mut c_void) -> c_int;
    pub fn mg_xyzz_limbs(curve: mg_curve_t, group: c_int) -> usize;
    pub fn mg_xyzz_sum(curve: mg_curve_t, group: c_int, xyzz: *const u64, n: usize, out_affine_mont: *mut u64) -> c_int;
    pub fn mg_points_sum(curve: mg_curve_t, group: c_int, affine_mont: *const u64, n: usize, out_affine_mont: *mut u64) -> c_int;
    pub fn mg_fixed_base_mul(
        curve: mg_curve_t,
        group: c_int,
        base_affine_mont: *const u64,
        d_scalars: *const u64,
        n: usize,
        d_out_affine_mont: *mut u64,
    ) -> c_int;
    pub fn mg_ec_elementwise(
        curve: mg_curve_t,
        group: c_int,
        op: c_int,
        a_affine: *const u64,
        b: *const u64,
        n: usize,
        out_affine: *mut u64,
    ) -> c_int;
    pub fn mg_field_op(
        field: c_int,
        op: c_int,
        repr: c_int,
        lazy_a: c_int,
        lazy_b: c_int,
        a: *const u64,
        b: *const u64,
        n: usize,
        out: *mut u64,
    ) -> c_int;
    pub fn mg_group_ntt(
        curve: mg_curve_t,
        group: c_int,
        points_affine: *const u64,
        log_n: c_uint,
        inverse: c_int,
        out_affine: *mut u64,
    ) -> c_int;
    pub fn mg_point_serialize(curve: mg_curve_t, group: c_int, affine_mont: *const u64, compressed: c_int, out: *mut u8) -> c_int;

    // ---- radix-2 NTT over Fr (ark_poly::Radix2EvaluationDomain)
    pub fn mg_ntt(curve: mg_curve_t, data_mont: *mut u64, log_n: c_uint, inverse: c_int, coset: c_int) -> c_int;
    pub fn mg_ntt_device(curve: mg_curve_t, d_data_mont: *mut u64, log_n: c_uint, inverse: c_int, coset: c_int) -> c_int;

    // ---- Groth16 key generation / proving context / prove
    pub fn mg_groth16_setup(
        curve: mg_curve_t,
        a: *const mg_csr,
        b: *const mg_csr,
        c: *const mg_csr,
        num_constraints: u64,
        n_vars: u64,
        n_inputs: u64,
        toxic_mont: *const u64,
        g1_generator: *const u64,
        g2_generator: *const u64,
        out: *const mg_pk_out,
    ) -> c_int;
    pub fn mg_ctx_create(curve: mg_curve_t, pk: *const mg_pk_view, out: *mut *mut mg_ctx) -> c_int;
    pub fn mg_tuning_init(t: *mut mg_tuning) -> c_int;
    pub fn mg_get_tuning(t: *mut mg_tuning) -> c_int;
    pub fn mg_set_tuning(t: *const mg_tuning) -> c_int;
    pub fn mg_tuning_env_names() -> *const *const c_char;
    pub fn mg_ctx_opts_init(opts: *mut mg_ctx_opts) -> c_int;
    pub fn mg_ctx_create_ex(curve: mg_curve_t, pk: *const mg_pk_view, opts: *const mg_ctx_opts, out: *mut *mut mg_ctx) -> c_int;
    pub fn mg_ctx_create_from_bytes_ex(
        curve: mg_curve_t,
        bytes: *const u8,
        len: usize,
        checksum32: *const u8,
        opts: *const mg_ctx_opts,
        out: *mut *mut mg_ctx,
    ) -> c_int;
    pub fn mg_ctx_create_sharded(
        curve: mg_curve_t,
        pk: *const mg_pk_view,
        devices: *const c_int,
        n_devices: c_int,
        out: *mut *mut mg_ctx,
    ) -> c_int;
    pub fn mg_ctx_create_shard(curve: mg_curve_t, pk: *const mg_pk_view, shard: c_int, n_shards: c_int, out: *mut *mut mg_ctx) -> c_int;
    pub fn mg_ctx_create_task(curve: mg_curve_t, pk: *const mg_pk_view, task_mask: c_uint, out: *mut *mut mg_ctx) -> c_int;
    pub fn mg_partials_slot_limbs(ctx: *const mg_ctx) -> usize;
    pub fn mg_groth16_partials_launch(
        ctx: *const mg_ctx,
        k: u64,
        z_mont: *const u64,
        d_out: *mut u64,
        stream: *mut c_void,
        job: *mut *mut mg_partials_job,
    ) -> c_int;
    pub fn mg_groth16_partials_finish(job: *mut mg_partials_job) -> c_int;
    pub fn mg_groth16_assemble(
        ctx: *const mg_ctx,
        k: u64,
        n_parts: c_int,
        parts: *const u64,
        r_mont: *const u64,
        s_mont: *const u64,
        proofs_out: *mut u8,
    ) -> c_int;
    pub fn mg_ctx_create_from_bytes(curve: mg_curve_t, bytes: *const u8, len: usize, out: *mut *mut mg_ctx) -> c_int;
    pub fn mg_ctx_create_from_bytes_checked(
        curve: mg_curve_t,
        bytes: *const u8,
        len: usize,
        checksum: *const u8,
        out: *mut *mut mg_ctx,
    ) -> c_int;
    pub fn mg_blake3(data: *const u8, len: usize, out32: *mut u8) -> c_int;
    pub fn mg_ctx_create_from_bytes_sharded(
        curve: mg_curve_t,
        bytes: *const u8,
        len: usize,
        devices: *const c_int,
        n_devices: c_int,
        out: *mut *mut mg_ctx,
    ) -> c_int;
    pub fn mg_ctx_set_r1cs(ctx: *mut mg_ctx, a: *const mg_csr, b: *const mg_csr, c: *const mg_csr, num_constraints: u64) -> c_int;
    pub fn mg_groth16_prove(ctx: *const mg_ctx, z_mont: *const u64, r_mont: *const u64, s_mont: *const u64, proof_out: *mut u8) -> c_int;
    pub fn mg_groth16_prove_batch(
        ctx: *const mg_ctx,
        k: u64,
        z_mont: *const u64,
        r_mont: *const u64,
        s_mont: *const u64,
        proofs_out: *mut u8,
    ) -> c_int;
    pub fn mg_witness_map(ctx: *const mg_ctx, z_mont: *const u64, h_out_mont: *mut u64) -> c_int;
    pub fn mg_ctx_domain_size(ctx: *const mg_ctx) -> u64;
    pub fn mg_ctx_table_bytes(ctx: *const mg_ctx, out2: *mut u64) -> c_int;
    pub fn mg_ctx_num_variables(ctx: *const mg_ctx) -> u64;
    pub fn mg_ctx_num_inputs(ctx: *const mg_ctx) -> u64;
    pub fn mg_ctx_num_shards(ctx: *const mg_ctx) -> c_int;
    pub fn mg_ctx_destroy(ctx: *mut mg_ctx);

    // ---- verification
    pub fn mg_vk_create(
        curve: mg_curve_t,
        alpha_g1: *const u64,
        beta_g2: *const u64,
        gamma_g2: *const u64,
        delta_g2: *const u64,
        gamma_abc_g1: *const u64,
        n_inputs: u64,
        out: *mut *mut mg_vk,
    ) -> c_int;
    pub fn mg_vk_create_from_bytes(curve: mg_curve_t, bytes: *const u8, len: usize, out: *mut *mut mg_vk) -> c_int;
    pub fn mg_vk_encoded_size(vk: *const mg_vk) -> usize;
    pub fn mg_vk_encode(vk: *const mg_vk, out: *mut u8) -> c_int;
    pub fn mg_vk_alpha_beta(vk: *const mg_vk, out: *mut u8) -> c_int;
    pub fn mg_vk_num_inputs(vk: *const mg_vk) -> u64;
    pub fn mg_vk_destroy(vk: *mut mg_vk);
    pub fn mg_groth16_verify(vk: *const mg_vk, inputs_mont: *const u64, proof_points: *const u64, ok: *mut c_int) -> c_int;
    pub fn mg_groth16_verify_batch(
        vk: *const mg_vk,
        k: u64,
        inputs_mont: *const u64,
        proof_points: *const u64,
        rand128: *const u64,
        ok: *mut c_int,
    ) -> c_int;
    pub fn mg_pairing_check(curve: mg_curve_t, g1_affine: *const u64, g2_affine: *const u64, n: usize, ok: *mut c_int) -> c_int;
    pub fn mg_proof_decode(curve: mg_curve_t, proof_bytes: *const u8, points_out: *mut u64) -> c_int;
}
