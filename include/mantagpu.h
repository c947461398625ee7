/* mantagpu.h -- C ABI of the MI355X-native Groth16 prove hot path (libmantagpu.so).
 *
 * This is the drop-in boundary for ONE path of Manta-Network/manta-rs: the body of
 *     manta_crypto::arkworks::groth16::Groth16::<E>::prove      (manta-crypto/src/arkworks/groth16.rs:589-600)
 * i.e. `ArkGroth16::prove(&context.proving_key, compiler, &mut SizedRng(rng))` and the arkworks 0.3
 * primitives underneath it (ark-ec VariableBaseMSM, ark-poly Radix2EvaluationDomain, ark-groth16
 * R1CStoQAP::witness_map + create_proof). Everything else in manta-rs keeps calling arkworks.
 * INTEGRATION.md shows the Rust `extern "C"` block and the replacement body of `prove`.
 *
 * Conventions (same as arkworks' in-memory data, so the Rust side passes slices without conversion):
 *   - field element  = little-endian u64 limbs (4 for Fr and BN254 Fq, 6 for BLS12-381 Fq), Montgomery
 *     form with R = 2^(64*limbs)  [ark-ff Fp256/Fp384 `.0.0`]
 *   - MSM scalars    = 4 x u64 canonical integers [`Fr::into_repr()`], unless a call says "mont"
 *   - affine point   = x || y (G2: x.c0 x.c1 y.c0 y.c1); infinity = all limbs zero (the shim writes
 *     zeros for `GroupAffine{infinity: true}`; (0,0) is never on either curve)
 *   - proof bytes    = arkworks canonical compressed A || B || C (128 B BN254, 192 B BLS12-381),
 *     byte-identical to `proof_as_bytes` (manta-crypto/src/arkworks/groth16.rs:186-195)
 *   - every function returns 0 on success; any non-zero maps to the reference's opaque unit `Error`
 *     (groth16.rs:50-60). No exceptions, no abort. mg_strerror()/mg_last_error() give detail.
 *   - objects belong to the HIP device current at creation (or to the devices of the list a `_sharded`
 *     constructor was given) and remember it: calls on them may come from any thread, whatever that
 *     thread's current device. One process per GPU (torch.distributed / RCCL between them) and one process
 *     driving several GPUs (the `_sharded` entry points) are both supported.
 *   - all entry points are re-entrant; `mg_groth16_prove` may be called concurrently on one context (the
 *     reference's `prove` takes `&ProvingContext`, groth16.rs:589-596). `mg_ctx_set_r1cs` is the one mutator:
 *     it waits for the proofs in flight on the context, and proofs that start later see the new circuit.
 *   - the library never retains host pointers after a call returns.
 */
#ifndef MANTAGPU_H
#define MANTAGPU_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum { MG_BN254 = 0, MG_BLS12_381 = 1 } mg_curve_t;

enum {
    MG_SUCCESS = 0,
    MG_ERROR_INVALID_ARGUMENT = 1,
    MG_ERROR_HIP = 2,
    MG_ERROR_OUT_OF_MEMORY = 3,
    MG_ERROR_DOMAIN_TOO_LARGE = 4, /* ark-relations SynthesisError::PolynomialDegreeTooLarge */
    MG_ERROR_STATE = 5,
    MG_ERROR_CHECKSUM = 6 /* the BLAKE3 digest of the data does not match (mg_ctx_create_from_bytes_checked) */
};

typedef struct mg_bases mg_bases;     /* a static vector of curve points resident in HBM */
typedef struct mg_msm_job mg_msm_job; /* an MSM in flight */
typedef struct mg_ctx mg_ctx;         /* device-resident ProvingContext (+ R1CS matrices) */

/* ---- runtime ---------------------------------------------------------------------------------- */
int mg_init(int device);                /* hipSetDevice + engine warm-up; optional */
const char *mg_strerror(int status);
const char *mg_last_error(void);        /* thread-local detail of the last failure */
int mg_device_count(int *count);
/* Raw HBM buffers, for hosts without a HIP binding (tests, bench). */
int mg_malloc(void **dptr, size_t bytes);
int mg_free(void *dptr);
int mg_memcpy_h2d(void *dptr, const void *hptr, size_t bytes);
int mg_memcpy_d2h(void *hptr, const void *dptr, size_t bytes);
int mg_device_synchronize(void);
/* Page-locked host memory. An assignment (`z_mont`) that lives in a buffer from mg_host_alloc -- or in any
 * memory the caller registered with HIP -- is DMA'd to the GPU straight from where it is; any other buffer is
 * first copied into the library's own pinned staging area (0.1 ms for the 1.1 MB PrivateTransfer assignment). The
 * Rust shim can collect `instance || witness` directly into such a buffer. */
int mg_host_alloc(void **hptr, size_t bytes);
int mg_host_free(void *hptr);
/* Measurement hook (bench.py roofline leg): when on, each MSM brackets its dominant kernel (bucket
 * accumulate) with HIP events on the stream it is launched on; mg_last_accumulate_ms() returns the
 * duration of the calling thread's most recently finished MSM. Off by default. */
int mg_set_kernel_timing(int on);
float mg_last_accumulate_ms(void);
float mg_last_accumulate_mhz(void); /* the shader clock that launch ran at: s_memtime ticks of its first wavefront per wall-clock second */
/* With kernel timing on, the same for the calling thread's last mg_ntt / mg_ntt_device -- out4 = { whole call on the device,
 * conversion in, butterfly passes, conversion out } in ms -- and for its last SINGLE proof, which is then enqueued with plain
 * launches instead of the captured graphs -- out10 = { upload of z, witness map, MSM a, b_g1, b_g2, l, h (each on its own
 * stream), part A (all but the G2 MSM) from upload to join, the G2 chain from upload to its end, host assembly after the
 * GPU } in ms: the per-phase report SURVEY.md 8(d) asks of BASELINE configs[2]. */
/* Clock probe: two wavefronts per SIMD on every CU spin on dependent v_mad_u64_u32 chains (the instruction the MSM is
 * bound by) for `iters` x 32 multiply-adds each, reading the shader clock counter (s_memtime) and the constant-rate wall
 * clock around the loop: *memtime_mhz = the rate the counter ran at under that load, *mad_issue_per_us_per_simd = wave-level
 * multiply-adds a SIMD issued per microsecond (the measured issue peak, no clock assumed), *ms = duration of the probe. */
int mg_clock_probe(unsigned iters, double *memtime_mhz, double *mad_issue_per_us_per_simd, double *ms);
int mg_last_ntt_ms(float out4[4]);
int mg_last_prove_phases_ms(float out10[10]);
/* Always on (three clock reads per pass): the HOST side of the calling thread's last proving pass on an unsharded context --
 * out3 = { staging z and enqueuing the pass (graph launches or ~100 kernel launches), waiting for the GPU, assembly after the
 * last MSM result arrived } in ms. A pass whose first figure approaches its wall time is bound by launches, not by kernels. */
int mg_last_pass_host_ms(float out3[3]);
/* Introspection: the hardware queues the library found behind the current device's streams -- out2 = { normal priority, high
 * priority }, 0 / 0 before the first proving context exists or with MANTA_QUEUE_AWARE=0. The HIP runtime multiplexes streams onto a
 * few hardware queues per priority level and kernels of streams that share one run one behind the other; the library measures the
 * sharing once per device (csrc/queues.hip) and gives every single-proof slot three streams on three different queues. */
int mg_hw_queues(int out2[2]);

/* ---- variable-base MSM: replaces ark_ec::msm::VariableBaseMSM::multi_scalar_mul(bases, scalars)
 *      (ark-ec 0.3.0 msm/variable_base.rs; called 5x per proof from ark-groth16 create_proof, reached
 *      from manta-crypto/src/arkworks/groth16.rs:597) ------------------------------------------------ */
/* Register `n` affine points (host pointer, or device pointer if on_device). group = 1 (G1) or 2 (G2).
 * precompute_window_bits > 0 additionally stores 2^(c*w)*P for every window (HBM for speed: all
 * windows then share one bucket set and no doubling chain remains); 0 = plain bases;
 * -12 .. -2 = FULL tables of window width c = -precompute_window_bits: every multiple m*2^(c*w)*P, m = 1 .. 2^(c-1), so
 * that a signed digit addresses its summand and the MSM is one plain sum (no buckets, no sort, no bucket reduce) --
 * ceil(bits/c) * 2^(c-1) points per base, fewer than 2^31 in all, for fixed proving-key queries of proof size. */
int mg_bases_create(mg_curve_t curve, int group, const uint64_t *affine_mont, size_t n, int on_device,
                    int precompute_window_bits, mg_bases **out);
/* The same vector range-sharded over a list of devices ("MSM shards by scalar/base
 * range across the GPUs of one node"): shard g = points [n*g/G, n*(g+1)/G) on devices[g], host pointer only. A
 * device may appear more than once (functional testing of the multi-GPU path on one GPU). */
int mg_bases_create_sharded(mg_curve_t curve, int group, const uint64_t *affine_mont, size_t n, const int *devices,
                            int n_devices, int precompute_window_bits, mg_bases **out);
int mg_bases_num_shards(const mg_bases *bases);
int mg_bases_shard(const mg_bases *bases, int shard, int *device, size_t *lo, size_t *hi);
void mg_bases_destroy(mg_bases *bases);
size_t mg_bases_device_bytes(const mg_bases *bases);
/* Host-to-host convenience: result = sum_i scalars[i] * bases[i], i < n <= len(bases).
 * scalars: n x 4 u64 canonical. out: affine Montgomery (infinity = zeros). */
int mg_msm(const mg_bases *bases, const uint64_t *scalars_canonical, size_t n, uint64_t *out_affine_mont);
/* Scalars already resident in HBM (the timed path). `scalar_flags`: MG_SCALARS_MONT -- the scalars are Montgomery Fr
 * and are converted on the device (`into_repr`), else canonical; MG_SCALARS_SPARSE -- a hint that many digits are
 * zero (a witness: mostly 0 / 1 / small values), the zero digits are then compacted away before the sort.
 * window_bits = 0 lets the library choose. Returns immediately; mg_msm_finish waits and writes the affine result. */
#define MG_SCALARS_MONT 1
#define MG_SCALARS_SPARSE 2
int mg_msm_launch(const mg_bases *bases, const uint64_t *d_scalars, size_t n, int scalar_flags, int window_bits,
                  mg_msm_job **job);
/* Sharded bases: d_scalars_per_shard[g] points at the scalars of shard g's range, resident on shard g's device.
 * One Pippenger pass per device, all in flight at once; the per-device partial points are the only data exchanged
 * and mg_msm_finish adds them. (mg_msm, the host-to-host call, accepts sharded bases too and uploads the slices.) */
int mg_msm_launch_sharded(const mg_bases *bases, const uint64_t *const *d_scalars_per_shard, int scalar_flags,
                          int window_bits, mg_msm_job **job);
int mg_msm_finish(mg_msm_job *job, uint64_t *out_affine_mont); /* waits, folds, adds the shards' partial points, frees the job */
/* The result of a job left on the DEVICE instead of the host: the window sums are folded by one more small kernel behind
 * the MSM (bases with precomputed multiples only, else MG_ERROR_STATE: plain bases end in a 255-doubling Horner chain,
 * a host job) and the point -- X | Y | ZZ | ZZZ in arkworks' Montgomery limbs, x = X/ZZ, y = Y/ZZZ, ZZ = 0 for infinity;
 * mg_xyzz_limbs() u64 -- is written to d_out_xyzz. `stream` (a hipStream_t; NULL = the default stream) is made to wait for it, so a
 * consumer on another stream -- the RCCL all_gather of the range-sharded MSM (partial points gathered
 * from device memory) -- needs no host synchronisation between launch and collective. The job is still released with
 * mg_msm_finish (out_affine_mont may be NULL). Single-shard jobs only. */
int mg_msm_result_to_device(mg_msm_job *job, uint64_t *d_out_xyzz, void *stream);
size_t mg_xyzz_limbs(mg_curve_t curve, int group); /* u64 per XYZZ point: 16 / 24 (G1), 32 / 48 (G2) */
/* sum of n XYZZ points (host memory, the layout above) -> one affine point: the N-term sum after the all_gather */
int mg_xyzz_sum(mg_curve_t curve, int group, const uint64_t *xyzz, size_t n, uint64_t *out_affine_mont);
/* sum of the registered points themselves (multi-GPU partial-point reduction, tests) */
int mg_points_sum(mg_curve_t curve, int group, const uint64_t *affine_mont, size_t n, uint64_t *out_affine_mont);
/* [k_i] * base for n canonical scalars in HBM -> n affine points in HBM (fixed-base batch multiply:
 * synthetic base generation; key generation as in ark-groth16 generate_parameters) */
int mg_fixed_base_mul(mg_curve_t curve, int group, const uint64_t *base_affine_mont, const uint64_t *d_scalars,
                      size_t n, uint64_t *d_out_affine_mont);
/* Element-wise group operations on host arrays of n affine points, computed on the GPU with the MSM kernels' own
 * device functions (mixed / general addition, doubling, double-and-add) and normalised with the batched
 * Montgomery-trick inversion -- the primitive menu the reference itself benchmarks and cross-checks
 * (manta-benchmark/src/ecc.rs:30-128, consistency tests :138-172):
 *   MG_EC_ADD_MIXED  out = a + b   `projective += affine`            (ecc.rs:69-74)
 *   MG_EC_ADD        out = a + b   `projective += projective`        (ecc.rs:78-83)
 *   MG_EC_DOUBLE     out = 2a      (b unused)
 *   MG_EC_MUL        out = [k]a    b = n x 4 u64 canonical scalars   (ecc.rs:87-101)
 *   MG_EC_SUB_MIXED  out = a - b
 * out = n affine points (batch normalisation, ecc.rs:114-119). */
#define MG_EC_ADD_MIXED 0
#define MG_EC_ADD 1
#define MG_EC_DOUBLE 2
#define MG_EC_MUL 3
#define MG_EC_SUB_MIXED 4
/*   MG_EC_MUL_FIXED  out = [k]a    b = ONE canonical scalar (4 u64) for all points: `batch_mul_fixed_scalar`
 *                                   (manta-trusted-setup/src/util.rs:440-445; Groth16 MPC `contribute`, mpc.rs:451-468);
 *   MG_EC_MUL with per-point scalars is `batch_mul_pointwise` (util.rs:447-455; kzg `Accumulator::update`, kzg.rs:444-468) */
#define MG_EC_MUL_FIXED 5
int mg_ec_elementwise(mg_curve_t curve, int group, int op, const uint64_t *a_affine, const uint64_t *b, size_t n,
                      uint64_t *out_affine);
/* Element-wise prime-field arithmetic on the GPU with the kernels' own device functions: the direct parity surface
 * for ark-ff 0.3 Fp256 / Fp384 (`ark_ff::Fp256<..>::{add_assign, sub_assign, mul_assign, square, neg, inverse, from_repr,
 * into_repr}`; re-exported manta-crypto/src/arkworks/mod.rs:25-35).
 *   field: 0 BN254 Fr, 1 BN254 Fq, 2 BLS12-381 Fr, 3 BLS12-381 Fq;  elements = 4 / 4 / 4 / 6 u64 limbs, Montgomery
 *          (MG_FIELD_FROM_CANONICAL takes, MG_FIELD_TO_CANONICAL returns, plain integers)
 *   repr:  0 = the saturated 32-bit-limb Montgomery arithmetic of the NTT / SpMV kernels (ABI format);
 *          1 = the reduced-radix lazily-reduced arithmetic inside the MSM kernels: a (b) is converted, lazy_a (lazy_b)
 *              in 0..3 times p is added so that the operation runs on a non-canonical representative, and the result
 *              comes back canonical. out[i] = a[i] op b[i]; b is ignored by the unary operations. */
#define MG_FIELD_ADD 0
#define MG_FIELD_SUB 1
#define MG_FIELD_MUL 2
#define MG_FIELD_SQR 3
#define MG_FIELD_NEG 4
#define MG_FIELD_FROM_CANONICAL 5
#define MG_FIELD_TO_CANONICAL 6
#define MG_FIELD_INV 7
int mg_field_op(int field, int op, int repr, int lazy_a, int lazy_b, const uint64_t *a, const uint64_t *b, size_t n,
                uint64_t *out);
/* Radix-2 (inverse) NTT over a vector of 2^log_n GROUP elements, natural order in and out: ark-poly
 * `Radix2EvaluationDomain::{fft, ifft}` applied to points -- how `mpc::initialize` turns the powers of tau into the
 * Lagrange basis (manta-trusted-setup/src/groth16/mpc.rs:378-381). Host arrays of affine Montgomery points. */
int mg_group_ntt(mg_curve_t curve, int group, const uint64_t *points_affine, unsigned log_n, int inverse, uint64_t *out_affine);
/* arkworks canonical serialisation of one affine point (compressed: 32/48/64/96 B) */
int mg_point_serialize(mg_curve_t curve, int group, const uint64_t *affine_mont, int compressed, uint8_t *out);

/* ---- radix-2 NTT over Fr: replaces ark_poly::Radix2EvaluationDomain::{fft,ifft,coset_fft,
 *      coset_ifft}_in_place (ark-poly 0.3.0; used 7x per proof by R1CStoQAP::witness_map) ------------ */
/* data: 2^log_n Montgomery Fr elements, natural order in and out, transformed in place. */
int mg_ntt(mg_curve_t curve, uint64_t *data_mont, unsigned log_n, int inverse, int coset);
int mg_ntt_device(mg_curve_t curve, uint64_t *d_data_mont, unsigned log_n, int inverse, int coset);

/* ---- Groth16 proving context: replaces ProvingContext<E>{proving_key} (groth16.rs:216-245) +
 *      ark_groth16::create_proof ----------------------------------------------------------------------- */
typedef struct {
    uint64_t n_vars;   /* V: instance + witness variables (len of a_query) */
    uint64_t n_inputs; /* P: instance variables incl. the constant one */
    uint64_t h_len;    /* len(h_query): D-1 (ark setup) or D (MPC keys, mpc.rs:372-377) */
    const uint64_t *alpha_g1, *beta_g1, *delta_g1; /* G1 affine */
    const uint64_t *beta_g2, *delta_g2;            /* G2 affine */
    const uint64_t *a_query;    /* V   x G1 */
    const uint64_t *b_g1_query; /* V   x G1 */
    const uint64_t *b_g2_query; /* V   x G2 */
    const uint64_t *h_query;    /* h_len x G1 */
    const uint64_t *l_query;    /* V-P x G1 */
} mg_pk_view;

typedef struct {
    const uint32_t *row_ptr; /* m+1 */
    const uint32_t *col;     /* nnz  */
    const uint64_t *val;     /* nnz x 4, Montgomery Fr */
    uint64_t nnz;
} mg_csr;

/* ---- key generation: the expensive part of `Groth16::compile` (manta-crypto/src/arkworks/groth16.rs:571-586 ->
 *      ark-groth16 0.3 generate_parameters): every group element of the proving and verifying key is a fixed-base
 *      multiple of a generator, 3V + D of them in G1 and V in G2 -- computed on the GPU. The caller supplies what
 *      the reference draws from its RNG, in its order: alpha, beta, gamma, delta (then the two generators), tau --
 *      all Fr in Montgomery form -- so a seeded RNG reproduces the reference's key. Outputs are caller-allocated
 *      arrays laid out like the fields of the mg_pk_view struct -- affine Montgomery, infinity = zeros: gamma_abc_g1[n_inputs],
 *      a_query / b_g1_query / b_g2_query[n_vars], h_query[D - 1] with D = next_pow2(m + n_inputs),
 *      l_query[n_vars - n_inputs]. */
typedef struct mg_pk_out {
    uint64_t *alpha_g1, *beta_g1, *delta_g1;
    uint64_t *beta_g2, *gamma_g2, *delta_g2;
    uint64_t *gamma_abc_g1;
    uint64_t *a_query, *b_g1_query, *b_g2_query, *h_query, *l_query;
} mg_pk_out;
int mg_groth16_setup(mg_curve_t curve, const mg_csr *a, const mg_csr *b, const mg_csr *c, uint64_t num_constraints,
                     uint64_t n_vars, uint64_t n_inputs, const uint64_t *toxic_mont /* 5 x 4: alpha beta gamma delta tau */,
                     const uint64_t *g1_generator, const uint64_t *g2_generator, const mg_pk_out *out);

/* Uploads and re-lays the proving key once (lifetime = the Rust ProvingContext). */
int mg_ctx_create(mg_curve_t curve, const mg_pk_view *pk, mg_ctx **out);
/* The same with everything a deployment decides per context in ONE struct (the entry points below are shorthands for it):
 * a signer holds three contexts at once -- `MultiProvingContext { to_private, private_transfer, to_public }`,
 * manta-accounting/src/transfer/canonical.rs:561-588 -- so what a context may spend on speed is a property of the context,
 * not of the process.
 *   struct_size       sizeof(mg_ctx_opts), written by mg_ctx_opts_init: the struct can grow without breaking callers
 *   exchange          how the partial points of a context sharded over `devices` meet: MG_EXCHANGE_HOST -- through pinned host
 *                     memory, summed on the host -- or MG_EXCHANGE_RCCL -- every device folds its five partial points per proof
 *                     on the GPU, one grouped ncclAllGather of 5 x 256 B (BN254) per device and proof over xGMI inside the
 *                     library (ncclCommInitAll over the list: no duplicate devices), assembly as in mg_groth16_assemble.
 *                     BASELINE north_star: "final RCCL reduce of partial EC points over xGMI" behind the C ABI; caller
 *                     manta-accounting/src/transfer/mod.rs:695-715. librccl.so is loaded when first asked for; if it is
 *                     missing the call fails with MG_ERROR_STATE (no silent fallback).
 *   full_table_bytes  HBM this context may spend on FULL tables of its five queries together (single proofs run on them:
 *                     no sort, no bucket reduce on their latency chain), per device: the widest windows that fit are chosen,
 *                     0 = none (bucket tables only), negative = the default, a tenth of the device's HBM (28.8 GB on an
 *                     MI355X: three contexts use 30 %). Never more than 40 % of what is free at creation. Negative: the tuning's
 *                     full_table_bytes applies (where MANTA_FULL_TABLE_GB lands), else the default.
 *   devices/n_devices range-shard every MSM over these devices inside this process (mg_ctx_create_sharded)
 *   shard/n_shards    this process holds one slice (mg_ctx_create_shard); n_shards <= 1: the whole key
 *   task_mask         mg_ctx_create_task; 0 or 0x1f: all five MSMs
 * At most one of the three placements may be used. */
#define MG_EXCHANGE_HOST 0u
#define MG_EXCHANGE_RCCL 1u
/* ---- tuning: everything a DEPLOYMENT decides about how the library schedules its work, as one struct (none of it changes a
 *      result -- tests/test_gpu_profiles.py proves every field and every environment name below leaves proof bytes unchanged).
 *      Reference counterpart: `ProvingContext` carries no tuning at all (manta-crypto/src/arkworks/groth16.rs:216-245): the
 *      arkworks prover has one schedule; this struct is what the GPU path adds next to it. A Rust host fills it once
 *      (rust/mantagpu-sys mirrors it) instead of exporting environment variables. Process-wide values: mg_get_tuning /
 *      mg_set_tuning (contexts copy them when they are created); per context: mg_ctx_opts.tuning.
 *      The shipped library reads the environment in ONE place, once, as the initial process-wide values -- the 14 variables of
 *      mg_tuning_env_names() (plus MANTA_RCCL_LIB, the path of librccl.so): MANTA_GRAPH (single | split | off),
 *      MANTA_GRAPH_BATCH, MANTA_PROVE_STREAMS, MANTA_Z3_LINEAR, MANTA_COALESCE, MANTA_COALESCE_GATHER_US, MANTA_BATCH_INFLIGHT,
 *      MANTA_QUEUE_AWARE, MANTA_MSM_DEDICATED_QUEUES, MANTA_PROVE_C / _CW / _CH / _CG2, MANTA_FULL_TABLE_GB. Every other knob of
 *      the measurement campaigns is compiled in at its measured optimum (a unit rebuilt with -DMG_DIAG reads it again).
 *   struct_size           sizeof(mg_tuning), written by mg_tuning_init / mg_get_tuning
 *   graph_mode            replay of a pass's GPU side as hipGraphs: 1 = two graphs per pass (default), 2 = six single-stream graphs,
 *                         0 = eager launches
 *   graph_mode_batch      the same for passes of >= 4 proofs; -1 = as graph_mode
 *   prove_streams         streams of a forked pass: 3, 4, 5 or 6 (default)
 *   linear_chains         single proofs as three linear graphs on three hardware queues: 0 never, 1 a lone proof only, 2 also
 *                         beside other passes, 3 (default) the same with the yielding chain chosen by what the proof runs beside
 *   coalesce_inflight     passes of coalesced concurrent mg_groth16_prove calls on the GPU: 0 = no coalescing, default 2, <= 4
 *   coalesce_gather_us    how long the leader of such a pass waits for the callers of the pass that has just ended (default 100)
 *   batch_inflight        passes of one mg_groth16_prove_batch call in flight (default 3)
 *   queue_aware           1 (default): single-proof slots get streams on measured hardware queues; 0: plain pooled streams
 *   msm_dedicated_queues  a stand-alone MSM (mg_msm_launch) on a stream with a hardware queue of its own: its pipelined rate then no
 *                         longer depends on the streams the rest of the process created (390-398 Mscalar/s at 2^20 for every
 *                         creation order against 317-394). 1 (default) = only while NO proving / verifying context is alive in
 *                         the process (beside proof passes the dedicated queues cost far more than they give), 2 = always,
 *                         0 = never. Such streams are BLOCKING streams: they order themselves against the host's NULL-stream work
 *                         (the library itself puts nothing on the NULL stream)
 *   window_bits_*         window widths of a context's key tables, 0 = the library's choice: narrow = the latency tables of
 *                         a / b_g1 / l (setting it forces ONE width for every bucket table and leaves the full and wide tables
 *                         out), wide = the batched-pass tables, h = the h query, g2 = b_g2
 *   full_table_bytes      default HBM budget of a context's full tables (mg_ctx_opts.full_table_bytes >= 0 takes precedence);
 *                         -1 = a tenth of the device's HBM, 0 = none */
typedef struct mg_tuning {
    uint32_t struct_size;
    int32_t graph_mode;
    int32_t graph_mode_batch;
    int32_t prove_streams;
    int32_t linear_chains;
    int32_t coalesce_inflight;
    int32_t coalesce_gather_us;
    int32_t batch_inflight;
    int32_t queue_aware;
    int32_t msm_dedicated_queues;
    int32_t window_bits_narrow;
    int32_t window_bits_wide;
    int32_t window_bits_h;
    int32_t window_bits_g2;
    int64_t full_table_bytes;
} mg_tuning;
int mg_tuning_init(mg_tuning *t);      /* the compiled-in defaults (no environment) */
int mg_get_tuning(mg_tuning *t);       /* the process-wide values in force */
int mg_set_tuning(const mg_tuning *t); /* validated as a whole (MG_ERROR_INVALID_ARGUMENT leaves everything as it was); contexts
                                          created afterwards use it, the environment no longer applies */
/* NULL-terminated list of the environment variables the shipped library reads (the tuning table + MANTA_RCCL_LIB) */
const char *const *mg_tuning_env_names(void);

typedef struct mg_ctx_opts {
    uint32_t struct_size;
    uint32_t exchange;
    int64_t full_table_bytes;
    const int *devices;
    int32_t n_devices;
    int32_t shard, n_shards;
    uint32_t task_mask;
    const mg_tuning *tuning; /* this context's tuning (read during the call); NULL = the process-wide values */
} mg_ctx_opts;
int mg_ctx_opts_init(mg_ctx_opts *opts); /* defaults: host exchange, default table budget, current device, whole key */
int mg_ctx_create_ex(mg_curve_t curve, const mg_pk_view *pk, const mg_ctx_opts *opts /* NULL = defaults */, mg_ctx **out);
/* ... and from the key's wire format (mg_ctx_create_from_bytes below), checksum32 = the expected BLAKE3 digest or NULL:
 * the integrity check of mg_ctx_create_from_bytes_checked in front of EVERY placement. */
int mg_ctx_create_from_bytes_ex(mg_curve_t curve, const uint8_t *bytes, size_t len, const uint8_t *checksum32,
                                const mg_ctx_opts *opts, mg_ctx **out);
/* The proving key range-sharded over a list of devices (BASELINE configs[3]: "PrivateTransfer full proof, MSM
 * sharded across 8 GPUs"): device g holds the g-th contiguous slice of every query with its window tables. A proof
 * uploads the assignment to every device, each recomputes the witness map (cheaper than broadcasting h, SURVEY.md
 * 8(e)) and runs its five partial MSMs; the 5 x G partial points are added on the host by the same code that folds
 * the single-GPU results. Proof bytes are identical to the single-device context's. devices may repeat. */
int mg_ctx_create_sharded(mg_curve_t curve, const mg_pk_view *pk, const int *devices, int n_devices, mg_ctx **out);
int mg_ctx_create_from_bytes_sharded(mg_curve_t curve, const uint8_t *bytes, size_t len, const int *devices,
                                     int n_devices, mg_ctx **out);
/* One process per GPU (what `python -m torch.distributed.run` starts; BASELINE configs[3] "MSM sharded across 8 x MI355X via
 * RCCL/xGMI"; caller: manta-accounting/src/transfer/mod.rs:695-715 -> groth16.rs:589-600): THIS process holds shard
 * `shard` of `n_shards` -- the same contiguous slices mg_ctx_create_sharded gives device g -- on the current device.
 *   mg_groth16_partials_launch: uploads z (every rank has the whole assignment and recomputes the witness map), runs the five
 *     MSMs over this shard's slices and leaves, per proof, the five partial results a | b_g1 | b_g2 | l | h folded on the
 *     device in slots of mg_partials_slot_limbs() u64 (XYZZ points as in mg_msm_result_to_device, G1 ones padded) at
 *     d_out[k][5][slot]; `stream` (hipStream_t, e.g. the stream RCCL runs on) waits for them. k <= 32. No host sync.
 *   mg_groth16_partials_finish: returns the pass's slot once the consumer has read d_out.
 *   mg_groth16_assemble: parts = n_parts gathered copies of [k][5][slot] in HOST memory (one fused all_gather of <= 1.9 KB
 *     per rank and proof); adds them and finishes the proofs exactly like mg_groth16_prove -- bytes identical to the
 *     single-GPU context's. Host-only work: any rank (or all) may call it. */
/* A shard context with n_shards > 1 holds ONE slice of every query: mg_groth16_prove / mg_groth16_prove_batch on it return
 * MG_ERROR_STATE (a proof built from one slice's MSMs would be silently invalid); only the partials interface and
 * mg_witness_map work on it. */
int mg_ctx_create_shard(mg_curve_t curve, const mg_pk_view *pk, int shard, int n_shards, mg_ctx **out);
/* The alternative placement of SURVEY.md 8(e): TASK-parallel -- this process holds the whole key and computes the MSMs of
 * task_mask in full (bit 0 a, 1 b_g1, 2 b_g2, 3 l, 4 h; the witness map only when it owns h); for the others
 * mg_groth16_partials_launch writes the point at infinity, so the same gather + mg_groth16_assemble finish the proof. Five point
 * transfers instead of a sum over ranks, at most five busy GPUs. mg_groth16_prove on such a context returns MG_ERROR_STATE. */
#define MG_TASK_A 1u
#define MG_TASK_B_G1 2u
#define MG_TASK_B_G2 4u
#define MG_TASK_L 8u
#define MG_TASK_H 16u
int mg_ctx_create_task(mg_curve_t curve, const mg_pk_view *pk, unsigned task_mask, mg_ctx **out);
typedef struct mg_partials_job mg_partials_job;
size_t mg_partials_slot_limbs(const mg_ctx *ctx);
int mg_groth16_partials_launch(const mg_ctx *ctx, uint64_t k, const uint64_t *z_mont, uint64_t *d_out, void *stream,
                               mg_partials_job **job);
int mg_groth16_partials_finish(mg_partials_job *job);
int mg_groth16_assemble(const mg_ctx *ctx, uint64_t k, int n_parts, const uint64_t *parts, const uint64_t *r_mont,
                        const uint64_t *s_mont, uint8_t *proofs_out);
/* Same, from the key's wire format: arkworks 0.3 `ProvingKey::serialize_unchecked` bytes (uncompressed points,
 * no curve checks) exactly as `ProvingContext::decode` reads them (manta-crypto/src/arkworks/groth16.rs:268-288)
 * and `generate_parameters` / manta-parameters ship them (data/pay/proving/ *.lfs). */
int mg_ctx_create_from_bytes(mg_curve_t curve, const uint8_t *bytes, size_t len, mg_ctx **out);
/* The same behind manta-parameters' integrity check: `manta_parameters::verify(data, checksum)` = `blake3::hash(data) ==
 * checksum` (manta-parameters/src/lib.rs:173-177; `Get::get` refuses a file whose digest differs from data.checkfile,
 * lib.rs:150-170). checksum = the 32-byte BLAKE3 digest the caller expects (`HasChecksum::CHECKSUM`); a mismatch returns
 * MG_ERROR_CHECKSUM and nothing is uploaded. mg_blake3 is the digest itself (host code, any length). */
int mg_ctx_create_from_bytes_checked(mg_curve_t curve, const uint8_t *bytes, size_t len, const uint8_t checksum[32],
                                     mg_ctx **out);
int mg_blake3(const uint8_t *data, size_t len, uint8_t out32[32]);
/* Once per circuit shape: the matrices of `cs.to_matrices()` (identical for every proof of a shape). Validated
 * in full before anything changes (row_ptr monotone from 0 to nnz, column indices < V); a rejected call leaves
 * the context as it was. */
int mg_ctx_set_r1cs(mg_ctx *ctx, const mg_csr *a, const mg_csr *b, const mg_csr *c, uint64_t num_constraints);
/* One proof. z = instance || witness (V x 4 u64 Montgomery), r, s = the two blinding scalars drawn by
 * the shim with the reference's own RNG in create_random_proof's order (Montgomery).
 * proof_out: 128 B (BN254) / 192 B (BLS12-381). Re-entrant on one context: calls that arrive while two passes are on
 * the GPU are grouped into ONE batched pass by the next caller (the reference's simulation drives a context from six
 * threads, manta-pay/src/bin/simulation.rs:36-38); the bytes of a proof do not depend on the grouping. */
int mg_groth16_prove(const mg_ctx *ctx, const uint64_t *z_mont, const uint64_t r_mont[4], const uint64_t s_mont[4],
                     uint8_t *proof_out);
/* k proofs of the context's circuit (throughput mode: a wallet / ledger simulation proving many transfers
 * against one ProvingContext, manta-pay/src/simulation/mod.rs:75-79; the shim collects k `prove` calls or
 * exposes a `prove_many`). Up to 32 proofs are ONE pass of the GPU pipeline; a longer batch is streamed
 * through as passes of 32 with three in flight (library threads), so one call with k >= 96 keeps the
 * GPU as busy as three callers would. z = k assignments back to back (k x V x 4 u64), r, s = k blinding
 * scalars each (k x 4 u64), proofs_out = k proofs back to back. Proof q is byte-identical to
 * mg_groth16_prove(ctx, z_q, r_q, s_q). 1 <= k <= 1024. */
int mg_groth16_prove_batch(const mg_ctx *ctx, uint64_t k, const uint64_t *z_mont, const uint64_t *r_mont,
                           const uint64_t *s_mont, uint8_t *proofs_out);
/* h = R1CStoQAP::witness_map(z): D x 4 u64 Montgomery coefficients (tests / parity) */
int mg_witness_map(const mg_ctx *ctx, const uint64_t *z_mont, uint64_t *h_out_mont);
uint64_t mg_ctx_domain_size(const mg_ctx *ctx);
/* HBM held by the context's key tables, bytes: out[0] = bucket tables (2^(c*w)*P per window, two widths), out[1] = FULL tables
 * (every multiple of every window; single proofs run on them; mg_ctx_opts.full_table_bytes bounds them, 0 = none). The h
 * tables exist once mg_ctx_set_r1cs has run. Summed over the devices of a sharded context. */
int mg_ctx_table_bytes(const mg_ctx *ctx, uint64_t out2[2]);
uint64_t mg_ctx_num_variables(const mg_ctx *ctx); /* V: the length every assignment must have */
uint64_t mg_ctx_num_inputs(const mg_ctx *ctx);    /* P */
int mg_ctx_num_shards(const mg_ctx *ctx);
void mg_ctx_destroy(mg_ctx *ctx);

/* ---- verification: replaces `Groth16::verify` (manta-crypto/src/arkworks/groth16.rs:603-609 -> ark-groth16 0.3
 *      verify_with_processed_vk) and `VerifyingContext` with its codec (groth16.rs:305-539) -------------------------- */
typedef struct mg_vk mg_vk; /* device-resident PreparedVerifyingKey */
/* `VerifyingContext::new(&vk)` = ArkGroth16::process_vk (groth16.rs:323-327): from the key's five components (affine
 * Montgomery; gamma_abc_g1 = n_inputs points incl. the one for the constant input) everything the prepared key holds is
 * computed on the GPU -- the G2Prepared line coefficients of -gamma_g2 and -delta_g2 and e(alpha_g1, beta_g2) with
 * arkworks' final exponentiation. */
int mg_vk_create(mg_curve_t curve, const uint64_t *alpha_g1, const uint64_t *beta_g2, const uint64_t *gamma_g2,
                 const uint64_t *delta_g2, const uint64_t *gamma_abc_g1, uint64_t n_inputs, mg_vk **out);
/* `impl Decode for VerifyingContext` (groth16.rs:498-517): the wire format of manta-parameters' verifying-key files
 * (vk compressed | e(alpha, beta) | two G2Prepared). Checked like `CanonicalDeserialize::deserialize`: canonical field
 * encodings, points on the curve and in the prime-order subgroup, no trailing bytes. */
int mg_vk_create_from_bytes(mg_curve_t curve, const uint8_t *bytes, size_t len, mg_vk **out);
/* `impl Encode for VerifyingContext` (groth16.rs:519-533): byte-identical to the reference's file for the same key. */
size_t mg_vk_encoded_size(const mg_vk *vk);
int mg_vk_encode(const mg_vk *vk, uint8_t *out);
int mg_vk_alpha_beta(const mg_vk *vk, uint8_t *out /* 12 Fq elements, 384 / 576 B */);
uint64_t mg_vk_num_inputs(const mg_vk *vk);
void mg_vk_destroy(mg_vk *vk);
/* One proof. inputs = the n_inputs - 1 public inputs (Montgomery Fr, `Input = Vec<E::Fr>`), proof_points = a | b | c
 * affine Montgomery (the in-memory ark_groth16::Proof<E>; mg_proof_decode gives it from the 128 / 192 proof bytes).
 * *ok = 1 iff the proof verifies; the return value reports only operational failures. */
int mg_groth16_verify(const mg_vk *vk, const uint64_t *inputs_mont, const uint64_t *proof_points, int *ok);
/* k proofs against one key by random linear combination: rand128 = k x 2 u64 -- 128 random bits per proof from the
 * caller's RNG, not both words zero. Proof i enters with the coefficient k1 + lambda k2, (k1, k2) its two words and
 * lambda the eigenvalue of G1's endomorphism (x, y) -> (beta x, y): 2^128 - 1 distinct non-zero values, and k_i A_i
 * then costs 64 doublings instead of 128. k + 3 Miller loops (two wavefronts each), k scalar multiplications, two small
 * MSMs and one final exponentiation. *ok = 1 iff ALL k proofs verify (up to the 2^-128 soundness error of the
 * combination); on 0 fall back to mg_groth16_verify to find the offender. */
int mg_groth16_verify_batch(const mg_vk *vk, uint64_t k, const uint64_t *inputs_mont, const uint64_t *proof_points,
                            const uint64_t *rand128, int *ok);
/* prod_i e(P_i, Q_i) == 1 ? -- the pairing-product test behind `PairingEngineExt::has_same` / `same_ratio`
 * (manta-crypto/src/arkworks/pairing.rs:88-109), which the trusted-setup verifier applies to random linear
 * combinations of whole queries (`verify_transform`, manta-trusted-setup/src/groth16/mpc.rs:470-508). P_i affine
 * G1, Q_i affine G2 (Montgomery limbs; an all-zero point is infinity and its pair contributes 1). n Miller loops
 * (one wavefront each, every Q_i prepared alongside by a second one) and one final exponentiation. */
int mg_pairing_check(mg_curve_t curve, const uint64_t *g1_affine, const uint64_t *g2_affine, size_t n, int *ok);
/* `Proof::deserialize` (arkworks compressed a | b | c) -> a | b | c affine Montgomery limbs; rejects non-canonical
 * encodings, points off the curve or outside the subgroup. */
int mg_proof_decode(mg_curve_t curve, const uint8_t *proof_bytes, uint64_t *points_out);

#ifdef __cplusplus
}
#endif
#endif
