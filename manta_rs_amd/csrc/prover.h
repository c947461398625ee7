// Scalar-field (Fr) device engine -- NTT, sparse matrix-vector product, QAP pointwise kernel -- and the
// Groth16 prover that strings them together with the MSM engines.
#pragma once
#include "tuning.h"
#include "../../include/mantagpu.h"
#include "engine.h"
#include <vector>

namespace mg {

struct DevCsr {
    u32 *row_ptr = nullptr, *col = nullptr, *val = nullptr;
    u64 nnz = 0;
};

class FrEngine {
  public:
    virtual ~FrEngine() {}
    virtual int two_adicity() const = 0;
    // In-place radix-2 NTT of 2^log_n Montgomery elements in HBM, natural order in/out
    // (ark-poly Radix2EvaluationDomain::{fft,ifft,coset_fft,coset_ifft}_in_place semantics).
    virtual int transform(u32 *d_data, unsigned log_n, bool inverse, bool coset, hipStream_t s) = 0;
    // Inside the witness-map pipeline an Fr element is held in the reduced-radix form of fpr_dev.h: work_words() u32 per
    // element (9). The three products A z, B z, C z of the witness map in one launch, written in that form; the A
    // vector also receives the input-consistency rows a[m + j] = z_j, j < P
    // (batch members: z vectors z_stride u32 apart, outputs out_stride u32 apart)
    virtual int work_words() const = 0;
    virtual int spmv3(const DevCsr &A, const DevCsr &B, const DevCsr &C, const u32 *d_z, u32 *d_a, u32 *d_b, u32 *d_c, u64 m, u64 P,
                      hipStream_t s, u32 batch = 1, size_t z_stride = 0, size_t out_stride = 0, u64 rows_total = 0) = 0;
    // a, b, c = constraint evaluations over the domain (work form) -> a = coefficients of h = (AB - C)/Z in
    // bit-reversed order, work form, < 2p (ifft, coset fft x3, pointwise, coset ifft; fused, permutation-free)
    // batch > 1: a, b, c each hold `batch` vectors back to back
    virtual int qap_quotient(u32 *d_a, u32 *d_b, u32 *d_c, unsigned log_n, hipStream_t s, u32 batch = 1) = 0;
    // work form -> arkworks format
    virtual int work_to_std(const u32 *d_work, size_t n, u32 *d_std, hipStream_t s) = 0;
    // the domain's device twiddle table (omega^k or omega^-k, k < n/2, Montgomery) and n^-1 as a canonical integer
    virtual int domain_twiddles(unsigned log_n, bool inverse, const u32 **d_tw, u64 n_inv_canonical[4]) = 0;
    // host-side Fr helpers (Montgomery in/out unless noted)
    virtual void fr_mul(const u64 a[4], const u64 b[4], u64 out[4]) const = 0;
    virtual void fr_to_canonical(const u64 a[4], u64 out[4]) const = 0;
    // Scalars of the Groth16 key for QAP(A, B, C) at the toxic waste (alpha, beta, gamma, delta, tau; Montgomery):
    //   a_j = sum_i A[i][j] L_i(tau) (+ L_{m+j}(tau) for j < P), b_j, c_j likewise over the domain of size
    //   2^log_d >= m + P;   gamma_abc_j = (beta a_j + alpha b_j + c_j)/gamma (j < P);   l_j the same over delta
    //   (j >= P);   h_i = tau^i (tau^D - 1)/delta, i < D - 1.
    // Outputs are CANONICAL 4 x u64 integers, ready for the fixed-base multiplication kernels:
    //   s1 = alpha | beta | delta | gamma_abc[P] | a[V] | b[V] | h[D-1] | l[V-P]      s2 = beta | gamma | delta | b[V]
    virtual int setup_scalars(const mg_csr *a, const mg_csr *b, const mg_csr *c, u64 m, u64 V, u64 P, unsigned log_d,
                              const u64 *toxic5, std::vector<u64> &s1, std::vector<u64> &s2) const = 0;
};
typedef FrEngine NttEngine;
FrEngine *make_fr_engine_bn254();
FrEngine *make_fr_engine_bls381();
FrEngine *get_ntt_engine(int curve);

class Prover {
  public:
    virtual ~Prover() {}
    virtual int set_r1cs(const mg_csr *a, const mg_csr *b, const mg_csr *c, u64 m) = 0;
    virtual int prove(const uint64_t *z, const uint64_t *r, const uint64_t *s, uint8_t *proof_out) = 0;
    // k proofs of the same circuit in one pass of the GPU pipeline: z = k assignments back to back, r and s = k
    // field elements each, proofs_out = k encoded proofs back to back
    virtual int prove_batch(u64 k, const uint64_t *z, const uint64_t *r, const uint64_t *s, uint8_t *proofs_out) = 0;
    virtual int witness_map_host(const uint64_t *z, uint64_t *h_out) = 0;
    virtual u64 domain_size() const = 0;
    virtual void table_bytes(u64 out2[2]) const = 0; // bucket tables, full tables (all shards)
    virtual u64 n_vars() const = 0;
    virtual u64 n_inputs() const = 0;
    virtual u32 n_shards() const = 0;
    // process-per-GPU sharding (prover_create_shard): see prover.cpp
    virtual int partials_launch(u64 k, const uint64_t *z, uint64_t *d_out, void *consumer_stream, void **job) = 0;
    virtual int partials_finish(void *job) = 0;
    virtual int assemble(u64 k, u32 n_parts, const uint64_t *parts, const uint64_t *r, const uint64_t *s, uint8_t *proofs_out) = 0;
    virtual size_t slot_words() const = 0;
};
// how a context is placed and what it may spend (mg_ctx_opts of the C ABI)
struct ProverOptions {
    const int *devices = nullptr; // in-process range sharding over these devices (nullptr: the current device)
    int n_devices = 0;
    u32 shard = 0, n_shards = 1;  // process-per-GPU range sharding: this process holds shard `shard` of `n_shards`
    u32 task_mask = 0x1f;         // task placement: the MSMs (bit 0 a, 1 b_g1, 2 b_g2, 3 l, 4 h) this context computes
    int64_t full_table_bytes = -1; // HBM budget of the full tables; < 0: the default (a tenth of the device's HBM)
    int exchange = 0;             // in-process sharding: 0 = partial points summed through pinned host memory, 1 = RCCL all_gather
    bool partials_interface = false; // the context will be driven through partials_launch / assemble (mg_ctx_create_shard)
    const Tuning *tuning = nullptr;  // per-context tuning (validated by the caller); nullptr: the process-wide values
};
int prover_create_ex(int curve, const mg_pk_view *pk, const ProverOptions &o, Prover **out);
int prover_create(int curve, const mg_pk_view *pk, Prover **out);
// every MSM of a proof range-sharded over the listed devices; devices may repeat
int prover_create_sharded(int curve, const mg_pk_view *pk, const int *devices, int n_devices, Prover **out);
int prover_create_shard(int curve, const mg_pk_view *pk, u32 shard, u32 n_shards, Prover **out);
int prover_create_task(int curve, const mg_pk_view *pk, u32 task_mask, Prover **out);
// arkworks `ProvingKey::serialize_unchecked` bytes (ProvingContext::decode, groth16.rs:268-288)
int prover_create_from_bytes(int curve, const uint8_t *bytes, size_t len, Prover **out, const int *devices = nullptr,
                             int n_devices = 0);
int prover_create_from_bytes_ex(int curve, const uint8_t *bytes, size_t len, const ProverOptions &o, Prover **out);
// Groth16 key generation from explicit toxic waste and group generators (setup.cpp)
int groth16_setup(int curve, const mg_csr *a, const mg_csr *b, const mg_csr *c, u64 m, u64 n_vars, u64 n_inputs,
                  const u64 *toxic5, const u64 *g1_gen, const u64 *g2_gen, const mg_pk_out *out);

} // namespace mg
