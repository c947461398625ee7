// C ABI of libmantagpu.so (include/mantagpu.h): thin, exception-free wrappers over the engines.
#include "../../include/mantagpu.h"
#include "engine.h"
#include "prover.h"
#include <cstring>
#include <new>
#include <vector>

using namespace mg;

struct mg_bases {
    GroupEngine *eng;
    BaseSet *bs;
};
struct mg_msm_job {
    GroupEngine *eng;
    MsmWorkspace *ws;
};

#define MG_API extern "C" __attribute__((visibility("default")))
#define MG_TRY try {
#define MG_CATCH                                                                                                  \
    }                                                                                                             \
    catch (const std::bad_alloc &) { return MG_ERROR_OUT_OF_MEMORY; }                                             \
    catch (...) { return MG_ERROR_STATE; }

MG_API int mg_init(int device) {
    MG_TRY
    MG_HIP(hipSetDevice(device));
    MG_HIP(hipFree(nullptr));
    return MG_SUCCESS;
    MG_CATCH
}
MG_API const char *mg_strerror(int status) {
    switch (status) {
    case MG_SUCCESS: return "success";
    case MG_ERROR_INVALID_ARGUMENT: return "invalid argument";
    case MG_ERROR_HIP: return "HIP runtime error";
    case MG_ERROR_OUT_OF_MEMORY: return "out of memory";
    case MG_ERROR_DOMAIN_TOO_LARGE: return "evaluation domain exceeds the field's two-adicity";
    case MG_ERROR_STATE: return "invalid state";
    }
    return "unknown error";
}
MG_API const char *mg_last_error(void) { return last_error_string(); }
MG_API int mg_device_count(int *count) {
    if (!count) return MG_ERROR_INVALID_ARGUMENT;
    MG_HIP(hipGetDeviceCount(count));
    return MG_SUCCESS;
}
MG_API int mg_malloc(void **dptr, size_t bytes) {
    if (!dptr) return MG_ERROR_INVALID_ARGUMENT;
    MG_HIP(hipMalloc(dptr, bytes ? bytes : 1));
    return MG_SUCCESS;
}
MG_API int mg_free(void *dptr) {
    MG_HIP(hipFree(dptr));
    return MG_SUCCESS;
}
MG_API int mg_host_alloc(void **hptr, size_t bytes) {
    if (!hptr) return MG_ERROR_INVALID_ARGUMENT;
    MG_HIP(hipHostMalloc(hptr, bytes ? bytes : 1, hipHostMallocDefault));
    return MG_SUCCESS;
}
MG_API int mg_host_free(void *hptr) {
    MG_HIP(hipHostFree(hptr));
    return MG_SUCCESS;
}
MG_API int mg_memcpy_h2d(void *d, const void *h, size_t bytes) {
    MG_HIP(hipMemcpy(d, h, bytes, hipMemcpyHostToDevice));
    return MG_SUCCESS;
}
MG_API int mg_memcpy_d2h(void *h, const void *d, size_t bytes) {
    MG_HIP(hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost));
    return MG_SUCCESS;
}
MG_API int mg_device_synchronize(void) {
    MG_HIP(hipDeviceSynchronize());
    return MG_SUCCESS;
}

MG_API int mg_set_kernel_timing(int on) {
    set_kernel_timing(on != 0);
    return MG_SUCCESS;
}
MG_API float mg_last_accumulate_ms(void) { return last_accumulate_ms(); }

// ---------------------------------------------------------------------------------------------- MSM
MG_API int mg_bases_create(mg_curve_t curve, int group, const uint64_t *affine, size_t n, int on_device,
                           int precompute_window_bits, mg_bases **out) {
    MG_TRY
    if (!out || !affine || n == 0 || precompute_window_bits < 0 || precompute_window_bits > 24)
        return MG_ERROR_INVALID_ARGUMENT;
    GroupEngine *e = get_engine((int)curve, group);
    if (!e) return MG_ERROR_INVALID_ARGUMENT;
    BaseSet *bs = nullptr;
    int rc = e->bases_create((const u32 *)affine, n, on_device != 0, precompute_window_bits, &bs);
    if (rc) return rc;
    *out = new mg_bases{e, bs};
    return MG_SUCCESS;
    MG_CATCH
}
MG_API void mg_bases_destroy(mg_bases *b) {
    if (!b) return;
    b->eng->bases_destroy(b->bs);
    delete b;
}
MG_API size_t mg_bases_device_bytes(const mg_bases *b) { return b ? b->bs->bytes : 0; }

MG_API int mg_msm_launch(const mg_bases *b, const uint64_t *d_scalars, size_t n, int scalar_flags, int window_bits,
                         mg_msm_job **job) {
    MG_TRY
    if (!b || !d_scalars || !job || n == 0) return MG_ERROR_INVALID_ARGUMENT;
    MsmWorkspace *ws = b->eng->ws_acquire();
    if (!ws) return MG_ERROR_HIP;
    int rc = b->eng->msm_launch(b->bs, (const u32 *)d_scalars, n, (scalar_flags & MG_SCALARS_MONT) != 0, window_bits, ws, 1, 0,
                                (scalar_flags & MG_SCALARS_SPARSE) != 0);
    if (rc) {
        hipStreamSynchronize(ws->stream);
        b->eng->ws_release(ws);
        return rc;
    }
    *job = new mg_msm_job{b->eng, ws};
    return MG_SUCCESS;
    MG_CATCH
}
MG_API int mg_msm_finish(mg_msm_job *job, uint64_t *out_affine) {
    MG_TRY
    if (!job) return MG_ERROR_INVALID_ARGUMENT;
    HostPoint hp;
    int rc = job->eng->msm_finish(job->ws, &hp);
    if (!rc && out_affine) job->eng->hp_to_affine(&hp, (u32 *)out_affine);
    job->eng->ws_release(job->ws);
    delete job;
    return rc;
    MG_CATCH
}
MG_API int mg_msm(const mg_bases *b, const uint64_t *scalars, size_t n, uint64_t *out_affine) {
    MG_TRY
    if (!b || !scalars || !out_affine) return MG_ERROR_INVALID_ARGUMENT;
    if (n > b->bs->n) n = b->bs->n; // multi_scalar_mul zips to the shorter of the two
    if (n == 0) {
        std::memset(out_affine, 0, (size_t)b->eng->affine_words() * 4);
        return MG_SUCCESS;
    }
    void *d = nullptr;
    MG_HIP(hipMalloc(&d, n * 32));
    hipError_t e = hipMemcpy(d, scalars, n * 32, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        hipFree(d);
        set_last_hip_error(e, "hipMemcpy(scalars)", __FILE__, __LINE__);
        return MG_ERROR_HIP;
    }
    mg_msm_job *job = nullptr;
    int rc = mg_msm_launch(b, (const uint64_t *)d, n, 0, 0, &job);
    if (!rc) rc = mg_msm_finish(job, out_affine);
    hipFree(d);
    return rc;
    MG_CATCH
}
MG_API int mg_points_sum(mg_curve_t curve, int group, const uint64_t *affine, size_t n, uint64_t *out_affine) {
    MG_TRY
    GroupEngine *e = get_engine((int)curve, group);
    if (!e || !affine || !out_affine) return MG_ERROR_INVALID_ARGUMENT;
    HostPoint acc, t;
    e->hp_set_inf(&acc);
    const u32 *w = (const u32 *)affine;
    for (size_t i = 0; i < n; ++i) {
        e->hp_from_affine(&t, w + i * (size_t)e->affine_words());
        e->hp_add(&acc, &t);
    }
    e->hp_to_affine(&acc, (u32 *)out_affine);
    return MG_SUCCESS;
    MG_CATCH
}
MG_API int mg_fixed_base_mul(mg_curve_t curve, int group, const uint64_t *base_affine, const uint64_t *d_scalars,
                             size_t n, uint64_t *d_out_affine) {
    MG_TRY
    GroupEngine *e = get_engine((int)curve, group);
    if (!e || !base_affine || !d_scalars || !d_out_affine || n == 0) return MG_ERROR_INVALID_ARGUMENT;
    return e->fixed_base_mul((const u32 *)base_affine, (const u32 *)d_scalars, n, (u32 *)d_out_affine, nullptr);
    MG_CATCH
}
MG_API int mg_ec_elementwise(mg_curve_t curve, int group, int op, const uint64_t *a_affine, const uint64_t *b,
                             size_t n, uint64_t *out_affine) {
    MG_TRY
    GroupEngine *e = get_engine((int)curve, group);
    if (!e) return MG_ERROR_INVALID_ARGUMENT;
    return e->ec_elementwise(op, (const u32 *)a_affine, (const u32 *)b, n, (u32 *)out_affine);
    MG_CATCH
}
MG_API int mg_point_serialize(mg_curve_t curve, int group, const uint64_t *affine, int compressed, uint8_t *out) {
    MG_TRY
    GroupEngine *e = get_engine((int)curve, group);
    if (!e || !affine || !out) return MG_ERROR_INVALID_ARGUMENT;
    HostPoint p;
    e->hp_from_affine(&p, (const u32 *)affine);
    e->hp_serialize(&p, out, compressed != 0);
    return MG_SUCCESS;
    MG_CATCH
}

// ---------------------------------------------------------------------------------------------- NTT
MG_API int mg_ntt_device(mg_curve_t curve, uint64_t *d_data, unsigned log_n, int inverse, int coset) {
    MG_TRY
    if (!d_data) return MG_ERROR_INVALID_ARGUMENT;
    NttEngine *n = get_ntt_engine((int)curve);
    if (!n) return MG_ERROR_INVALID_ARGUMENT;
    int rc = n->transform((u32 *)d_data, log_n, inverse != 0, coset != 0, nullptr);
    if (rc) return rc;
    MG_HIP(hipStreamSynchronize(nullptr));
    return MG_SUCCESS;
    MG_CATCH
}
MG_API int mg_ntt(mg_curve_t curve, uint64_t *data, unsigned log_n, int inverse, int coset) {
    MG_TRY
    if (!data || log_n > 32) return MG_ERROR_INVALID_ARGUMENT;
    const size_t bytes = ((size_t)1 << log_n) * 32;
    void *d = nullptr;
    MG_HIP(hipMalloc(&d, bytes));
    int rc = MG_SUCCESS;
    if (hipMemcpy(d, data, bytes, hipMemcpyHostToDevice) != hipSuccess) rc = MG_ERROR_HIP;
    if (!rc) rc = mg_ntt_device(curve, (uint64_t *)d, log_n, inverse, coset);
    if (!rc && hipMemcpy(data, d, bytes, hipMemcpyDeviceToHost) != hipSuccess) rc = MG_ERROR_HIP;
    hipFree(d);
    return rc;
    MG_CATCH
}

// ---------------------------------------------------------------------------------------------- Groth16
struct mg_ctx {
    Prover *p;
};
MG_API int mg_ctx_create(mg_curve_t curve, const mg_pk_view *pk, mg_ctx **out) {
    MG_TRY
    if (!pk || !out) return MG_ERROR_INVALID_ARGUMENT;
    Prover *p = nullptr;
    int rc = prover_create((int)curve, pk, &p);
    if (rc) return rc;
    *out = new mg_ctx{p};
    return MG_SUCCESS;
    MG_CATCH
}
MG_API int mg_ctx_create_from_bytes(mg_curve_t curve, const uint8_t *bytes, size_t len, mg_ctx **out) {
    MG_TRY
    if (!bytes || !out) return MG_ERROR_INVALID_ARGUMENT;
    Prover *p = nullptr;
    int rc = prover_create_from_bytes((int)curve, bytes, len, &p);
    if (rc) return rc;
    *out = new mg_ctx{p};
    return MG_SUCCESS;
    MG_CATCH
}
MG_API int mg_ctx_set_r1cs(mg_ctx *ctx, const mg_csr *a, const mg_csr *b, const mg_csr *c, uint64_t m) {
    MG_TRY
    if (!ctx || !a || !b || !c) return MG_ERROR_INVALID_ARGUMENT;
    return ctx->p->set_r1cs(a, b, c, m);
    MG_CATCH
}
MG_API int mg_groth16_setup(mg_curve_t curve, const mg_csr *a, const mg_csr *b, const mg_csr *c, uint64_t m, uint64_t n_vars,
                            uint64_t n_inputs, const uint64_t *toxic, const uint64_t *g1_gen, const uint64_t *g2_gen,
                            const mg_pk_out *out) {
    MG_TRY
    return groth16_setup((int)curve, a, b, c, m, n_vars, n_inputs, toxic, g1_gen, g2_gen, out);
    MG_CATCH
}
MG_API int mg_groth16_prove(const mg_ctx *ctx, const uint64_t *z, const uint64_t r[4], const uint64_t s[4],
                            uint8_t *proof_out) {
    MG_TRY
    if (!ctx || !z || !r || !s || !proof_out) return MG_ERROR_INVALID_ARGUMENT;
    return ctx->p->prove(z, r, s, proof_out);
    MG_CATCH
}
MG_API int mg_groth16_prove_batch(const mg_ctx *ctx, uint64_t k, const uint64_t *z, const uint64_t *r, const uint64_t *s,
                                  uint8_t *proofs_out) {
    MG_TRY
    if (!ctx || !z || !r || !s || !proofs_out || k == 0) return MG_ERROR_INVALID_ARGUMENT;
    return ctx->p->prove_batch(k, z, r, s, proofs_out);
    MG_CATCH
}
MG_API int mg_witness_map(const mg_ctx *ctx, const uint64_t *z, uint64_t *h_out) {
    MG_TRY
    if (!ctx || !z || !h_out) return MG_ERROR_INVALID_ARGUMENT;
    return ctx->p->witness_map_host(z, h_out);
    MG_CATCH
}
MG_API uint64_t mg_ctx_domain_size(const mg_ctx *ctx) { return ctx ? ctx->p->domain_size() : 0; }
MG_API void mg_ctx_destroy(mg_ctx *ctx) {
    if (!ctx) return;
    delete ctx->p;
    delete ctx;
}
