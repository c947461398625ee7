// C ABI of libmantagpu.so (include/mantagpu.h): thin, exception-free wrappers over the engines.
#include "../../include/mantagpu.h"
#include "engine.h"
#include "tuning.h"
#include <cstddef>
#include "prover.h"
#include "verify.h"
#include <cstring>
#include <new>
#include <vector>

using namespace mg;

// A registered base vector: one shard per device of the list it was created over (a plain mg_bases_create gives one
// shard on the current device). Shard g owns the contiguous range [lo, hi) of the points.
struct BasesShard {
    GroupEngine *eng;
    BaseSet *bs;
    int device;
    size_t lo, hi;
};
struct mg_bases {
    std::vector<BasesShard> sh;
    size_t n = 0;
};
struct JobShard {
    GroupEngine *eng;
    MsmWorkspace *ws;
    int device;
    void *d_tmp; // scalars uploaded by the host-to-host convenience call (freed at finish)
};
struct mg_msm_job {
    std::vector<JobShard> sh;
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
    case MG_ERROR_CHECKSUM: return "BLAKE3 checksum mismatch";
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
    MG_HIP(memcpy_sync(d, h, bytes, hipMemcpyHostToDevice)); // (the calling thread's setup stream, not the NULL stream: engine.h)
    return MG_SUCCESS;
}
MG_API int mg_memcpy_d2h(void *h, const void *d, size_t bytes) {
    MG_HIP(memcpy_sync(h, d, bytes, hipMemcpyDeviceToHost));
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
MG_API float mg_last_accumulate_mhz(void) { return last_accumulate_mhz(); }
namespace mg {
int clock_probe(u32 iters, double *memtime_mhz, double *mad_issue_per_us_per_simd, double *ms);
}
MG_API int mg_clock_probe(unsigned iters, double *memtime_mhz, double *mad_issue_per_us_per_simd, double *ms) {
    MG_TRY
    if (iters == 0 || iters > (1u << 24)) return MG_ERROR_INVALID_ARGUMENT;
    return clock_probe(iters, memtime_mhz, mad_issue_per_us_per_simd, ms);
    MG_CATCH
}
MG_API int mg_last_ntt_ms(float out4[4]) {
    if (!out4) return MG_ERROR_INVALID_ARGUMENT;
    get_last_ntt_ms(out4);
    return MG_SUCCESS;
}
MG_API int mg_hw_queues(int out2[2]) {
    if (!out2) return MG_ERROR_INVALID_ARGUMENT;
    out2[0] = out2[1] = 0;
    (void)stream_queue_counts(&out2[0], &out2[1]);
    return MG_SUCCESS;
}
MG_API int mg_last_pass_host_ms(float out3[3]) {
    if (!out3) return MG_ERROR_INVALID_ARGUMENT;
    get_last_pass_host_ms(out3);
    return MG_SUCCESS;
}
MG_API int mg_last_prove_phases_ms(float out10[10]) {
    if (!out10) return MG_ERROR_INVALID_ARGUMENT;
    get_last_prove_ms(out10);
    return MG_SUCCESS;
}

// ---------------------------------------------------------------------------------------------- MSM
static void bases_free(mg_bases *b) {
    DeviceGuard guard;
    for (BasesShard &s : b->sh) {
        hipSetDevice(s.device);
        if (s.bs) s.eng->bases_destroy(s.bs);
    }
    delete b;
}
static int bases_create_on(mg_curve_t curve, int group, const uint64_t *affine, size_t n, int on_device, int pre_c,
                           const int *devices, int n_devices, mg_bases **out) {
    if (!out || !affine || n == 0 || pre_c < -12 || pre_c == -1 || pre_c > 24 || n_devices < 1 || n_devices > 64 || (size_t)n_devices > n)
        return MG_ERROR_INVALID_ARGUMENT;
    if (on_device && n_devices != 1) return MG_ERROR_INVALID_ARGUMENT; // a device pointer belongs to one device
    int count = 0;
    MG_HIP(hipGetDeviceCount(&count));
    DeviceGuard guard;
    mg_bases *b = new mg_bases();
    b->n = n;
    for (int g = 0; g < n_devices; ++g) {
        const int dev = devices ? devices[g] : guard.prev;
        if (dev < 0 || dev >= count || dev >= MAX_DEVICES) {
            bases_free(b);
            return MG_ERROR_INVALID_ARGUMENT;
        }
        hipError_t he = hipSetDevice(dev);
        GroupEngine *e = he == hipSuccess ? get_engine((int)curve, group) : nullptr;
        if (!e) {
            bases_free(b);
            return he == hipSuccess ? MG_ERROR_INVALID_ARGUMENT : MG_ERROR_HIP;
        }
        const size_t lo = n * (size_t)g / n_devices, hi = n * (size_t)(g + 1) / n_devices;
        BaseSet *bs = nullptr;
        int rc = e->bases_create((const u32 *)affine + lo * (size_t)e->affine_words(), hi - lo, on_device != 0, pre_c, &bs);
        if (rc) {
            bases_free(b);
            return rc;
        }
        b->sh.push_back(BasesShard{e, bs, dev, lo, hi});
    }
    *out = b;
    return MG_SUCCESS;
}
MG_API int mg_bases_create(mg_curve_t curve, int group, const uint64_t *affine, size_t n, int on_device,
                           int precompute_window_bits, mg_bases **out) {
    MG_TRY
    HeavyOp no_capture_meanwhile;
    return bases_create_on(curve, group, affine, n, on_device, precompute_window_bits, nullptr, 1, out);
    MG_CATCH
}
MG_API int mg_bases_create_sharded(mg_curve_t curve, int group, const uint64_t *affine, size_t n, const int *devices,
                                   int n_devices, int precompute_window_bits, mg_bases **out) {
    MG_TRY
    HeavyOp no_capture_meanwhile;
    if (!devices) return MG_ERROR_INVALID_ARGUMENT;
    return bases_create_on(curve, group, affine, n, 0, precompute_window_bits, devices, n_devices, out);
    MG_CATCH
}
MG_API void mg_bases_destroy(mg_bases *b) {
    HeavyOp no_capture_meanwhile;
    if (b) bases_free(b);
}
MG_API size_t mg_bases_device_bytes(const mg_bases *b) {
    size_t t = 0;
    if (b)
        for (const BasesShard &s : b->sh) t += s.bs->bytes;
    return t;
}
MG_API int mg_bases_num_shards(const mg_bases *b) { return b ? (int)b->sh.size() : 0; }
MG_API int mg_bases_shard(const mg_bases *b, int shard, int *device, size_t *lo, size_t *hi) {
    if (!b || shard < 0 || (size_t)shard >= b->sh.size()) return MG_ERROR_INVALID_ARGUMENT;
    if (device) *device = b->sh[(size_t)shard].device;
    if (lo) *lo = b->sh[(size_t)shard].lo;
    if (hi) *hi = b->sh[(size_t)shard].hi;
    return MG_SUCCESS;
}

static void job_abandon(mg_msm_job *job) {
    DeviceGuard guard;
    for (JobShard &j : job->sh) {
        hipSetDevice(j.device);
        hipStreamSynchronize(msm_stream_of(j.ws));
        if (j.ws->side_stream) hipStreamSynchronize(j.ws->side_stream);
        j.ws->pending = 0;
        j.eng->ws_release(j.ws);
        if (j.d_tmp) hipFree(j.d_tmp);
    }
    delete job;
}
// one launch per shard; d_scalars[g] = the scalars of shard g's range, resident on shard g's device (n_g of them,
// n_g <= hi - lo); host_scalars != nullptr instead uploads the slices first
static int launch_shards(const mg_bases *b, const uint64_t *const *d_scalars, const uint64_t *host_scalars, size_t n_total,
                         int scalar_flags, int window_bits, mg_msm_job **out) {
    DeviceGuard guard;
    mg_msm_job *job = new mg_msm_job();
    for (size_t g = 0; g < b->sh.size(); ++g) {
        const BasesShard &s = b->sh[g];
        if (n_total <= s.lo) break; // multi_scalar_mul zips to the shorter side: later shards have no scalars
        const size_t n = (n_total < s.hi ? n_total : s.hi) - s.lo;
        void *d_tmp = nullptr;
        const u32 *d_sc = d_scalars ? (const u32 *)d_scalars[g] : nullptr;
        int rc = MG_SUCCESS;
        hipError_t he = hipSetDevice(s.device);
        if (he == hipSuccess && host_scalars) {
            he = hipMalloc(&d_tmp, n * 32);
            if (he == hipSuccess) he = memcpy_sync(d_tmp, host_scalars + s.lo * 4, n * 32, hipMemcpyHostToDevice);
            d_sc = (const u32 *)d_tmp;
        }
        if (he != hipSuccess) {
            set_last_hip_error(he, "sharded MSM launch", __FILE__, __LINE__);
            rc = he == hipErrorOutOfMemory ? MG_ERROR_OUT_OF_MEMORY : MG_ERROR_HIP;
        } else if (!d_sc) {
            rc = MG_ERROR_INVALID_ARGUMENT;
        }
        MsmWorkspace *ws = rc ? nullptr : s.eng->ws_acquire();
        if (!rc && !ws) rc = MG_ERROR_HIP;
        if (ws) {
            // a stand-alone MSM: on a stream with a hardware queue of its own (engine.h MsmWorkspace::solo)
            // (tuning msm_dedicated_queues: 1 = only while no proving / verifying context is alive in the process, 2 = always)
            const int dq = tuning().msm_dedicated_queues;
            const bool want = dq == 2 || (dq == 1 && graph_clients_alive() == 0);
            if (want && !ws->solo) ws->solo = stream_pool_get_dedicated();
            ws->use_solo = want && ws->solo != nullptr;
            job->sh.push_back(JobShard{s.eng, ws, s.device, d_tmp});
            rc = s.eng->msm_launch(s.bs, d_sc, n, (scalar_flags & MG_SCALARS_MONT) ? SCALARS_MONT : SCALARS_CANONICAL, window_bits, ws, 1, 0,
                                   (scalar_flags & MG_SCALARS_SPARSE) != 0);
        } else if (d_tmp) {
            hipFree(d_tmp);
        }
        if (rc) {
            job_abandon(job);
            return rc;
        }
    }
    *out = job;
    return MG_SUCCESS;
}
MG_API int mg_msm_launch(const mg_bases *b, const uint64_t *d_scalars, size_t n, int scalar_flags, int window_bits,
                         mg_msm_job **job) {
    MG_TRY
    HeavyOp may_allocate_no_capture_meanwhile;
    if (!b || !d_scalars || !job || n == 0 || b->sh.size() != 1) return MG_ERROR_INVALID_ARGUMENT;
    if (n > b->n) return MG_ERROR_INVALID_ARGUMENT;
    const uint64_t *one[1] = {d_scalars};
    return launch_shards(b, one, nullptr, n, scalar_flags, window_bits, job);
    MG_CATCH
}
MG_API int mg_msm_launch_sharded(const mg_bases *b, const uint64_t *const *d_scalars_per_shard, int scalar_flags,
                                 int window_bits, mg_msm_job **job) {
    MG_TRY
    HeavyOp may_allocate_no_capture_meanwhile;
    if (!b || !d_scalars_per_shard || !job) return MG_ERROR_INVALID_ARGUMENT;
    for (size_t g = 0; g < b->sh.size(); ++g)
        if (!d_scalars_per_shard[g]) return MG_ERROR_INVALID_ARGUMENT;
    return launch_shards(b, d_scalars_per_shard, nullptr, b->n, scalar_flags, window_bits, job);
    MG_CATCH
}
// waits for every shard, folds its staged points, and adds the partial results: the exchange step of the
// sharded MSM (the per-device partial points arrive through pinned host staging and are summed here)
MG_API int mg_msm_finish(mg_msm_job *job, uint64_t *out_affine) {
    MG_TRY
    if (!job) return MG_ERROR_INVALID_ARGUMENT;
    DeviceGuard guard;
    int rc = MG_SUCCESS;
    HostPoint total, hp;
    GroupEngine *e0 = job->sh.empty() ? nullptr : job->sh[0].eng;
    if (e0) e0->hp_set_inf(&total);
    for (JobShard &j : job->sh) {
        hipSetDevice(j.device);
        int rc2 = j.eng->msm_finish(j.ws, &hp);
        if (rc2) {
            hipStreamSynchronize(msm_stream_of(j.ws));
            if (j.ws->side_stream) hipStreamSynchronize(j.ws->side_stream);
            j.ws->pending = 0;
        } else {
            e0->hp_add(&total, &hp);
        }
        if (!rc) rc = rc2;
        j.eng->ws_release(j.ws);
        if (j.d_tmp) hipFree(j.d_tmp);
    }
    if (!rc && out_affine && e0) e0->hp_to_affine(&total, (u32 *)out_affine);
    delete job;
    return rc;
    MG_CATCH
}
MG_API int mg_msm_result_to_device(mg_msm_job *job, uint64_t *d_out_xyzz, void *stream) {
    MG_TRY
    if (!job || job->sh.size() != 1 || !d_out_xyzz) return MG_ERROR_INVALID_ARGUMENT;
    JobShard &j = job->sh[0];
    DeviceGuard guard;
    MG_HIP(hipSetDevice(j.device));
    int rc = j.eng->msm_fold_device(j.ws, (u32 *)d_out_xyzz, (size_t)j.eng->xyzz_words());
    if (rc) return rc;
    hipStream_t ms = msm_stream_of(j.ws);
    MG_HIP(hipEventRecord(j.ws->done, ms)); // mg_msm_finish waits for this event: now it covers the fold as well
    if ((hipStream_t)stream != ms) MG_HIP(hipStreamWaitEvent((hipStream_t)stream, j.ws->done, 0)); // NULL = the default stream
    return MG_SUCCESS;
    MG_CATCH
}
MG_API size_t mg_xyzz_limbs(mg_curve_t curve, int group) {
    GroupEngine *e = get_engine((int)curve, group);
    return e ? (size_t)e->xyzz_words() / 2 : 0;
}
MG_API int mg_xyzz_sum(mg_curve_t curve, int group, const uint64_t *xyzz, size_t n, uint64_t *out_affine) {
    MG_TRY
    GroupEngine *e = get_engine((int)curve, group);
    if (!e || !xyzz || !out_affine) return MG_ERROR_INVALID_ARGUMENT;
    HostPoint acc, t;
    e->hp_set_inf(&acc);
    for (size_t i = 0; i < n; ++i) {
        e->hp_from_xyzz(&t, (const u32 *)xyzz + i * (size_t)e->xyzz_words());
        e->hp_add(&acc, &t);
    }
    e->hp_to_affine(&acc, (u32 *)out_affine);
    return MG_SUCCESS;
    MG_CATCH
}
MG_API int mg_msm(const mg_bases *b, const uint64_t *scalars, size_t n, uint64_t *out_affine) {
    MG_TRY
    if (!b || !scalars || !out_affine || b->sh.empty()) return MG_ERROR_INVALID_ARGUMENT;
    if (n > b->n) n = b->n; // multi_scalar_mul zips to the shorter of the two
    if (n == 0) {
        std::memset(out_affine, 0, (size_t)b->sh[0].eng->affine_words() * 4);
        return MG_SUCCESS;
    }
    mg_msm_job *job = nullptr;
    int rc = launch_shards(b, nullptr, scalars, n, 0, 0, &job);
    if (!rc) rc = mg_msm_finish(job, out_affine);
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
// Radix2EvaluationDomain::{fft, ifft} over group elements (manta-trusted-setup/src/groth16/mpc.rs:378-381)
MG_API int mg_group_ntt(mg_curve_t curve, int group, const uint64_t *points_affine, unsigned log_n, int inverse,
                        uint64_t *out_affine) {
    MG_TRY
    GroupEngine *e = get_engine((int)curve, group);
    FrEngine *fr = get_ntt_engine((int)curve);
    if (!e || !fr || !points_affine || !out_affine) return MG_ERROR_INVALID_ARGUMENT;
    const u32 *tw = nullptr;
    u64 ninv[4];
    int rc = fr->domain_twiddles(log_n, inverse != 0, &tw, ninv);
    if (rc) return rc;
    return e->group_ntt((const u32 *)points_affine, log_n, tw, inverse ? (const u32 *)ninv : nullptr, (u32 *)out_affine);
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

namespace mg {
int field_op(int field, int op, int repr, int lazy_a, int lazy_b, const u32 *a, const u32 *b, size_t n, u32 *out);
}
MG_API int mg_field_op(int field, int op, int repr, int lazy_a, int lazy_b, const uint64_t *a, const uint64_t *b, size_t n,
                       uint64_t *out) {
    MG_TRY
    return field_op(field, op, repr, lazy_a, lazy_b, (const u32 *)a, (const u32 *)b, n, (u32 *)out);
    MG_CATCH
}

// ---------------------------------------------------------------------------------------------- NTT
MG_API int mg_ntt_device(mg_curve_t curve, uint64_t *d_data, unsigned log_n, int inverse, int coset) {
    MG_TRY
    if (!d_data) return MG_ERROR_INVALID_ARGUMENT;
    NttEngine *n = get_ntt_engine((int)curve);
    if (!n) return MG_ERROR_INVALID_ARGUMENT;
    int rc = n->transform((u32 *)d_data, log_n, inverse != 0, coset != 0, setup_stream());
    if (rc) return rc;
    MG_HIP(setup_sync());
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
    if (memcpy_sync(d, data, bytes, hipMemcpyHostToDevice) != hipSuccess) rc = MG_ERROR_HIP;
    if (!rc) rc = mg_ntt_device(curve, (uint64_t *)d, log_n, inverse, coset);
    if (!rc && memcpy_sync(data, d, bytes, hipMemcpyDeviceToHost) != hipSuccess) rc = MG_ERROR_HIP;
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
MG_API int mg_ctx_create_sharded(mg_curve_t curve, const mg_pk_view *pk, const int *devices, int n_devices, mg_ctx **out) {
    MG_TRY
    if (!pk || !out || !devices) return MG_ERROR_INVALID_ARGUMENT;
    Prover *p = nullptr;
    int rc = prover_create_sharded((int)curve, pk, devices, n_devices, &p);
    if (rc) return rc;
    *out = new mg_ctx{p};
    return MG_SUCCESS;
    MG_CATCH
}
MG_API int mg_ctx_create_shard(mg_curve_t curve, const mg_pk_view *pk, int shard, int n_shards, mg_ctx **out) {
    MG_TRY
    if (!out || shard < 0 || n_shards < 1) return MG_ERROR_INVALID_ARGUMENT;
    Prover *p = nullptr;
    int rc = prover_create_shard((int)curve, pk, (u32)shard, (u32)n_shards, &p);
    if (rc) return rc;
    *out = new mg_ctx{p};
    return MG_SUCCESS;
    MG_CATCH
}
MG_API int mg_ctx_create_task(mg_curve_t curve, const mg_pk_view *pk, unsigned task_mask, mg_ctx **out) {
    MG_TRY
    if (!out) return MG_ERROR_INVALID_ARGUMENT;
    Prover *p = nullptr;
    int rc = prover_create_task((int)curve, pk, (u32)task_mask, &p);
    if (rc) return rc;
    *out = new mg_ctx{p};
    return MG_SUCCESS;
    MG_CATCH
}
// mg_ctx_opts -> ProverOptions; a struct shorter than ours (an older caller) is read up to its own size
static_assert(sizeof(mg_tuning) == sizeof(Tuning) && offsetof(mg_tuning, full_table_bytes) == offsetof(Tuning, full_table_bytes) &&
                  offsetof(mg_tuning, window_bits_g2) == offsetof(Tuning, window_bits_g2),
              "mg_tuning (mantagpu.h) and mg::Tuning (tuning.h) are the same struct");
// a caller's mg_tuning -> Tuning: a shorter struct (an older caller) is read up to its own size over the process-wide values
static int tuning_from_abi(const mg_tuning *in, Tuning &t) {
    if (!in || in->struct_size < 8 || in->struct_size > 4096) return MG_ERROR_INVALID_ARGUMENT;
    t = tuning();
    std::memcpy(&t, in, std::min<size_t>(in->struct_size, sizeof(t)));
    return normalize_tuning(t) == MG_OK ? MG_SUCCESS : MG_ERROR_INVALID_ARGUMENT;
}
MG_API int mg_tuning_init(mg_tuning *t) {
    if (!t) return MG_ERROR_INVALID_ARGUMENT;
    const Tuning d = tuning_defaults();
    std::memcpy(t, &d, sizeof(d));
    return MG_SUCCESS;
}
MG_API int mg_get_tuning(mg_tuning *t) {
    MG_TRY
    if (!t) return MG_ERROR_INVALID_ARGUMENT;
    const Tuning d = tuning();
    std::memcpy(t, &d, sizeof(d));
    return MG_SUCCESS;
    MG_CATCH
}
MG_API int mg_set_tuning(const mg_tuning *t) {
    MG_TRY
    Tuning v;
    const int rc = tuning_from_abi(t, v);
    if (rc) return rc;
    return set_tuning(v) == MG_OK ? MG_SUCCESS : MG_ERROR_INVALID_ARGUMENT;
    MG_CATCH
}
MG_API const char *const *mg_tuning_env_names(void) { return tuning_env_names(); }
// (the Tuning a context's options point at lives in the caller's frame until prover_create_ex returns)
static int opts_from_abi(const mg_ctx_opts *in, ProverOptions &o, Tuning &tn) {
    if (!in) return MG_SUCCESS;
    mg_ctx_opts t;
    mg_ctx_opts_init(&t);
    if (in->struct_size < 8 || in->struct_size > 4096) return MG_ERROR_INVALID_ARGUMENT; // not initialised by mg_ctx_opts_init
    std::memcpy(&t, in, std::min<size_t>(in->struct_size, sizeof(t)));
    if (t.n_devices < 0 || (t.n_devices > 0 && !t.devices) || t.shard < 0 || t.n_shards < 0) return MG_ERROR_INVALID_ARGUMENT;
    o.devices = t.n_devices > 0 ? t.devices : nullptr;
    o.n_devices = t.n_devices;
    o.shard = (u32)t.shard;
    o.n_shards = t.n_shards > 0 ? (u32)t.n_shards : 1;
    o.task_mask = t.task_mask ? t.task_mask : 0x1f;
    o.full_table_bytes = t.full_table_bytes;
    o.exchange = (int)t.exchange;
    if (t.tuning) {
        const int rc = tuning_from_abi(t.tuning, tn);
        if (rc) return rc;
        o.tuning = &tn;
    }
    return MG_SUCCESS;
}
MG_API int mg_ctx_opts_init(mg_ctx_opts *o) {
    if (!o) return MG_ERROR_INVALID_ARGUMENT;
    std::memset(o, 0, sizeof(*o));
    o->struct_size = (uint32_t)sizeof(*o);
    o->exchange = MG_EXCHANGE_HOST;
    o->full_table_bytes = -1;
    o->n_shards = 1;
    o->task_mask = 0x1f;
    return MG_SUCCESS;
}
MG_API int mg_ctx_create_ex(mg_curve_t curve, const mg_pk_view *pk, const mg_ctx_opts *opts, mg_ctx **out) {
    MG_TRY
    if (!pk || !out) return MG_ERROR_INVALID_ARGUMENT;
    ProverOptions o;
    Tuning tn;
    int rc = opts_from_abi(opts, o, tn);
    if (rc) return rc;
    Prover *p = nullptr;
    rc = prover_create_ex((int)curve, pk, o, &p);
    if (rc) return rc;
    *out = new mg_ctx{p};
    return MG_SUCCESS;
    MG_CATCH
}
namespace mg {
void blake3_hash(const uint8_t *data, size_t len, uint8_t out[32]);
}
MG_API int mg_ctx_create_from_bytes_ex(mg_curve_t curve, const uint8_t *bytes, size_t len, const uint8_t *checksum32,
                                       const mg_ctx_opts *opts, mg_ctx **out) {
    MG_TRY
    HeavyOp no_capture_meanwhile;
    if (!bytes || !out) return MG_ERROR_INVALID_ARGUMENT;
    ProverOptions o;
    Tuning tn;
    int rc = opts_from_abi(opts, o, tn);
    if (rc) return rc;
    if (checksum32) { // manta_parameters::verify in front of the loader, whatever the placement
        uint8_t h[32];
        blake3_hash(bytes, len, h);
        if (std::memcmp(h, checksum32, 32) != 0) return MG_ERROR_CHECKSUM;
    }
    Prover *p = nullptr;
    rc = prover_create_from_bytes_ex((int)curve, bytes, len, o, &p);
    if (rc) return rc;
    *out = new mg_ctx{p};
    return MG_SUCCESS;
    MG_CATCH
}
struct mg_partials_job {
    Prover *p;
    void *job;
};
MG_API size_t mg_partials_slot_limbs(const mg_ctx *ctx) { return ctx ? ctx->p->slot_words() / 2 : 0; }
MG_API int mg_groth16_partials_launch(const mg_ctx *ctx, uint64_t k, const uint64_t *z, uint64_t *d_out, void *stream,
                                      mg_partials_job **job) {
    MG_TRY
    if (!ctx || !job) return MG_ERROR_INVALID_ARGUMENT;
    void *j = nullptr;
    int rc = ctx->p->partials_launch(k, z, d_out, stream, &j);
    if (rc) return rc;
    *job = new mg_partials_job{ctx->p, j};
    return MG_SUCCESS;
    MG_CATCH
}
MG_API int mg_groth16_partials_finish(mg_partials_job *job) {
    MG_TRY
    if (!job) return MG_ERROR_INVALID_ARGUMENT;
    int rc = job->p->partials_finish(job->job);
    delete job;
    return rc;
    MG_CATCH
}
MG_API int mg_groth16_assemble(const mg_ctx *ctx, uint64_t k, int n_parts, const uint64_t *parts, const uint64_t *r,
                               const uint64_t *s, uint8_t *proofs_out) {
    MG_TRY
    if (!ctx || n_parts < 1) return MG_ERROR_INVALID_ARGUMENT;
    return ctx->p->assemble(k, (u32)n_parts, parts, r, s, proofs_out);
    MG_CATCH
}
MG_API int mg_ctx_create_from_bytes_sharded(mg_curve_t curve, const uint8_t *bytes, size_t len, const int *devices,
                                            int n_devices, mg_ctx **out) {
    MG_TRY
    HeavyOp no_capture_meanwhile;
    if (!bytes || !out || !devices || n_devices < 1) return MG_ERROR_INVALID_ARGUMENT;
    Prover *p = nullptr;
    int rc = prover_create_from_bytes((int)curve, bytes, len, &p, devices, n_devices);
    if (rc) return rc;
    *out = new mg_ctx{p};
    return MG_SUCCESS;
    MG_CATCH
}
MG_API int mg_blake3(const uint8_t *data, size_t len, uint8_t out32[32]) {
    MG_TRY
    if ((!data && len) || !out32) return MG_ERROR_INVALID_ARGUMENT;
    static const uint8_t none = 0;
    blake3_hash(data ? data : &none, len, out32);
    return MG_SUCCESS;
    MG_CATCH
}
MG_API int mg_ctx_create_from_bytes(mg_curve_t curve, const uint8_t *bytes, size_t len, mg_ctx **out);
MG_API int mg_ctx_create_from_bytes_checked(mg_curve_t curve, const uint8_t *bytes, size_t len, const uint8_t checksum[32],
                                            mg_ctx **out) {
    MG_TRY
    if (!bytes || !checksum || !out) return MG_ERROR_INVALID_ARGUMENT;
    uint8_t h[32];
    blake3_hash(bytes, len, h);
    if (std::memcmp(h, checksum, 32) != 0) return MG_ERROR_CHECKSUM;
    return mg_ctx_create_from_bytes(curve, bytes, len, out);
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
    HeavyOp no_capture_meanwhile;
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
MG_API int mg_ctx_table_bytes(const mg_ctx *ctx, uint64_t out2[2]) {
    if (!ctx || !out2) return MG_ERROR_INVALID_ARGUMENT;
    ctx->p->table_bytes(out2);
    return MG_OK;
}
MG_API uint64_t mg_ctx_num_variables(const mg_ctx *ctx) { return ctx ? ctx->p->n_vars() : 0; }
MG_API uint64_t mg_ctx_num_inputs(const mg_ctx *ctx) { return ctx ? ctx->p->n_inputs() : 0; }
MG_API int mg_ctx_num_shards(const mg_ctx *ctx) { return ctx ? (int)ctx->p->n_shards() : 0; }
MG_API void mg_ctx_destroy(mg_ctx *ctx) {
    if (!ctx) return;
    HeavyOp no_capture_meanwhile;
    delete ctx->p;
    delete ctx;
}

// ---------------------------------------------------------------------------------------------- verification (f-2)
struct mg_vk {
    Verifier *v;
};
MG_API int mg_vk_create(mg_curve_t curve, const uint64_t *alpha_g1, const uint64_t *beta_g2, const uint64_t *gamma_g2,
                        const uint64_t *delta_g2, const uint64_t *gamma_abc_g1, uint64_t n_inputs, mg_vk **out) {
    MG_TRY
    HeavyOp no_capture_meanwhile;
    if (!out) return MG_ERROR_INVALID_ARGUMENT;
    Verifier *v = nullptr;
    int rc = verifier_create((int)curve, alpha_g1, beta_g2, gamma_g2, delta_g2, gamma_abc_g1, n_inputs, &v);
    if (rc) return rc;
    *out = new mg_vk{v};
    return MG_SUCCESS;
    MG_CATCH
}
MG_API int mg_vk_create_from_bytes(mg_curve_t curve, const uint8_t *bytes, size_t len, mg_vk **out) {
    MG_TRY
    HeavyOp no_capture_meanwhile;
    if (!out) return MG_ERROR_INVALID_ARGUMENT;
    Verifier *v = nullptr;
    int rc = verifier_create_from_bytes((int)curve, bytes, len, &v);
    if (rc) return rc;
    *out = new mg_vk{v};
    return MG_SUCCESS;
    MG_CATCH
}
MG_API void mg_vk_destroy(mg_vk *vk) {
    HeavyOp no_capture_meanwhile;
    if (!vk) return;
    delete vk->v;
    delete vk;
}
MG_API uint64_t mg_vk_num_inputs(const mg_vk *vk) { return vk ? vk->v->n_inputs() : 0; }
MG_API size_t mg_vk_encoded_size(const mg_vk *vk) { return vk ? vk->v->encoded_size() : 0; }
MG_API int mg_vk_encode(const mg_vk *vk, uint8_t *out) {
    MG_TRY
    if (!vk) return MG_ERROR_INVALID_ARGUMENT;
    return vk->v->encode(out);
    MG_CATCH
}
MG_API int mg_vk_alpha_beta(const mg_vk *vk, uint8_t *out) {
    MG_TRY
    if (!vk) return MG_ERROR_INVALID_ARGUMENT;
    return vk->v->alpha_beta_bytes(out);
    MG_CATCH
}
MG_API int mg_groth16_verify(const mg_vk *vk, const uint64_t *inputs_mont, const uint64_t *proof_points, int *ok) {
    MG_TRY
    if (!vk) return MG_ERROR_INVALID_ARGUMENT;
    return vk->v->verify(inputs_mont, proof_points, ok);
    MG_CATCH
}
MG_API int mg_groth16_verify_batch(const mg_vk *vk, uint64_t k, const uint64_t *inputs_mont, const uint64_t *proof_points,
                                   const uint64_t *rand128, int *ok) {
    MG_TRY
    if (!vk) return MG_ERROR_INVALID_ARGUMENT;
    return vk->v->verify_batch(k, inputs_mont, proof_points, rand128, ok);
    MG_CATCH
}
MG_API int mg_pairing_check(mg_curve_t curve, const uint64_t *g1_affine, const uint64_t *g2_affine, size_t n, int *ok) {
    MG_TRY
    PairingEngine *pe = get_pairing_engine((int)curve);
    if (!pe) return MG_ERROR_INVALID_ARGUMENT;
    return pe->product_is_one((const u32 *)g1_affine, (const u32 *)g2_affine, n, ok);
    MG_CATCH
}
MG_API int mg_proof_decode(mg_curve_t curve, const uint8_t *proof_bytes, uint64_t *points_out) {
    MG_TRY
    return proof_decode((int)curve, proof_bytes, points_out);
    MG_CATCH
}
