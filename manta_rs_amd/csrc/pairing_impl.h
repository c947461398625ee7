// Kernels and host orchestration of the pairing engine (pairing_dev.h), one instantiation per curve
// (pairing_bn254.hip / pairing_bls381.hip). SURVEY.md row f-2.
#pragma once
#include <mutex>
#include <vector>
#include "engine.h"
#include "pairing_coop.h"
#include "verify.h"
#include <cstring>
#include <vector>

namespace mg {

// One WAVEFRONT per unit of work (pairing_coop.h); every workgroup is a single wavefront with its Fq12 state in LDS.
// workgroup i: G2Prepared::from(Q_i) -> NCOEFF line-coefficient triples (infinity: all-zero block, never consumed)
template <class K>
__global__ __launch_bounds__(64) void g2_prepare_kernel(const u32 *__restrict__ q, size_t n, u32 *__restrict__ out) {
    typedef Pairing<K> P;
    const size_t i = blockIdx.x;
    typedef PairingWave<K> w;
    const typename P::F2 qx = P::F2::load(q + i * 2 * P::F2W), qy = P::F2::load(q + i * 2 * P::F2W + P::F2W);
    u32 *o = out + i * (size_t)P::NCOEFF * P::COEFFW;
    if (qx.is_zero() && qy.is_zero()) { // uniform over the wavefront
        for (int k = threadIdx.x; k < P::NCOEFF * P::COEFFW; k += 64) o[k] = 0;
        return;
    }
    w::template prepare<false>(qx, qy, o, P::F::zero(), P::F::zero(), P::F::zero());
}

// workgroup i (two wavefronts): f_i = Miller(P_i, Q_i); a pair with an infinity member contributes 1 (ark-ec filters such
// pairs). Q_i is either prepared already (coeffs[i] != null; wave 1 has nothing to do) or given as an affine point in
// q[i]: then wave 1 runs G2Prepared::from(Q_i) and feeds the line coefficients to wave 0's Miller loop through LDS as they
// appear -- the two chains overlap instead of running one after the other in two kernels.
// (p = null: ONE pair whose G1 point rides in the kernel arguments -- a point the host has just computed, e.g. a
// verification's prepared inputs, then needs no upload of its own in front of the kernel)
template <class K> struct G1Arg {
    u32 w[2 * Pairing<K>::N];
};
template <class K>
__global__ __launch_bounds__(128) void miller_kernel(const u32 *__restrict__ p, const u32 *const *__restrict__ coeffs,
                                                     const u32 *__restrict__ q, const unsigned char *__restrict__ skip, size_t n,
                                                     u32 *__restrict__ out, const G1Arg<K> p_arg, const u32 *__restrict__ p_xyzz,
                                                     u32 n_xyzz) {
    typedef Pairing<K> P;
    typedef PairingWave<K> PW;
    typedef PW w;
    const size_t i = blockIdx.x;
    const u32 *co = coeffs[i];
    typename P::F px, py, pz = P::F::one();
    bool p_inf;
    if (i < n_xyzz) { // (X, Y, ZZ, ZZZ) in device memory (the result of a scalar multiplication): the lines times ZZ ZZZ, no inversion
        const u32 *g = p_xyzz + i * 4 * P::N;
        const typename P::F X = P::F::load(g), Y = P::F::load(g + P::N), ZZ = P::F::load(g + 2 * P::N), ZZZ = P::F::load(g + 3 * P::N);
        p_inf = ZZ.is_zero();
        px = P::F::mul(X, ZZZ), py = P::F::mul(Y, ZZ), pz = P::F::mul(ZZ, ZZZ);
    } else {
        if (p) {
            px = P::F::load(p + i * 2 * P::N), py = P::F::load(p + i * 2 * P::N + P::N);
        } else {
#pragma unroll
            for (int k = 0; k < P::N; ++k) px.v[k] = p_arg.w[k], py.v[k] = p_arg.w[P::N + k];
        }
        p_inf = px.is_zero() && py.is_zero();
    }
    const bool one = skip[i] || p_inf;
    if (threadIdx.x == 0) w::counters()[0] = 0, w::counters()[1] = 0;
    __syncthreads(); // the only workgroup barrier: from here on the two wavefronts run different programs
    if (w::wave_id() == 1) { // the lines, as ring entries: from the stored table of Q, or from G2Prepared::from(Q) as it runs
        if (one) return;
        if (co) w::scale_stored(co, px, py, pz);
        else w::template prepare<true>(P::F2::load(q + i * 2 * P::F2W), P::F2::load(q + i * 2 * P::F2W + P::F2W), nullptr, px, py, pz);
        return;
    }
    if (one) w::set_one(PW::R(0));
    else w::miller(PW::R(0));
    if (threadIdx.x < 6) w::ld(PW::R(0) + threadIdx.x).store(out + i * P::F12W + threadIdx.x * P::F2W);
}

// workgroup t: out[t] = product of in[t*chunk .. min(n, (t+1)*chunk))
template <class K>
__global__ __launch_bounds__(64) void f12_product_kernel(const u32 *__restrict__ in, size_t n, size_t chunk, u32 *__restrict__ out) {
    typedef Pairing<K> P;
    typedef PairingWave<K> PW;
    const size_t t = blockIdx.x;
    typedef PW w;
    const size_t lo = t * chunk, hi = lo + chunk < n ? lo + chunk : n;
    if (threadIdx.x < 6) w::st(PW::R(0) + threadIdx.x, P::F2::load(in + lo * P::F12W + threadIdx.x * P::F2W));
    PW::sync();
    for (size_t j = lo + 1; j < hi; ++j) {
        if (threadIdx.x < 6) w::st(PW::R(1) + threadIdx.x, P::F2::load(in + j * P::F12W + threadIdx.x * P::F2W));
        PW::sync();
        w::mul12(PW::R(0), PW::R(0), PW::R(1));
    }
    if (threadIdx.x < 6) w::ld(PW::R(0) + threadIdx.x).store(out + t * P::F12W + threadIdx.x * P::F2W);
}

// one workgroup: out = final_exponentiation(in)
template <class K> __global__ __launch_bounds__(64) void final_exp_kernel(const u32 *__restrict__ in, u32 *__restrict__ out) {
    typedef Pairing<K> P;
    typedef PairingWave<K> PW;
    typedef PW w;
    if (threadIdx.x < 6) w::st(PW::R(0) + threadIdx.x, P::F2::load(in + threadIdx.x * P::F2W));
    PW::sync();
    w::final_exp();
    if (threadIdx.x < 6) w::ld(PW::R(0) + threadIdx.x).store(out + threadIdx.x * P::F2W);
}

template <class K> class PairingEngineT : public PairingEngine {
  public:
    typedef Pairing<K> P;
    typedef PairingWave<K> PW;
    int n_coeffs() const override { return P::NCOEFF; }
    int coeff_words() const override { return P::COEFFW; }
    int f12_words() const override { return P::F12W; }
    int fq_words() const override { return P::N; }

    int prepare(const u32 *q_affine_host, size_t n, u32 **d_out) override {
        if (!q_affine_host || !n || !d_out) return MG_ERR_ARG;
        u32 *dq = nullptr, *dc = nullptr;
        const size_t qb = n * 2 * P::F2W * 4, cb = n * (size_t)P::NCOEFF * P::COEFFW * 4;
        hipError_t e = hipMalloc((void **)&dq, qb);
        if (e == hipSuccess) e = hipMalloc((void **)&dc, cb);
        if (e == hipSuccess) e = memcpy_sync(dq, q_affine_host, qb, hipMemcpyHostToDevice);
        if (e == hipSuccess) {
            hipLaunchKernelGGL((g2_prepare_kernel<K>), dim3((unsigned)n), dim3(64), PW::prep_lds_bytes(), setup_stream(), dq, n, dc);
            e = setup_sync();
        }
        hipFree(dq);
        if (e != hipSuccess) {
            hipFree(dc);
            set_last_hip_error(e, "pairing prepare", __FILE__, __LINE__);
            return e == hipErrorOutOfMemory ? MG_ERR_OOM : MG_ERR_HIP;
        }
        *d_out = dc;
        return MG_OK;
    }

    int pairing_product(const u32 *p_affine_host, const u32 *const *d_coeffs, const u32 *q_affine_host,
                        const unsigned char *skip, size_t n, bool do_final_exp, u32 *out_f12_host) override {
        void *h = nullptr;
        const int rc = pairing_product_begin(p_affine_host, d_coeffs, q_affine_host, skip, n, n, &h);
        return rc ? rc : pairing_product_end(h, nullptr, do_final_exp, out_f12_host);
    }
    // the same with the G1 points of the first n_xyzz pairs taken from DEVICE memory as (X, Y, ZZ, ZZZ) -- what `producer`, the
    // stream of the kernel that makes them, leaves there (p_affine_host's first n_xyzz entries are ignored)
    int pairing_product_xyzz(const u32 *p_affine_host, const u32 *d_p_xyzz, size_t n_xyzz, void *producer, const u32 *const *d_coeffs,
                             const u32 *q_affine_host, const unsigned char *skip, size_t n, bool do_final_exp,
                             u32 *out_f12_host) override {
        if (n_xyzz > n || (n_xyzz && (!d_p_xyzz || !producer))) return MG_ERR_ARG;
        void *h = nullptr;
        const int rc = begin_impl(p_affine_host, d_coeffs, q_affine_host, skip, n, n, &h, d_p_xyzz, n_xyzz, (hipStream_t)producer);
        return rc ? rc : pairing_product_end(h, nullptr, do_final_exp, out_f12_host);
    }

    // The same product in two steps: begin() starts the Miller loops of the first n_early pairs; end() takes the G1 points of
    // the remaining ones (their G2 sides -- prepared coefficients -- were given to begin()), runs their Miller loops on a second
    // stream next to the ones still in flight, then the product tree and the final exponentiation. A verification computes
    // its prepared-inputs point (an MSM) between the two calls instead of in front of all three Miller loops.
    // factors per wavefront and level of the product tree: a level is chunk - 1 dependent Fq12 products (5.6 us each), so a batch
    // of 259 Miller values costs 18 of them in chunks of 8 (three levels) and 11 in chunks of 3 (six levels)
    static constexpr size_t PRODUCT_CHUNK = 3;
    int pairing_product_begin(const u32 *p_affine_host, const u32 *const *d_coeffs, const u32 *q_affine_host,
                              const unsigned char *skip, size_t n, size_t n_early, void **handle) override {
        return begin_impl(p_affine_host, d_coeffs, q_affine_host, skip, n, n_early, handle, nullptr, 0, nullptr);
    }
    int begin_impl(const u32 *p_affine_host, const u32 *const *d_coeffs, const u32 *q_affine_host, const unsigned char *skip, size_t n,
                   size_t n_early, void **handle, const u32 *d_p_xyzz, size_t n_xyzz, hipStream_t producer) {
        if (!p_affine_host || !d_coeffs || !n || !handle || n_early > n || !n_early) return MG_ERR_ARG;
        for (size_t i = 0; i < n; ++i)
            if (!d_coeffs[i] && (!q_affine_host || i >= n_early)) return MG_ERR_ARG;
        // one device block and one upload: [P | Q | coefficient pointers | skip flags] then the Fq12 work arrays
        auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
        const size_t pb = n * 2 * P::N * 4, qb = q_affine_host ? n * 2 * P::F2W * 4 : 0, cb = n * sizeof(u32 *);
        const size_t o_q = up(pb), o_c = o_q + up(qb), o_s = o_c + up(cb), in_bytes = o_s + up(n);
        const size_t o_f = in_bytes, o_g = o_f + up(n * P::F12W * 4), total = o_g + up((n / PRODUCT_CHUNK + 2) * P::F12W * 4);
        // pooled workspace (device block + pinned staging + non-blocking streams of its own): no hipMalloc / hipFree per call --
        // hipFree synchronises the whole device and would stall every proof in flight on this GPU -- and nothing on stream 0
        const size_t late_bytes = up((n - n_early) * 2 * P::N * 4);
        const size_t hmain = up(in_bytes > (size_t)P::F12W * 4 ? in_bytes : (size_t)P::F12W * 4);
        Ws *w = ws_get(total, hmain + late_bytes);
        if (!w) return MG_ERR_OOM;
        unsigned char *stage = (unsigned char *)w->h;
        std::memset(stage, 0, in_bytes);
        std::memcpy(stage, p_affine_host, n_early * 2 * P::N * 4);
        if (qb) std::memcpy(stage + o_q, q_affine_host, qb);
        std::memcpy(stage + o_c, d_coeffs, cb);
        if (skip) std::memcpy(stage + o_s, skip, n);
        w->n = n, w->n_early = n_early, w->o_q = qb ? o_q : 0, w->o_c = o_c, w->o_s = o_s, w->o_f = o_f, w->o_g = o_g, w->h_late = hmain;
        unsigned char *d = w->d;
        hipError_t e = hipMemcpyAsync(d, stage, in_bytes, hipMemcpyHostToDevice, w->s);
        if (e == hipSuccess && n_early < n) e = hipEventRecord(w->ev, w->s); // the late pairs' coefficient pointers and flags sit in the block this upload fills
        if (e == hipSuccess && n_xyzz) { // the points another stream's kernel is still making
            e = hipEventRecord(w->evx, producer);
            if (e == hipSuccess) e = hipStreamWaitEvent(w->s, w->evx, 0);
        }
        if (e == hipSuccess)
            hipLaunchKernelGGL((miller_kernel<K>), dim3((unsigned)n_early), dim3(128), PW::miller_lds_bytes(), w->s, (const u32 *)d,
                               (const u32 *const *)(d + o_c), qb ? (const u32 *)(d + o_q) : nullptr, (const unsigned char *)(d + o_s),
                               n_early, (u32 *)(d + o_f), G1Arg<K>{}, d_p_xyzz, (u32)(n_xyzz < n_early ? n_xyzz : n_early));
        if (e == hipSuccess && n_early < n) e = hipEventRecord(w->ev3, w->s); // the early Miller loops are done
        if (e != hipSuccess) {
            hipStreamSynchronize(w->s);
            ws_put(w);
            set_last_hip_error(e, "pairing product", __FILE__, __LINE__);
            return e == hipErrorOutOfMemory ? MG_ERR_OOM : MG_ERR_HIP;
        }
        *handle = w;
        return MG_OK;
    }

    int pairing_product_end(void *handle, const u32 *p_late_affine_host, bool do_final_exp, u32 *out_f12_host) override {
        Ws *w = (Ws *)handle;
        if (!w) return MG_ERR_ARG;
        const size_t n = w->n, n_late = n - w->n_early;
        unsigned char *d = w->d;
        hipStream_t st = w->s;
        hipError_t e = hipSuccess;
        if (n_late && (!p_late_affine_host || !out_f12_host)) e = hipErrorInvalidValue;
        if (e == hipSuccess && n_late) {
            const size_t off = w->n_early * 2 * P::N * 4, lb = n_late * 2 * P::N * 4;
            G1Arg<K> arg{};
            e = hipStreamWaitEvent(w->s2, w->ev, 0);
            if (n_late == 1) { // the usual case (a verification's prepared inputs): the point travels with the launch
                std::memcpy(arg.w, p_late_affine_host, lb);
            } else if (e == hipSuccess) {
                std::memcpy((unsigned char *)w->h + w->h_late, p_late_affine_host, lb);
                e = hipMemcpyAsync(d + off, (unsigned char *)w->h + w->h_late, lb, hipMemcpyHostToDevice, w->s2);
            }
            if (e == hipSuccess) {
                hipLaunchKernelGGL((miller_kernel<K>), dim3((unsigned)n_late), dim3(128), PW::miller_lds_bytes(), w->s2,
                                   n_late == 1 ? (const u32 *)nullptr : (const u32 *)(d + off),
                                   (const u32 *const *)(d + w->o_c) + w->n_early, (const u32 *)nullptr,
                                   (const unsigned char *)(d + w->o_s) + w->n_early, n_late,
                                   (u32 *)(d + w->o_f) + w->n_early * P::F12W, arg, (const u32 *)nullptr, 0u);
            }
            // the rest runs behind the late loops on THEIR stream: they end last, and by then the early ones' event has long been
            // signalled -- waiting the other way round (the first stream for the late loops) left 30-70 us between the end of
            // the late Miller kernel and the product
            st = w->s2;
            if (e == hipSuccess) e = hipStreamWaitEvent(st, w->ev3, 0);
        }
        if (e == hipSuccess) {
            u32 *df = (u32 *)(d + w->o_f), *dg = (u32 *)(d + w->o_g);
            // product tree: chunks of PRODUCT_CHUNK per wavefront until one element is left
            u32 *src = df, *dst = dg;
            size_t m = n;
            while (m > 1) {
                const size_t chunk = PRODUCT_CHUNK, outn = (m + chunk - 1) / chunk;
                hipLaunchKernelGGL((f12_product_kernel<K>), dim3((unsigned)outn), dim3(64), PW::lds_bytes(2), st, src, m, chunk, dst);
                u32 *t = src;
                src = dst;
                dst = t;
                m = outn;
            }
            if (do_final_exp) {
                hipLaunchKernelGGL((final_exp_kernel<K>), dim3(1), dim3(64), PW::lds_bytes(PW::FINAL_EXP_REGS), st, src, dst);
                src = dst;
            }
            e = hipMemcpyAsync(w->h, src, P::F12W * 4, hipMemcpyDeviceToHost, st);
        }
        const hipError_t e2 = hipStreamSynchronize(st); // (also on the error paths: nothing of this call stays in flight)
        if (n_late) hipStreamSynchronize(w->s);
        if (e == hipSuccess) e = e2;
        if (e == hipSuccess) std::memcpy(out_f12_host, w->h, P::F12W * 4);
        ws_put(w);
        if (e != hipSuccess) {
            set_last_hip_error(e, "pairing product", __FILE__, __LINE__);
            return e == hipErrorOutOfMemory ? MG_ERR_OOM : MG_ERR_HIP;
        }
        return MG_OK;
    }
    void pairing_product_abandon(void *handle) override {
        Ws *w = (Ws *)handle;
        if (!w) return;
        hipStreamSynchronize(w->s);
        ws_put(w);
    }

    // ---- workspace pool (per engine = per device and curve)
    struct Ws {
        unsigned char *d = nullptr;
        void *h = nullptr;
        size_t dcap = 0, hcap = 0;
        hipStream_t s = nullptr, s2 = nullptr; // s2: the Miller loops of pairs handed in late
        hipEvent_t ev = nullptr, ev3 = nullptr, evx = nullptr; // upload done; early Miller loops done; a producer stream's points are there
        size_t n = 0, n_early = 0, o_q = 0, o_c = 0, o_s = 0, o_f = 0, o_g = 0, h_late = 0; // the product in flight
    };
    std::mutex ws_mu_;
    std::vector<Ws *> ws_free_;
    Ws *ws_get(size_t dbytes, size_t hbytes) {
        Ws *w = nullptr;
        {
            std::lock_guard<std::mutex> g(ws_mu_);
            if (!ws_free_.empty()) {
                w = ws_free_.back();
                ws_free_.pop_back();
            }
        }
        if (!w) {
            w = new Ws();
            if (!(w->s = stream_pool_get_normal()) || !(w->s2 = stream_pool_get_normal()) ||
                hipEventCreateWithFlags(&w->ev, hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&w->ev3, hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&w->evx, hipEventDisableTiming) != hipSuccess) {
                delete w; // (creation failures at start-up only; the handles made so far are left to process exit)
                return nullptr;
            }
        }
        if (w->dcap < dbytes) {
            if (w->d) hipFree(w->d);
            w->d = nullptr, w->dcap = 0;
            const size_t cap = dbytes + dbytes / 4 + 4096;
            if (hipMalloc((void **)&w->d, cap) != hipSuccess) {
                ws_put(w);
                return nullptr;
            }
            w->dcap = cap;
        }
        if (w->hcap < hbytes) {
            if (w->h) hipHostFree(w->h);
            w->h = nullptr, w->hcap = 0;
            const size_t cap = hbytes + hbytes / 4 + 4096;
            if (hipHostMalloc(&w->h, cap, hipHostMallocDefault) != hipSuccess) {
                ws_put(w);
                return nullptr;
            }
            w->hcap = cap;
        }
        return w;
    }
    void ws_put(Ws *w) {
        std::lock_guard<std::mutex> g(ws_mu_);
        ws_free_.push_back(w);
    }

    int product_is_one(const u32 *p_affine_host, const u32 *q_affine_host, size_t n, int *ok) override {
        if (!p_affine_host || !q_affine_host || !n || !ok) return MG_ERR_ARG;
        *ok = 0;
        std::vector<const u32 *> cp(n, nullptr);
        std::vector<unsigned char> skip(n, 0);
        for (size_t i = 0; i < n; ++i) {
            u32 any = 0;
            for (int k = 0; k < 2 * P::F2W; ++k) any |= q_affine_host[i * 2 * P::F2W + k];
            skip[i] = any == 0;
        }
        std::vector<u32> out(P::F12W);
        const int rc = pairing_product(p_affine_host, cp.data(), q_affine_host, skip.data(), n, true, out.data());
        if (rc) return rc;
        bool one = true;
        for (int k = 0; k < P::F12W; ++k) one = one && out[k] == (k < P::N ? K::Fq::R[k] : 0u);
        *ok = one ? 1 : 0;
        return MG_OK;
    }
};

} // namespace mg
