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
    w::template prepare<false>(qx, qy, o);
}

// workgroup i (two wavefronts): f_i = Miller(P_i, Q_i); a pair with an infinity member contributes 1 (ark-ec filters such
// pairs). Q_i is either prepared already (coeffs[i] != null; wave 1 has nothing to do) or given as an affine point in
// q[i]: then wave 1 runs G2Prepared::from(Q_i) and feeds the line coefficients to wave 0's Miller loop through LDS as they
// appear -- the two chains overlap instead of running one after the other in two kernels.
template <class K>
__global__ __launch_bounds__(128) void miller_kernel(const u32 *__restrict__ p, const u32 *const *__restrict__ coeffs,
                                                     const u32 *__restrict__ q, const unsigned char *__restrict__ skip, size_t n,
                                                     u32 *__restrict__ out) {
    typedef Pairing<K> P;
    typedef PairingWave<K> PW;
    typedef PW w;
    const size_t i = blockIdx.x;
    const u32 *co = coeffs[i];
    const typename P::F px = P::F::load(p + i * 2 * P::N), py = P::F::load(p + i * 2 * P::N + P::N);
    const bool one = skip[i] || (px.is_zero() && py.is_zero());
    if (threadIdx.x == 0) w::counters()[0] = 0, w::counters()[1] = 0;
    __syncthreads(); // the only workgroup barrier: from here on the two wavefronts run different programs
    if (w::wave_id() == 1) {
        if (!co && !one) {
            const u32 *qi = q + i * 2 * P::F2W;
            w::template prepare<true>(P::F2::load(qi), P::F2::load(qi + P::F2W), nullptr);
        }
        return;
    }
    if (one) w::set_one(PW::R(0));
    else if (co) w::template miller<false>(PW::R(0), px, py, co);
    else w::template miller<true>(PW::R(0), px, py, nullptr);
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
        if (e == hipSuccess) e = hipMemcpy(dq, q_affine_host, qb, hipMemcpyHostToDevice);
        if (e == hipSuccess) {
            hipLaunchKernelGGL((g2_prepare_kernel<K>), dim3((unsigned)n), dim3(64), PW::prep_lds_bytes(), 0, dq, n, dc);
            e = hipDeviceSynchronize();
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
        if (!p_affine_host || !d_coeffs || !n || !out_f12_host) return MG_ERR_ARG;
        for (size_t i = 0; i < n; ++i)
            if (!d_coeffs[i] && !q_affine_host) return MG_ERR_ARG;
        // one device block and one upload: [P | Q | coefficient pointers | skip flags] then the Fq12 work arrays
        auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
        const size_t pb = n * 2 * P::N * 4, qb = q_affine_host ? n * 2 * P::F2W * 4 : 0, cb = n * sizeof(u32 *);
        const size_t o_q = up(pb), o_c = o_q + up(qb), o_s = o_c + up(cb), in_bytes = o_s + up(n);
        const size_t o_f = in_bytes, o_g = o_f + up(n * P::F12W * 4), total = o_g + up((n / 8 + 2) * P::F12W * 4);
        // pooled workspace (device block + pinned staging + a non-blocking stream of its own): no hipMalloc / hipFree per call --
        // hipFree synchronises the whole device and would stall every proof in flight on this GPU -- and nothing on stream 0
        Ws *w = ws_get(total, in_bytes > (size_t)P::F12W * 4 ? in_bytes : (size_t)P::F12W * 4);
        if (!w) return MG_ERR_OOM;
        unsigned char *stage = (unsigned char *)w->h;
        std::memset(stage, 0, in_bytes);
        std::memcpy(stage, p_affine_host, pb);
        if (qb) std::memcpy(stage + o_q, q_affine_host, qb);
        std::memcpy(stage + o_c, d_coeffs, cb);
        if (skip) std::memcpy(stage + o_s, skip, n);
        unsigned char *d = w->d;
        hipStream_t st = w->s;
        hipError_t e = hipMemcpyAsync(d, stage, in_bytes, hipMemcpyHostToDevice, st);
        if (e == hipSuccess) {
            u32 *df = (u32 *)(d + o_f), *dg = (u32 *)(d + o_g);
            hipLaunchKernelGGL((miller_kernel<K>), dim3((unsigned)n), dim3(128), PW::miller_lds_bytes(), st, (const u32 *)d,
                               (const u32 *const *)(d + o_c), qb ? (const u32 *)(d + o_q) : nullptr, (const unsigned char *)(d + o_s),
                               n, df);
            // product tree: chunks of 8 per wavefront until one element is left
            u32 *src = df, *dst = dg;
            size_t m = n;
            while (m > 1) {
                const size_t chunk = 8, outn = (m + chunk - 1) / chunk;
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
            if (e == hipSuccess) e = hipStreamSynchronize(st);
            if (e == hipSuccess) std::memcpy(out_f12_host, w->h, P::F12W * 4);
        }
        ws_put(w);
        if (e != hipSuccess) {
            set_last_hip_error(e, "pairing product", __FILE__, __LINE__);
            return e == hipErrorOutOfMemory ? MG_ERR_OOM : MG_ERR_HIP;
        }
        return MG_OK;
    }

    // ---- workspace pool (per engine = per device and curve)
    struct Ws {
        unsigned char *d = nullptr;
        void *h = nullptr;
        size_t dcap = 0, hcap = 0;
        hipStream_t s = nullptr;
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
            if (hipStreamCreateWithFlags(&w->s, hipStreamNonBlocking) != hipSuccess) {
                delete w;
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
