// Kernels and host orchestration of the pairing engine (pairing_dev.h), one instantiation per curve
// (pairing_bn254.hip / pairing_bls381.hip). SURVEY.md row f-2.
#pragma once
#include "engine.h"
#include "pairing_dev.h"
#include "verify.h"
#include <vector>

namespace mg {

// lane i: G2Prepared::from(Q_i) -> NCOEFF line-coefficient triples (infinity: all-zero block, never consumed)
template <class K>
__global__ __launch_bounds__(64) void g2_prepare_kernel(const u32 *__restrict__ q, size_t n, u32 *__restrict__ out) {
    typedef Pairing<K> P;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const typename P::F2 qx = P::F2::load(q + i * 2 * P::F2W), qy = P::F2::load(q + i * 2 * P::F2W + P::F2W);
    u32 *o = out + i * (size_t)P::NCOEFF * P::COEFFW;
    if (qx.is_zero() & qy.is_zero()) {
        for (int k = 0; k < P::NCOEFF * P::COEFFW; ++k) o[k] = 0;
        return;
    }
    P::prepare(qx, qy, o);
}

// lane i: f_i = Miller(P_i, coeffs[i]); a pair with an infinity member contributes 1 (ark-ec filters such pairs)
template <class K>
__global__ __launch_bounds__(64) void miller_kernel(const u32 *__restrict__ p, const u32 *const *__restrict__ coeffs,
                                                    const unsigned char *__restrict__ skip, size_t n, u32 *__restrict__ out) {
    typedef Pairing<K> P;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    typename P::F12 f = P::one12();
    const typename P::F px = P::F::load(p + i * 2 * P::N), py = P::F::load(p + i * 2 * P::N + P::N);
    if (!(skip[i] || (px.is_zero() & py.is_zero()))) P::miller(f, px, py, coeffs[i]);
    P::store12(f, out + i * P::F12W);
}

// lane t: out[t] = product of in[t*chunk .. min(n, (t+1)*chunk))
template <class K>
__global__ __launch_bounds__(64) void f12_product_kernel(const u32 *__restrict__ in, size_t n, size_t chunk, u32 *__restrict__ out) {
    typedef Pairing<K> P;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t lo = t * chunk;
    if (lo >= n) return;
    const size_t hi = lo + chunk < n ? lo + chunk : n;
    typename P::F12 acc, x, y;
    P::load12(acc, in + lo * P::F12W);
    for (size_t j = lo + 1; j < hi; ++j) {
        P::load12(x, in + j * P::F12W);
        P::mul12(y, acc, x);
        acc = y;
    }
    P::store12(acc, out + t * P::F12W);
}

template <class K>
__global__ __launch_bounds__(64) void final_exp_kernel(const u32 *__restrict__ in, size_t n, u32 *__restrict__ out) {
    typedef Pairing<K> P;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    typename P::F12 f, r;
    P::load12(f, in + i * P::F12W);
    P::final_exp(r, f);
    P::store12(r, out + i * P::F12W);
}

template <class K> class PairingEngineT : public PairingEngine {
  public:
    typedef Pairing<K> P;
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
            hipLaunchKernelGGL((g2_prepare_kernel<K>), dim3((unsigned)((n + 63) / 64)), dim3(64), 0, 0, dq, n, dc);
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

    int pairing_product(const u32 *p_affine_host, const u32 *const *d_coeffs, const unsigned char *skip, size_t n,
                        bool do_final_exp, u32 *out_f12_host) override {
        if (!p_affine_host || !d_coeffs || !n || !out_f12_host) return MG_ERR_ARG;
        u32 *dp = nullptr, *df = nullptr, *dg = nullptr;
        const u32 **dcp = nullptr;
        unsigned char *dskip = nullptr;
        std::vector<unsigned char> sk(n, 0);
        if (skip) sk.assign(skip, skip + n);
        hipError_t e = hipMalloc((void **)&dp, n * 2 * P::N * 4);
        if (e == hipSuccess) e = hipMalloc((void **)&df, n * P::F12W * 4);
        if (e == hipSuccess) e = hipMalloc((void **)&dg, (n / 8 + 2) * P::F12W * 4);
        if (e == hipSuccess) e = hipMalloc((void **)&dcp, n * sizeof(u32 *));
        if (e == hipSuccess) e = hipMalloc((void **)&dskip, n);
        if (e == hipSuccess) e = hipMemcpy(dp, p_affine_host, n * 2 * P::N * 4, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(dcp, d_coeffs, n * sizeof(u32 *), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(dskip, sk.data(), n, hipMemcpyHostToDevice);
        if (e == hipSuccess) {
            hipLaunchKernelGGL((miller_kernel<K>), dim3((unsigned)((n + 63) / 64)), dim3(64), 0, 0, dp, dcp, dskip, n, df);
            // product tree: chunks of 8 per lane until one element is left
            u32 *src = df, *dst = dg;
            size_t m = n;
            while (m > 1) {
                const size_t chunk = 8, outn = (m + chunk - 1) / chunk;
                hipLaunchKernelGGL((f12_product_kernel<K>), dim3((unsigned)((outn + 63) / 64)), dim3(64), 0, 0, src, m, chunk, dst);
                u32 *t = src;
                src = dst;
                dst = t;
                m = outn;
            }
            if (do_final_exp) {
                hipLaunchKernelGGL((final_exp_kernel<K>), dim3(1), dim3(64), 0, 0, src, (size_t)1, dst);
                src = dst;
            }
            e = hipMemcpy(out_f12_host, src, P::F12W * 4, hipMemcpyDeviceToHost);
        }
        hipFree(dp);
        hipFree(df);
        hipFree(dg);
        hipFree(dcp);
        hipFree(dskip);
        if (e != hipSuccess) {
            set_last_hip_error(e, "pairing product", __FILE__, __LINE__);
            return e == hipErrorOutOfMemory ? MG_ERR_OOM : MG_ERR_HIP;
        }
        return MG_OK;
    }
};

} // namespace mg
