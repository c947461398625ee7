// Scalar-field kernels on gfx950: radix-2 NTT, CSR SpMV, QAP pointwise step. Templated on the Fr
// parameters; instantiated in fr_bn254.hip / fr_bls381.hip.
//
// Replaces: ark-poly ^0.3.0 Radix2EvaluationDomain::{fft,ifft,coset_fft,
// coset_ifft}_in_place and ark-groth16 ^0.3.0 R1CStoQAP::witness_map's evaluate_constraint /
// mul_polynomials_in_evaluation_domain / divide_by_vanishing_poly_on_coset (reached from
// manta-crypto/src/arkworks/groth16.rs:597). Conventions restated from SURVEY.md App. A.1/B.3:
// omega_D = omega_{2^s}^(2^(s - log D)); ifft scales by D^-1; coset shift g = Fr multiplicative
// generator; natural order in and out.
#pragma once
#include "fp_dev.h"
#include "host_ec.h"
#include "params_gen.h"
#include "prover.h"
#include <map>
#include <mutex>

namespace mg {

// out[i] = base^i for i < n, given sq[k] = base^(2^k) (device array of `bits` elements)
template <class FrC>
__global__ __launch_bounds__(256) void powers_kernel(u32 *__restrict__ out, const u32 *__restrict__ sq, u32 n,
                                                     int bits, const u32 *__restrict__ scale /* 8 words or null */) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    typedef Fp<FrC> F;
    F acc = scale ? F::load(scale) : F::one();
    for (int k = 0; k < bits; ++k)
        if ((i >> k) & 1) acc = F::mul(acc, F::load(sq + 8 * k));
    acc.store(out + (size_t)i * 8);
}

template <class FrC> __global__ __launch_bounds__(256) void bitrev_kernel(u32 *__restrict__ data, unsigned log_n) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (1u << log_n)) return;
    const u32 j = __brev(i) >> (32 - log_n);
    if (i < j) {
        uint4 *p = reinterpret_cast<uint4 *>(data);
        uint4 a0 = p[2 * (size_t)i], a1 = p[2 * (size_t)i + 1];
        uint4 b0 = p[2 * (size_t)j], b1 = p[2 * (size_t)j + 1];
        p[2 * (size_t)i] = b0;
        p[2 * (size_t)i + 1] = b1;
        p[2 * (size_t)j] = a0;
        p[2 * (size_t)j + 1] = a1;
    }
}

// One DIT butterfly stage (after bit reversal). stage s: m = 2^s, butterflies (i0, i0 + m/2),
// twiddle = tw[j * (n/m)], tw[k] = omega^k for k < n/2.
template <class FrC>
__global__ __launch_bounds__(256) void ntt_stage_kernel(u32 *__restrict__ data, const u32 *__restrict__ tw,
                                                        unsigned log_n, unsigned s) {
    const u32 k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= (1u << (log_n - 1))) return;
    typedef Fp<FrC> F;
    const u32 half = 1u << (s - 1);
    const u32 j = k & (half - 1);
    const u32 i0 = ((k >> (s - 1)) << s) + j, i1 = i0 + half;
    F a = F::load(data + (size_t)i0 * 8), b = F::load(data + (size_t)i1 * 8);
    if (s > 1) { // stage 1 twiddle is 1
        F w = F::load(tw + ((size_t)j << (log_n - s)) * 8);
        b = F::mul(b, w);
    }
    F::add(a, b).store(data + (size_t)i0 * 8);
    F::sub(a, b).store(data + (size_t)i1 * 8);
}

template <class FrC>
__global__ __launch_bounds__(256) void scale_table_kernel(u32 *__restrict__ data, const u32 *__restrict__ table,
                                                          u32 n) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    typedef Fp<FrC> F;
    F::mul(F::load(data + (size_t)i * 8), F::load(table + (size_t)i * 8)).store(data + (size_t)i * 8);
}
template <class FrC>
__global__ __launch_bounds__(256) void scale_const_kernel(u32 *__restrict__ data, const u32 *__restrict__ k, u32 n) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    typedef Fp<FrC> F;
    F::mul(F::load(data + (size_t)i * 8), F::load(k)).store(data + (size_t)i * 8);
}

// CSR row dot products: one lane per row (rows of the manta-pay circuits hold <= a handful of terms)
template <class FrC>
__global__ __launch_bounds__(256) void spmv_kernel(const u32 *__restrict__ row_ptr, const u32 *__restrict__ col,
                                                   const u32 *__restrict__ val, const u32 *__restrict__ z,
                                                   u32 *__restrict__ out, u32 m) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    typedef Fp<FrC> F;
    F acc = F::zero();
    const F one = F::one();
    for (u32 k = row_ptr[i]; k < row_ptr[i + 1]; ++k) {
        F c = F::load(val + (size_t)k * 8);
        F x = F::load(z + (size_t)col[k] * 8);
        acc = F::add(acc, c == one ? x : F::mul(x, c));
    }
    acc.store(out + (size_t)i * 8);
}

template <class FrC>
__global__ __launch_bounds__(256) void qap_pointwise_kernel(u32 *__restrict__ a, const u32 *__restrict__ b,
                                                            const u32 *__restrict__ c, const u32 *__restrict__ zinv,
                                                            u32 n) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    typedef Fp<FrC> F;
    F ab = F::mul(F::load(a + (size_t)i * 8), F::load(b + (size_t)i * 8));
    ab = F::sub(ab, F::load(c + (size_t)i * 8));
    F::mul(ab, F::load(zinv)).store(a + (size_t)i * 8);
}

template <class FrC> class FrEngineT : public FrEngine {
  public:
    typedef host::HFp<FrC> HF;
    struct Domain {
        u32 *tw_fwd = nullptr, *tw_inv = nullptr; // omega^k, omega^-k, k < n/2
        u32 *coset_fwd = nullptr;                 // g^i
        u32 *coset_inv = nullptr;                 // n^-1 * g^-i
        u32 *consts = nullptr;                    // [0] n^-1, [1] (g^n - 1)^-1
    };
    int two_adicity() const override { return FrC::TWO_ADICITY; }

    static HF host_load(const u32 *w) {
        HF x;
        x.load_words(w);
        return x;
    }
    static HF hpow2k(HF x, int k) {
        for (int i = 0; i < k; ++i) x = HF::sqr(x);
        return x;
    }
    static HF from_u64(u64 v) {
        HF x = HF::zero();
        x.v[0] = v;
        return HF::to_mont(x);
    }

    int get_domain(unsigned log_n, Domain **out) {
        std::lock_guard<std::mutex> g(mu_);
        auto it = domains_.find(log_n);
        if (it != domains_.end()) {
            *out = &it->second;
            return MG_OK;
        }
        if ((int)log_n > FrC::TWO_ADICITY) return MG_ERR_DOMAIN;
        const size_t n = (size_t)1 << log_n;
        Domain d;
        HF root = hpow2k(host_load(FrC::ROOT), FrC::TWO_ADICITY - (int)log_n);
        HF root_inv = HF::inv(root);
        HF gen = host_load(FrC::GEN), gen_inv = HF::inv(gen);
        HF n_inv = HF::inv(from_u64((u64)n));
        HF zinv = HF::inv(HF::sub(hpow2k(gen, (int)log_n), HF::one()));
        const size_t half = n > 1 ? n / 2 : 1;
        MG_HIP(hipMalloc((void **)&d.tw_fwd, half * 32));
        MG_HIP(hipMalloc((void **)&d.tw_inv, half * 32));
        MG_HIP(hipMalloc((void **)&d.coset_fwd, n * 32));
        MG_HIP(hipMalloc((void **)&d.coset_inv, n * 32));
        MG_HIP(hipMalloc((void **)&d.consts, 64));
        // squarings of the four bases on the host, powers on the device
        u32 *d_sq = nullptr;
        MG_HIP(hipMalloc((void **)&d_sq, 4 * 33 * 32 + 32));
        std::vector<u32> hsq(4 * 33 * 8 + 8);
        HF bases[4] = {root, root_inv, gen, gen_inv};
        for (int b = 0; b < 4; ++b) {
            HF x = bases[b];
            for (int k = 0; k < 33; ++k) {
                x.store_words(&hsq[((size_t)b * 33 + k) * 8]);
                x = HF::sqr(x);
            }
        }
        n_inv.store_words(&hsq[4 * 33 * 8]);
        MG_HIP(hipMemcpy(d_sq, hsq.data(), hsq.size() * 4, hipMemcpyHostToDevice));
        const int bits = (int)log_n;
        const u32 gh = (u32)((half + 255) / 256), gn = (u32)((n + 255) / 256);
        hipLaunchKernelGGL((powers_kernel<FrC>), dim3(gh), dim3(256), 0, 0, d.tw_fwd, d_sq, (u32)half, bits,
                           (const u32 *)nullptr);
        hipLaunchKernelGGL((powers_kernel<FrC>), dim3(gh), dim3(256), 0, 0, d.tw_inv, d_sq + 33 * 8, (u32)half, bits,
                           (const u32 *)nullptr);
        hipLaunchKernelGGL((powers_kernel<FrC>), dim3(gn), dim3(256), 0, 0, d.coset_fwd, d_sq + 2 * 33 * 8, (u32)n,
                           bits + 1, (const u32 *)nullptr);
        hipLaunchKernelGGL((powers_kernel<FrC>), dim3(gn), dim3(256), 0, 0, d.coset_inv, d_sq + 3 * 33 * 8, (u32)n,
                           bits + 1, (const u32 *)(d_sq + 4 * 33 * 8));
        u32 hc[16];
        n_inv.store_words(hc);
        zinv.store_words(hc + 8);
        MG_HIP(hipMemcpy(d.consts, hc, 64, hipMemcpyHostToDevice));
        MG_HIP(hipDeviceSynchronize());
        hipFree(d_sq);
        auto ins = domains_.emplace(log_n, d);
        *out = &ins.first->second;
        return MG_OK;
    }

    int transform(u32 *d_data, unsigned log_n, bool inverse, bool coset, hipStream_t s) override {
        Domain *d;
        int rc = get_domain(log_n, &d);
        if (rc) return rc;
        const u32 n = 1u << log_n;
        const u32 gn = (n + 255) / 256, gh = (n / 2 + 255) / 256;
        if (!inverse && coset) hipLaunchKernelGGL((scale_table_kernel<FrC>), dim3(gn), dim3(256), 0, s, d_data, d->coset_fwd, n);
        if (log_n > 0) {
            hipLaunchKernelGGL((bitrev_kernel<FrC>), dim3(gn), dim3(256), 0, s, d_data, log_n);
            const u32 *tw = inverse ? d->tw_inv : d->tw_fwd;
            for (unsigned st = 1; st <= log_n; ++st)
                hipLaunchKernelGGL((ntt_stage_kernel<FrC>), dim3(gh ? gh : 1), dim3(256), 0, s, d_data, tw, log_n, st);
        }
        if (inverse) {
            if (coset)
                hipLaunchKernelGGL((scale_table_kernel<FrC>), dim3(gn), dim3(256), 0, s, d_data, d->coset_inv, n);
            else
                hipLaunchKernelGGL((scale_const_kernel<FrC>), dim3(gn), dim3(256), 0, s, d_data, d->consts, n);
        }
        MG_HIP(hipGetLastError());
        return MG_OK;
    }
    int spmv(const DevCsr &M, const u32 *d_z, u32 *d_out, u64 m, hipStream_t s) override {
        if (m == 0) return MG_OK;
        hipLaunchKernelGGL((spmv_kernel<FrC>), dim3((u32)((m + 255) / 256)), dim3(256), 0, s, M.row_ptr, M.col, M.val,
                           d_z, d_out, (u32)m);
        MG_HIP(hipGetLastError());
        return MG_OK;
    }
    int qap_pointwise(u32 *d_a, const u32 *d_b, const u32 *d_c, unsigned log_n, hipStream_t s) override {
        Domain *d;
        int rc = get_domain(log_n, &d);
        if (rc) return rc;
        const u32 n = 1u << log_n;
        hipLaunchKernelGGL((qap_pointwise_kernel<FrC>), dim3((n + 255) / 256), dim3(256), 0, s, d_a, d_b, d_c,
                           d->consts + 8, n);
        MG_HIP(hipGetLastError());
        return MG_OK;
    }
    void fr_mul(const u64 a[4], const u64 b[4], u64 out[4]) const override {
        HF x, y;
        std::memcpy(x.v, a, 32);
        std::memcpy(y.v, b, 32);
        HF r = HF::mul(x, y);
        std::memcpy(out, r.v, 32);
    }
    void fr_to_canonical(const u64 a[4], u64 out[4]) const override {
        HF x;
        std::memcpy(x.v, a, 32);
        HF r = HF::from_mont(x);
        std::memcpy(out, r.v, 32);
    }

  private:
    std::mutex mu_;
    std::map<unsigned, Domain> domains_;
};

} // namespace mg
