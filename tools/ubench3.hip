// prototype: reduced-radix (28/29-bit limb) carry-free Montgomery product vs the 32-bit product-scanning one
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "../manta_rs_amd/csrc/params_gen.h"
#include "../manta_rs_amd/csrc/fp_dev.h"
using namespace mg;
template <int K, int LB, bool INL> struct R {
    u32 v[K];
    static constexpr u32 MASK = (1u << LB) - 1;
    static __device__ __forceinline__ R mul_body(const R &a, const R &b, const u32 *P, u32 INV) {
        u64 acc = 0; u32 m[K]; R t;
#pragma unroll
        for (int k = 0; k < K; ++k) {
#pragma unroll
            for (int i = 0; i < k; ++i) { acc += (u64)a.v[i] * b.v[k - i]; acc += (u64)m[i] * P[k - i]; }
            acc += (u64)a.v[k] * b.v[0];
            m[k] = ((u32)acc * INV) & MASK;
            acc += (u64)m[k] * P[0];
            acc >>= LB;
        }
#pragma unroll
        for (int k = K; k < 2 * K; ++k) {
#pragma unroll
            for (int i = k - K + 1; i < K; ++i) { acc += (u64)a.v[i] * b.v[k - i]; acc += (u64)m[i] * P[k - i]; }
            t.v[k - K] = (u32)acc & MASK; acc >>= LB;
        }
        return t;
    }
};
// constants: any odd "modulus-like" limbs; throughput only
template <int K> struct PC { static constexpr u32 P[16] = {0x0ffaaab,0x9feffff,0x153ffff,0xeabfffe,0x6b0f624,0x730d2a0,0x38512bf,0x4774b84,0x34bacd7,0xb1ba7b6,0x97fe69a,0xa0111ea,0x1234567,0x0abcdef,0x1111111,0x2222222}; };
template <int K, int LB> __device__ __noinline__ R<K, LB, false> mul_call(const R<K, LB, false> a, const R<K, LB, false> b) {
    return R<K, LB, false>::mul_body(a, b, PC<K>::P, 0x12345677u);
}
template <int K, int LB, bool CALL> __global__ __launch_bounds__(256) void k_r(u32 *out, const u32 *in, int iters) {
    typedef R<K, LB, false> F;
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    F a, b;
    for (int i = 0; i < K; ++i) { a.v[i] = in[(t % 1024) * 12 + (i % 12)] & F::MASK; b.v[i] = in[((t + 1) % 1024) * 12 + (i % 12)] & F::MASK; }
    for (int i = 0; i < iters; ++i) {
        if (CALL) { a = mul_call<K, LB>(a, b); b = mul_call<K, LB>(b, a); }
        else { a = F::mul_body(a, b, PC<K>::P, 0x12345677u); b = F::mul_body(b, a, PC<K>::P, 0x12345677u); }
    }
    for (int i = 0; i < K; ++i) out[(size_t)t * 16 + i] = a.v[i];
}
template <class C> __global__ __launch_bounds__(256) void k_fpmul(u32 *out, const u32 *in, int iters) {
    typedef Fp<C> F;
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    F a = F::load(in + (size_t)(t % 1024) * F::N), b = F::load(in + (size_t)((t + 1) % 1024) * F::N);
    for (int i = 0; i < iters; ++i) { a = F::mul(a, b); b = F::mul(b, a); }
    a.store(out + (size_t)t * 16);
}
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const double clk = p.clockRate * 1e3;
    u32 *out; hipMalloc(&out, (size_t)p.multiProcessorCount * 8 * 256 * 16 * 4);
    u32 *in; hipMalloc(&in, 1024 * 12 * 4);
    std::vector<u32> h(1024 * 12); for (size_t i = 0; i < h.size(); ++i) h[i] = (u32)(i * 2654435761u) >> 3; hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int bpc : {1, 2, 4, 8}) {
        const int blocks = p.multiProcessorCount * bpc;
        auto fp = [&](const char *name, auto kern, int iters) {
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, in, 2); hipDeviceSynchronize();
            hipEventRecord(e0); hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, in, iters); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double muls = (double)blocks * 256 * iters * 2.0;
            printf("waves/SIMD %d  %-34s %8.3f ms  %7.2f G mul/s  %6.0f cycles/wave-mul/SIMD\n", bpc, name, ms, muls / (ms * 1e-3) / 1e9, (p.multiProcessorCount * 4.0 * clk) / (muls / 64 / (ms * 1e-3)));
        };
        fp("Fp<Bls381Fq> 12x32 (asm, call)", k_fpmul<Bls381FqCfg>, 500);
        fp("R<14,28> call", k_r<14, 28, true>, 500);
        fp("R<14,28> inline", k_r<14, 28, false>, 500);
        fp("Fp<Bn254Fq> 8x32 (asm, call)", k_fpmul<Bn254FqCfg>, 1000);
        fp("R<9,29> call", k_r<9, 29, true>, 1000);
        fp("R<9,29> inline", k_r<9, 29, false>, 1000);
        fp("R<10,28> call", k_r<10, 28, true>, 1000);
    }
    return 0;
}
