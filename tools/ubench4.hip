// Probe for VERDICT r1 item 5(d): could FP64 vector FMAs beat v_mad_u64_u32 for the 381-bit Montgomery product?
// Measures, per wavefront and SIMD: (1) the issue interval of v_fma_f64, v_add_f64, v_lshl_add_u64 and v_mad_u64_u32 in
// independent chains; (2) the instruction stream of a 52-bit-limb FMA product in Emmart's formulation (per limb
// product: hi = fma_rz(a, b, 2^104); lo = fma_rz(a, b, (2^104 + 2^52) - hi); two 64-bit integer accumulations of the bit
// patterns) for an 8 x 8 operand, twice (a*b and m*p) -- the arithmetic a Montgomery product over 8 x 52-bit limbs
// needs, WITHOUT the per-column m computation and carry resolution (so a lower bound on its cost); against the 14 x 28-bit
// integer product the MSM kernels use (R<14,28> of ubench3.hip: 2270-2310 cycles per wave at 2+ waves per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint64_t u64;
typedef uint32_t u32;

template <int MODE> __global__ __launch_bounds__(256) void k_rate(u64 *out, int iters) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    double d[8];
    u64 q[8];
    for (int i = 0; i < 8; ++i) d[i] = 1.0 + t * 1e-9 + i, q[i] = t * 977 + i;
    const double m = 1.0000001, c = 0.5;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) d[i] = __builtin_fma(d[i], m, c);                       // v_fma_f64
            if (MODE == 1) d[i] = d[i] + c;                                       // v_add_f64
            if (MODE == 2) q[i] = (q[i] << 1) + q[(i + 1) & 7];                   // v_lshl_add_u64
            if (MODE == 3) q[i] = (u64)(u32)q[i] * (u32)(q[(i + 1) & 7]) + q[i];  // v_mad_u64_u32
        }
    }
    u64 x = 0;
    for (int i = 0; i < 8; ++i) x ^= q[i] ^ (u64)__double_as_longlong(d[i]);
    out[t] = x;
}

// the FMA formulation's inner arithmetic for one 8 x 52-bit product pair (a*b, m*p)
__global__ __launch_bounds__(256) void k_fma_product(u64 *out, int iters) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    double a[8], b[8], mm[8], p[8];
    for (int i = 0; i < 8; ++i) {
        a[i] = (double)((u64)(t * 2654435761u + i * 97) & ((1ull << 52) - 1));
        b[i] = (double)((u64)(t * 40503u + i * 131071) & ((1ull << 52) - 1));
        mm[i] = a[i] + 3.0, p[i] = (double)(0xfffffffffffffull - 977 * i);
    }
    const double C1 = 0x1p104, C2 = 0x1p104 + 0x1p52;
    u64 acc[17];
    for (int k = 0; k < 17; ++k) acc[k] = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const double hi = __builtin_fma(a[i], b[j], C1);
                const double lo = __builtin_fma(a[i], b[j], C2 - hi);
                acc[i + j + 1] += (u64)__double_as_longlong(hi);
                acc[i + j] += (u64)__double_as_longlong(lo);
                const double hi2 = __builtin_fma(mm[i], p[j], C1);
                const double lo2 = __builtin_fma(mm[i], p[j], C2 - hi2);
                acc[i + j + 1] += (u64)__double_as_longlong(hi2);
                acc[i + j] += (u64)__double_as_longlong(lo2);
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = (double)(acc[i] & ((1ull << 52) - 1)), mm[i] = (double)(acc[i + 8] & ((1ull << 52) - 1));
    }
    u64 x = 0;
    for (int k = 0; k < 17; ++k) x ^= acc[k];
    out[t] = x;
}

// ---- the 14 x 28-bit integer product (fpr_dev.h FpR::mul) in two codings: as the compiler schedules the C expression
// (it keeps the a*b and m*p sums of a column in two accumulator chains and joins them with a v_lshl_add_u64) and with
// every multiply-add of a column forced into ONE dependent chain through inline asm (28 fewer instructions per product).
static constexpr u32 PL[14] = {0x0ffaaab, 0x9feffff, 0x153ffff, 0xeabfffe, 0x6b0f624, 0x730d2a0, 0x38512bf, 0x4774b84, 0x34bacd7, 0xb1ba7b6, 0x97fe69a, 0xa0111ea, 0x1234567, 0x0abcdef};
__device__ __forceinline__ void mad_vv(u64 &acc, u32 a, u32 b) { asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b) : "vcc"); }
__device__ __forceinline__ void mad_vs(u64 &acc, u32 a, u32 k) { asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "s"(k) : "vcc"); }
template <bool ASM> struct R14 {
    u32 v[14];
    static __device__ __forceinline__ R14 mul(const R14 &a, const R14 &b) {
        constexpr int K = 14, LB = 28;
        constexpr u32 MASK = (1u << LB) - 1, INV = 0x12345677u;
        u64 acc = 0;
        u32 m[K];
        R14 t;
#pragma unroll
        for (int k = 0; k < K; ++k) {
#pragma unroll
            for (int i = 0; i < k; ++i) {
                if (ASM) mad_vv(acc, a.v[i], b.v[k - i]), mad_vs(acc, m[i], PL[k - i]);
                else acc += (u64)a.v[i] * b.v[k - i], acc += (u64)m[i] * PL[k - i];
            }
            if (ASM) mad_vv(acc, a.v[k], b.v[0]);
            else acc += (u64)a.v[k] * b.v[0];
            m[k] = ((u32)acc * INV) & MASK;
            if (ASM) mad_vs(acc, m[k], PL[0]);
            else acc += (u64)m[k] * PL[0];
            acc >>= LB;
        }
#pragma unroll
        for (int k = K; k < 2 * K - 1; ++k) {
#pragma unroll
            for (int i = k - K + 1; i < K; ++i) {
                if (ASM) mad_vv(acc, a.v[i], b.v[k - i]), mad_vs(acc, m[i], PL[k - i]);
                else acc += (u64)a.v[i] * b.v[k - i], acc += (u64)m[i] * PL[k - i];
            }
            t.v[k - K] = (u32)acc & MASK;
            acc >>= LB;
        }
        t.v[K - 1] = (u32)acc;
        return t;
    }
};
template <bool ASM> __global__ __launch_bounds__(256) void k_r14(u64 *out, int iters) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    R14<ASM> a, b;
    for (int i = 0; i < 14; ++i) a.v[i] = (t * 2654435761u + i * 977) & 0xfffffff, b.v[i] = (t * 40503u + i * 131) & 0xfffffff;
    for (int i = 0; i < iters; ++i) {
        a = R14<ASM>::mul(a, b);
        b = R14<ASM>::mul(b, a);
    }
    u64 x = 0;
    for (int i = 0; i < 14; ++i) x ^= (u64)a.v[i] << (i & 31);
    out[t] = x;
}
__global__ __launch_bounds__(256) void k_dep_chain(u64 *out, int iters) { // ONE dependent chain of v_mad_u64_u32
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    u64 acc = t;
    u32 a = t * 977 + 1, b = t * 131 + 7;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) mad_vv(acc, a, b);
    }
    out[t] = acc;
}

int main() {
    hipDeviceProp_t pr;
    hipGetDeviceProperties(&pr, 0);
    const double clk = pr.clockRate * 1e3;
    u64 *out;
    hipMalloc(&out, (size_t)pr.multiProcessorCount * 8 * 256 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    const char *names[4] = {"v_fma_f64", "v_add_f64", "v_lshl_add_u64", "v_mad_u64_u32"};
    for (int bpc : {1, 2, 4}) {
        const int blocks = pr.multiProcessorCount * bpc;
        auto run = [&](const char *name, auto kern, int iters, double ops_per_iter) {
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 2);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double wave_ops = (double)blocks * 4 * iters * ops_per_iter; // per wavefront
            const double cyc = (pr.multiProcessorCount * 4.0 * clk) * (ms * 1e-3) / wave_ops;
            printf("waves/SIMD %d  %-26s %8.3f ms  %7.2f cycles per wave-%s per SIMD\n", bpc, name, ms, cyc,
                   ops_per_iter > 8.5 ? "product(8x8x2 limb products)" : "instruction");
        };
        run(names[0], k_rate<0>, 4000, 8);
        run(names[1], k_rate<1>, 4000, 8);
        run(names[2], k_rate<2>, 4000, 8);
        run(names[3], k_rate<3>, 4000, 8);
        run("v_mad_u64_u32 dependent", k_dep_chain, 4000, 8);
        for (int v = 0; v < 2; ++v) {
            hipEventRecord(e0);
            if (v) hipLaunchKernelGGL(k_r14<true>, dim3(blocks), dim3(256), 0, 0, out, 500);
            else hipLaunchKernelGGL(k_r14<false>, dim3(blocks), dim3(256), 0, 0, out, 500);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            printf("waves/SIMD %d  integer 14 x 28-bit product, %s: %.0f cycles per wave-product per SIMD\n", bpc,
                   v ? "one asm chain per column" : "compiler-scheduled (two chains)", (pr.multiProcessorCount * 4.0 * clk) * (ms * 1e-3) / ((double)blocks * 4 * 500 * 2));
        }
        {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_fma_product, dim3(blocks), dim3(256), 0, 0, out, 200);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double prods = (double)blocks * 4 * 200;
            printf("waves/SIMD %d  FMA formulation, 8 x 52-bit: %.0f cycles per wave-product per SIMD (integer 14 x 28-bit: 2270-2310)\n", bpc,
                   (pr.multiProcessorCount * 4.0 * clk) * (ms * 1e-3) / prods);
        }
    }
    return 0;
}
