// Round 4 probe: cycles per 381-bit Montgomery product at the accumulate kernel's occupancy for
//   A  14 x 28-bit limbs, compiler-scheduled (two accumulator chains)        -- 2 K^2 + K = 406 multiply-adds
//   B  14 x 28-bit limbs, one asm chain per column (what the kernel ships)   -- 406
//   C  13 x 30-bit limbs, one chain, the column accumulator flushed once in the nine columns that hold more than sixteen
//      60-bit products (26 products of 60 bits do not fit 64)                -- 2 K^2 + K = 351 multiply-adds, 9 flushes
//   D  13 x 30-bit limbs with the a*b and m*p sums of every column in two accumulators joined per column -- 351
// Same harness as tools/ubench4.hip: a dependent chain of products per lane, 1 / 2 / 4 wavefronts per SIMD on every CU.
// (Timing only: the modulus limbs are placeholders; the instruction stream is the real one.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint64_t u64;
typedef uint32_t u32;
__device__ __forceinline__ void mad_vv(u64 &acc, u32 a, u32 b) { asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b) : "vcc"); }
__device__ __forceinline__ void mad_vs(u64 &acc, u32 a, u32 k) { asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "s"(k) : "vcc"); }
static constexpr u32 PL[14] = {0x0ffaaab, 0x9feffff, 0x153ffff, 0xeabfffe, 0x6b0f624, 0x730d2a0, 0x38512bf, 0x4774b84, 0x34bacd7, 0xb1ba7b6, 0x97fe69a, 0xa0111ea, 0x1234567, 0x0abcdef};
static constexpr u32 PL30[13] = {0x3ffaaab, 0x29feffff, 0x3153ffff, 0x2eabfffe, 0x16b0f624, 0x3730d2a0, 0x238512bf, 0x14774b84, 0x334bacd7, 0x2b1ba7b6, 0x197fe69a, 0x3a0111ea, 0x00012345};

// MODE 0: compiler (A), 1: asm chain (B)
template <int K, int LB, int MODE> struct Rk {
    u32 v[K];
    static __device__ __forceinline__ u32 P(int i) { return K == 14 ? PL[i] : PL30[i]; }
    static __device__ __forceinline__ Rk mul(const Rk &a, const Rk &b) {
        constexpr u32 MASK = (1u << LB) - 1, INV = 0x12345677u;
        // a column holds cnt(k) = number of limb products; with 30-bit limbs more than 16 of them overflow 64 bits: flush before
        u64 acc = 0, spill = 0;
        u32 m[K];
        Rk t;
#pragma unroll
        for (int k = 0; k < 2 * K - 1; ++k) {
            const int lo = k < K ? 0 : k - K + 1, hi = k < K ? k : K - 1;
            const int nab = hi - lo + 1;                  // a*b products in this column
            const bool flush = LB == 30 && 2 * nab > 16;  // (compile time after unrolling)
#pragma unroll
            for (int i = lo; i <= hi; ++i) {
                if (MODE == 1) mad_vv(acc, a.v[i], b.v[k - i]);
                else acc += (u64)a.v[i] * b.v[k - i];
            }
            if (flush) { // the a*b half is in: move its upper part aside
                spill = acc >> LB;
                acc &= MASK;
            }
#pragma unroll
            for (int i = lo; i <= hi; ++i) {
                if (k < K && i == k) continue; // m[k] is not known yet
                if (MODE == 1) mad_vs(acc, m[i], P(k - i));
                else acc += (u64)m[i] * P(k - i);
            }
            if (k < K) {
                m[k] = ((u32)acc * INV) & MASK;
                if (MODE == 1) mad_vs(acc, m[k], P(0));
                else acc += (u64)m[k] * P(0);
            } else {
                t.v[k - K] = (u32)acc & MASK;
            }
            acc >>= LB;
            if (flush) acc += spill;
        }
        t.v[K - 1] = (u32)acc;
        return t;
    }
    // D: two accumulators per column (a*b | m*p), joined at the end of the column
    static __device__ __forceinline__ Rk mul2(const Rk &a, const Rk &b) {
        constexpr u32 MASK = (1u << LB) - 1, INV = 0x12345677u;
        u64 carry = 0;
        u32 m[K];
        Rk t;
#pragma unroll
        for (int k = 0; k < 2 * K - 1; ++k) {
            const int lo = k < K ? 0 : k - K + 1, hi = k < K ? k : K - 1;
            u64 s1 = carry, s2 = 0;
#pragma unroll
            for (int i = lo; i <= hi; ++i) mad_vv(s1, a.v[i], b.v[k - i]);
#pragma unroll
            for (int i = lo; i <= hi; ++i) {
                if (k < K && i == k) continue;
                mad_vs(s2, m[i], P(k - i));
            }
            u32 low = ((u32)s1 & MASK) + ((u32)s2 & MASK); // < 2^31
            if (k < K) {
                m[k] = (low * INV) & MASK;
                mad_vs(s2, m[k], P(0));
                low = ((u32)s1 & MASK) + ((u32)s2 & MASK);
            } else {
                t.v[k - K] = low & MASK;
            }
            carry = (s1 >> LB) + (s2 >> LB) + (low >> LB);
        }
        t.v[K - 1] = (u32)carry;
        return t;
    }
};
template <int K, int LB, int MODE> __global__ __launch_bounds__(256) void k_prod(u64 *out, int iters) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    typedef Rk<K, LB, MODE == 2 ? 1 : MODE> R;
    R a, b;
    for (int i = 0; i < K; ++i) a.v[i] = (t * 2654435761u + i * 977) & ((1u << LB) - 1), b.v[i] = (t * 40503u + i * 131) & ((1u << LB) - 1);
    for (int i = 0; i < iters; ++i) {
        if (MODE == 2) a = R::mul2(a, b), b = R::mul2(b, a);
        else a = R::mul(a, b), b = R::mul(b, a);
    }
    u64 x = 0;
    for (int i = 0; i < K; ++i) x ^= (u64)a.v[i] << (i & 31);
    out[t] = x;
}
int main() {
    hipDeviceProp_t pr;
    hipGetDeviceProperties(&pr, 0);
    const double clk = pr.clockRate * 1e3;
    u64 *out;
    hipMalloc(&out, (size_t)pr.multiProcessorCount * 8 * 256 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    printf("# cycles per wave-product per SIMD at the device's nominal %.0f MHz (the kernel's own clock under this load is ~2.0-2.1 GHz: compare rows)\n", clk / 1e6);
    for (int bpc : {1, 2, 4}) {
        const int blocks = pr.multiProcessorCount * bpc;
        auto run = [&](const char *name, auto kern) {
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 4);
            hipDeviceSynchronize();
            float best = 1e9f;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 500);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                best = ms < best ? ms : best;
            }
            printf("waves/SIMD %d  %-78s %6.0f cycles per wave-product per SIMD\n", bpc, name,
                   (pr.multiProcessorCount * 4.0 * clk) * (best * 1e-3) / ((double)blocks * 4 * 500 * 2));
        };
        run("A 14 x 28, compiler-scheduled (406 multiply-adds)", k_prod<14, 28, 0>);
        run("B 14 x 28, one asm chain per column (406) -- shipped", k_prod<14, 28, 1>);
        run("C 13 x 30, one chain, 9 columns flushed mid-column (351)", k_prod<13, 30, 1>);
        run("C' 13 x 30, compiler-scheduled, 9 columns flushed (351)", k_prod<13, 30, 0>);
        run("D 13 x 30, a*b and m*p in two accumulators per column (351)", k_prod<13, 30, 2>);
    }
    return 0;
}
