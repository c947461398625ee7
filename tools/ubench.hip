// Instruction-throughput microbenchmarks for the integer paths a 256/384-bit Montgomery product can
// be built from on gfx950. Prints wave-instructions/s and the implied cycles per wave-instruction per
// SIMD (1024 SIMDs; clock from hipDeviceProp). Build: hipcc --offload-arch=gfx950 -O3 tools/ubench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "../manta_rs_amd/csrc/params_gen.h"
#include "../manta_rs_amd/csrc/fp_dev.h"
using namespace mg;
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int MODE> __global__ __launch_bounds__(256) void k_instr(u32 *out, int iters) {
    u32 a = threadIdx.x * 2654435761u + 12345, b = blockIdx.x * 40503u + 977;
    u64 c0 = a, c1 = b, c2 = a ^ b, c3 = a + b;
    u32 d0 = a, d1 = b, d2 = a ^ 7, d3 = b ^ 9;
    double f0 = a, f1 = b, f2 = 1.5, f3 = 2.5, fa = 1.0000001, fb = 0.5;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) { REP64(asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b) : "vcc");) }
        if (MODE == 1) { REP64(asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b) : "vcc");) }
        if (MODE == 2) { REP64(asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a));) }
        if (MODE == 3) { REP64(asm volatile("v_mul_hi_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_hi_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a));) }
        if (MODE == 4) { REP64(asm volatile("v_add_co_u32 %0, vcc, %0, %4\n v_addc_co_u32 %1, vcc, %1, %4, vcc\n v_addc_co_u32 %2, vcc, %2, %4, vcc\n v_addc_co_u32 %3, vcc, %3, %4, vcc" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a) : "vcc");) }
        if (MODE == 5) { REP64(asm volatile("v_mad_u32_u24 %0, %0, %4, %5\n v_mad_u32_u24 %1, %1, %4, %5\n v_mad_u32_u24 %2, %2, %4, %5\n v_mad_u32_u24 %3, %3, %4, %5" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a), "v"(b));) }
        if (MODE == 6) { REP64(asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(fa), "v"(fb));) }
        if (MODE == 7) { REP64(asm volatile("v_lshl_add_u64 %0, %0, 0, %4\n v_lshl_add_u64 %1, %1, 0, %4\n v_lshl_add_u64 %2, %2, 0, %4\n v_lshl_add_u64 %3, %3, 0, %4" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(c3));) }
        if (MODE == 8) { REP64(asm volatile("v_mul_u32_u24 %0, %0, %4\n v_mul_hi_u32_u24 %1, %1, %4\n v_mul_u32_u24 %2, %2, %4\n v_mul_hi_u32_u24 %3, %3, %4" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a));) }
        if (MODE == 9) { REP64(asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_addc_co_u32 %2, vcc, 0, %2, vcc\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_addc_co_u32 %3, vcc, 0, %3, vcc" : "+v"(c0), "+v"(c1), "+v"(d2), "+v"(d3) : "v"(a), "v"(b) : "vcc");) }
        if (MODE == 10) { REP64(asm volatile("v_dot4_u32_u8 %0, %4, %5, %0\n v_dot4_u32_u8 %1, %4, %5, %1\n v_dot4_u32_u8 %2, %4, %5, %2\n v_dot4_u32_u8 %3, %4, %5, %3" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a), "v"(b));) }
        if (MODE == 12) { REP64(asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a));) }
        if (MODE == 13) { float g0 = __uint_as_float(d0), g1 = __uint_as_float(d1), g2 = __uint_as_float(d2), g3 = __uint_as_float(d3), ga = 1.0001f; REP64(asm volatile("v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4" : "+v"(g0), "+v"(g1), "+v"(g2), "+v"(g3) : "v"(ga));) d0 = __float_as_uint(g0); d1 = __float_as_uint(g1); d2 = __float_as_uint(g2); d3 = __float_as_uint(g3); }
        if (MODE == 14) { REP64(asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_addc_co_u32 %2, vcc, 0, %2, vcc\n v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_addc_co_u32 %2, vcc, 0, %2, vcc" : "+v"(c0), "+v"(c1), "+v"(d2), "+v"(d3) : "v"(a), "v"(b) : "vcc");) }
        if (MODE == 11) { REP64(asm volatile("v_mad_i32_i24 %0, %0, %4, %5\n v_mad_i32_i24 %1, %1, %4, %5\n v_mad_i32_i24 %2, %2, %4, %5\n v_mad_i32_i24 %3, %3, %4, %5" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a), "v"(b));) }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (u32)(c0 ^ c1 ^ c2 ^ c3) ^ d0 ^ d1 ^ d2 ^ d3 ^ (u32)(f0 + f1 + f2 + f3);
}

template <class C, bool INL> __global__ __launch_bounds__(256) void k_fpmul(u32 *out, const u32 *in, int iters) {
    typedef Fp<C> F;
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    F a = F::load(in + (size_t)(t % 1024) * F::N), b = F::load(in + (size_t)((t + 1) % 1024) * F::N);
    for (int i = 0; i < iters; ++i) { a = F::mul(a, b); b = F::mul(b, a); }
    a.store(out + (size_t)t * F::N);
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const double clk = p.clockRate * 1e3; // Hz
    printf("device %s CUs %d clock %.0f MHz\n", p.name, p.multiProcessorCount, clk / 1e6);
    int blocks = p.multiProcessorCount * 8; const int threads = 256;
    u32 *out; hipMalloc(&out, (size_t)blocks * threads * 12 * 4 * 2);
    u32 *in; hipMalloc(&in, 1024 * 12 * 4);
    std::vector<u32> h(1024 * 12); for (size_t i = 0; i < h.size(); ++i) h[i] = (u32)(i * 2654435761u) >> 3; hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char *names[] = {"v_mad_u64_u32 (4 indep)", "v_mad_u64_u32 (dependent)", "v_mul_lo_u32", "v_mul_hi_u32", "v_add_co/addc chain", "v_mad_u32_u24", "v_fma_f64", "v_lshl_add_u64", "v_mul_u32_u24 + mul_hi_u32_u24", "mad_u64_u32 + addc (2 chains)", "v_dot4_u32_u8", "v_mad_i32_i24", "v_add_u32", "v_fma_f32", "mad_u64_u32 + addc (1 chain, as in Fp::mul)"};
    auto run = [&](int mode, int iters) {
        switch (mode) {
#define CASE(M) case M: hipLaunchKernelGGL((k_instr<M>), dim3(blocks), dim3(threads), 0, 0, out, iters); break;
        CASE(0) CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9) CASE(10) CASE(11) CASE(12) CASE(13) CASE(14)
        }
    };
    for (int bpc = 8; bpc >= 1; bpc /= 2) { blocks = p.multiProcessorCount * bpc; printf("--- %d waves per SIMD\n", bpc); for (int mode = 0; mode < 15; ++mode) {
        const int iters = 3000;
        run(mode, 2); hipDeviceSynchronize();
        hipEventRecord(e0); run(mode, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double winstr = (double)blocks * (threads / 64) * iters * 64.0 * 4.0; // wave-instructions
        const double rate = winstr / (ms * 1e-3);
        printf("%-34s %8.3f ms  %8.2f G wave-instr/s  -> %.2f cycles/wave-instr/SIMD\n", names[mode], ms, rate / 1e9, (p.multiProcessorCount * 4.0 * clk) / rate);
    } }
    blocks = p.multiProcessorCount * 8;
    auto fp = [&](const char *name, auto kern, int nlimbs, int iters) {
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, in, 2); hipDeviceSynchronize();
        hipEventRecord(e0); hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, in, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double muls = (double)blocks * threads * iters * 2.0;
        printf("%-34s %8.3f ms  %8.2f G mont-mul/s  (%.0f cycles/wave-mul/SIMD, %d mads)\n", name, ms, muls / (ms * 1e-3) / 1e9, (p.multiProcessorCount * 4.0 * clk) / (muls / 64 / (ms * 1e-3)), 2 * nlimbs * nlimbs + nlimbs);
    };
    fp("Fp<Bn254Fq>::mul (call)", k_fpmul<Bn254FqCfg, false>, 8, 2000);
    fp("Fp<Bls381Fq>::mul (call)", k_fpmul<Bls381FqCfg, false>, 12, 1000);
    return 0;
}
