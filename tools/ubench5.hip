// Issue interval of the NON-multiply-add opcodes of the accumulate kernel's body (a third of its instructions):
// v_mul_lo_u32 (the Montgomery quotient digit m = t * (-1/p) mod 2^LB), v_and_b32, v_lshrrev_b64, v_lshl_add_u64, v_mov_b32,
// next to v_mad_u64_u32, in eight independent chains per lane. Decides whether the quotient digit should be a
// v_mad_u64_u32 (low half) instead of a v_mul_lo_u32.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint64_t u64;
typedef uint32_t u32;
template <int MODE> __global__ __launch_bounds__(256) void k_rate(u64 *out, int iters) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    u64 q[8];
    u32 w[8];
    for (int i = 0; i < 8; ++i) q[i] = t * 977 + i, w[i] = t * 31 + i;
    const u32 k = 0x12345677u + t;
    for (int it = 0; it < iters; it += 32) {
#pragma unroll
        for (int u = 0; u < 256; ++u) {
            const int i = u & 7;
            if (MODE == 0) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q[i]) : "v"(w[i]), "v"(k) : "vcc");
            if (MODE == 1) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(w[i]) : "v"(k));
            if (MODE == 2) asm volatile("v_and_b32 %0, %0, %1" : "+v"(w[i]) : "v"(k));
            if (MODE == 3) asm volatile("v_lshrrev_b64 %0, 1, %0" : "+v"(q[i]));
            if (MODE == 4) asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(q[i]) : "v"(q[(i + 1) & 7]));
            if (MODE == 5) asm volatile("v_mov_b32 %0, %1" : "+v"(w[i]) : "v"(k));
            if (MODE == 6) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(w[i]) : "v"(k));
            if (MODE == 7) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(w[i]) : "v"(k));
            if (MODE == 8) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(w[i]) : "v"(k));
            if (MODE == 9) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(w[i]) : "v"(k));
        }
    }
    u64 x = 0;
    for (int i = 0; i < 8; ++i) x ^= q[i] ^ w[i];
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
    const char *names[10] = {"v_mad_u64_u32", "v_mul_lo_u32", "v_and_b32", "v_lshrrev_b64", "v_lshl_add_u64", "v_mov_b32", "v_mul_hi_u32", "v_mul_u32_u24", "v_add3_u32", "v_mad_u32_u24"};
    for (int bpc : {1, 2}) {
        const int blocks = pr.multiProcessorCount * bpc;
        auto run = [&](const char *name, auto kern) {
            const int iters = 200000;
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 2);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double wave_ops = (double)blocks * 4 * iters * 8;
            printf("waves/SIMD %d  %-16s %8.3f ms  %6.2f cycles per wave-instruction per SIMD\n", bpc, name, ms,
                   (pr.multiProcessorCount * 4.0 * clk) * (ms * 1e-3) / wave_ops);
        };
        run(names[0], k_rate<0>); run(names[1], k_rate<1>); run(names[2], k_rate<2>); run(names[3], k_rate<3>); run(names[4], k_rate<4>);
        run(names[5], k_rate<5>); run(names[6], k_rate<6>); run(names[7], k_rate<7>); run(names[8], k_rate<8>); run(names[9], k_rate<9>);
    }
    return 0;
}
