// Does straight-line code size limit per-wave issue rate on gfx950? Same instruction mix, loop bodies of
// 8 .. 2048 instructions; 1/2/4/8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32; typedef uint64_t u64;
#define I4 "v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3\n"
#define I8 I4 I4
#define I32 I8 I8 I8 I8
#define I128 I32 I32 I32 I32
#define I512 I128 I128 I128 I128
#define I2048 I512 I512 I512 I512
#define I8192 I2048 I2048 I2048 I2048
#define I16384 I8192 I8192
template <int BODY> __global__ __launch_bounds__(256) void k(u32 *out, int total) {
    u32 a = threadIdx.x * 2654435761u + 12345, b = blockIdx.x * 40503u + 977;
    u64 c0 = a, c1 = b, c2 = a ^ b, c3 = a + b;
    const int iters = total / BODY;
    for (int i = 0; i < iters; ++i) {
        if (BODY == 8) asm volatile(I8 : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b) : "vcc");
        if (BODY == 32) asm volatile(I32 : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b) : "vcc");
        if (BODY == 128) asm volatile(I128 : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b) : "vcc");
        if (BODY == 512) asm volatile(I512 : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b) : "vcc");
        if (BODY == 2048) asm volatile(I2048 : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b) : "vcc");
        if (BODY == 8192) asm volatile(I8192 : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b) : "vcc");
        if (BODY == 16384) asm volatile(I16384 : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b) : "vcc");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (u32)(c0 ^ c1 ^ c2 ^ c3);
}
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const double clk = p.clockRate * 1e3;
    u32 *out; hipMalloc(&out, (size_t)p.multiProcessorCount * 8 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int total = 1 << 20;
    for (int bpc = 1; bpc <= 8; bpc *= 2) {
        const int blocks = p.multiProcessorCount * bpc;
        for (int body : {512, 2048, 8192, 16384}) {
            auto launch = [&]() {
                switch (body) {
                case 8: hipLaunchKernelGGL(k<8>, dim3(blocks), dim3(256), 0, 0, out, total); break;
                case 32: hipLaunchKernelGGL(k<32>, dim3(blocks), dim3(256), 0, 0, out, total); break;
                case 128: hipLaunchKernelGGL(k<128>, dim3(blocks), dim3(256), 0, 0, out, total); break;
                case 512: hipLaunchKernelGGL(k<512>, dim3(blocks), dim3(256), 0, 0, out, total); break;
                case 2048: hipLaunchKernelGGL(k<2048>, dim3(blocks), dim3(256), 0, 0, out, total); break;
                case 8192: hipLaunchKernelGGL(k<8192>, dim3(blocks), dim3(256), 0, 0, out, total); break;
                case 16384: hipLaunchKernelGGL(k<16384>, dim3(blocks), dim3(256), 0, 0, out, total); break;
                }
            };
            launch(); hipDeviceSynchronize();
            hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double per_wave_cycles = ms * 1e-3 * clk / total;
            printf("waves/SIMD %d body %4d instrs (%5d B): %7.3f ms  %.2f cycles/instr per wave, %.2f cycles/wave-instr per SIMD\n", bpc, body, body * 8, ms, per_wave_cycles, per_wave_cycles / bpc);
        }
    }
    return 0;
}
