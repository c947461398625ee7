// dev tool (round 6): does hipExtStreamCreateWithCUMask work on this box, and which physical CUs does bit b of the mask name?
// A grid of short spin workgroups records (XCC_ID, SE, SH, CU) per workgroup; run once on a plain stream and once per mask.
// Also: a "fat" kernel (2 waves per SIMD worth of registers is emulated by launching exactly 8 waves per CU that spin) on the masked
// stream next to small workgroups on a plain stream -- where do the small ones land and how long do they wait?
// build: hipcc --offload-arch=gfx950 -O2 tools/cumask_probe.hip -o tools/bin/cumask_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void where(unsigned *out, long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {
    }
    if (threadIdx.x == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));   // HW_REG_HW_ID, all 32 bits
        const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));  // HW_REG_XCC_ID[3:0]
        out[blockIdx.x] = (xcc << 16) | (hw & 0xffffu);
    }
}
static int cu_key(unsigned v) { // xcc, se (3 bits at 13), sh (bit 12), cu (4 bits at 8)
    const unsigned xcc = v >> 16, hw = v & 0xffff;
    return (int)((xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15));
}
static std::set<int> run(hipStream_t s, unsigned *d, unsigned *h, int wgs) {
    hipLaunchKernelGGL(where, dim3(wgs), dim3(256), 0, s, d, 2000); // 20 us each
    hipStreamSynchronize(s);
    hipMemcpy(h, d, wgs * 4, hipMemcpyDeviceToHost);
    std::set<int> cus;
    for (int i = 0; i < wgs; ++i) cus.insert(cu_key(h[i]));
    return cus;
}
int main(int argc, char **argv) {
    const int R = argc > 1 ? atoi(argv[1]) : 8;
    hipDeviceProp_t pr;
    CK(hipGetDeviceProperties(&pr, 0));
    printf("CUs %d\n", pr.multiProcessorCount);
    const int wgs = 8192;
    unsigned *d, *h = (unsigned *)malloc(wgs * 4);
    CK(hipMalloc(&d, wgs * 4));
    hipStream_t plain;
    CK(hipStreamCreateWithFlags(&plain, hipStreamNonBlocking));
    std::set<int> all = run(plain, d, h, wgs);
    printf("plain stream: %zu distinct (xcc,se,sh,cu)\n", all.size());
    std::map<int, int> per_xcc;
    for (int k : all) per_xcc[k >> 8]++;
    for (auto &kv : per_xcc) printf("  xcc %d: %d CUs\n", kv.first, kv.second);
    for (int variant = 0; variant < 3; ++variant) {
        std::vector<uint32_t> mask(8, 0xffffffffu);
        if (variant == 0) for (int b = 0; b < R; ++b) mask[b / 32] &= ~(1u << (b % 32));                 // low R bits off
        if (variant == 1) for (int b = 0; b < R; ++b) mask[(b * 32) / 32 % 8] &= ~(1u << 0), (void)b;    // bit 0 of every word off
        if (variant == 2) for (int b = 256 - R; b < 256; ++b) mask[b / 32] &= ~(1u << (b % 32));           // high R bits off
        hipStream_t ms;
        hipError_t e = hipExtStreamCreateWithCUMask(&ms, 8, mask.data());
        if (e != hipSuccess) { printf("variant %d: hipExtStreamCreateWithCUMask -> %s\n", variant, hipGetErrorString(e)); continue; }
        std::set<int> got = run(ms, d, h, wgs);
        printf("variant %d (%s): %zu distinct CUs; excluded:", variant, variant == 0 ? "low R bits off" : variant == 1 ? "bit 0 of every word off" : "high R bits off",
               got.size());
        for (int k : all) if (!got.count(k)) printf(" [x%d se%d sh%d cu%d]", k >> 8, (k >> 5) & 7, (k >> 4) & 1, k & 15);
        printf("\n");
        uint32_t back[8] = {};
        if (hipExtStreamGetCUMask(ms, 8, back) == hipSuccess) printf("   mask read back: %08x %08x ... %08x\n", back[0], back[1], back[7]);
        // concurrency: a chip-filling spin (2048 workgroups of 256 = 8 waves per CU... on the masked stream, 3 ms) + small kernel on the plain stream
        if (variant == 0) {
            hipDeviceSynchronize();
            hipLaunchKernelGGL(where, dim3(8 * 256), dim3(256), 0, ms, d, 300000); // 3 ms; 2048 x 4 waves: 8 waves per SIMD
            auto t0 = std::chrono::steady_clock::now();
            hipLaunchKernelGGL(where, dim3(64), dim3(256), 0, plain, d + 4096, 1000);
            hipStreamSynchronize(plain);
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            hipMemcpy(h, d + 4096, 64 * 4, hipMemcpyDeviceToHost);
            std::set<int> small;
            for (int i = 0; i < 64; ++i) small.insert(cu_key(h[i]));
            int in_reserved = 0;
            for (int k : small) in_reserved += !got.count(k);
            printf("   64 small workgroups beside the masked chip-filler: %.0f us, on %zu CUs of which %d are reserved ones\n", us, small.size(), in_reserved);
            hipDeviceSynchronize();
        }
    }
    return 0;
}
