// dev probe (round 4): when do the branches of a captured multi-branch hipGraph start on the GPU?
// A root event forks B branches of K dependent "spin" kernels (each spins `us` microseconds on one small workgroup, or -- big=1 --
// on enough workgroups to fill every SIMD slot) and joins them; every kernel stores the wall clock (s_memrealtime, 100 MHz) at
// which its first wavefront began. Printed: start of every kernel relative to the first, per branch; graph replay against the
// same launches issued eagerly. hipcc --offload-arch=gfx950 -O2 -o /tmp/gb tools/ubench_graph_branches.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__global__ void spin(unsigned long long *t, int slot, unsigned ticks) {
    const unsigned long long t0 = __builtin_readcyclecounter() * 0 + wall_clock64();
    if (blockIdx.x == 0 && threadIdx.x == 0) t[slot] = t0;
    while (wall_clock64() - t0 < ticks) {}
    if (blockIdx.x == 0 && threadIdx.x == 0) t[slot + 4096] = wall_clock64();
}
int main(int argc, char **argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 4, K = argc > 2 ? atoi(argv[2]) : 8, us = argc > 3 ? atoi(argv[3]) : 30, big = argc > 4 ? atoi(argv[4]) : 0;
    const int prio = argc > 5 ? atoi(argv[5]) : 1;
    int lo = 0, hi = 0;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    std::vector<hipStream_t> st(B + 1);
    for (auto &s : st) CK(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, prio ? hi : lo));
    std::vector<hipEvent_t> ev(B + 1);
    for (auto &e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    unsigned long long *d_t, h_t[8192];
    CK(hipMalloc(&d_t, sizeof(h_t)));
    const dim3 grid(big ? 256 * 8 : 1), blk(big ? 256 : 64);
    auto enqueue = [&]() {
        CK(hipEventRecord(ev[B], st[B]));
        for (int b = 0; b < B; ++b) {
            CK(hipStreamWaitEvent(st[b], ev[B], 0));
            for (int k = 0; k < K; ++k) hipLaunchKernelGGL(spin, grid, blk, 0, st[b], d_t, b * K + k, (unsigned)(us * 100));
            CK(hipEventRecord(ev[b], st[b]));
        }
        for (int b = 0; b < B; ++b) CK(hipStreamWaitEvent(st[B], ev[b], 0));
    };
    auto report = [&](const char *what) {
        CK(hipMemcpy(h_t, d_t, sizeof(h_t), hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull, t1 = 0;
        for (int i = 0; i < B * K; ++i) t0 = h_t[i] < t0 ? h_t[i] : t0, t1 = h_t[i + 4096] > t1 ? h_t[i + 4096] : t1;
        printf("%s: B=%d K=%d us=%d big=%d prio=%d  total %.1f us (ideal %d)\n", what, B, K, us, big, prio, (t1 - t0) / 100.0, K * us);
        for (int b = 0; b < B; ++b) {
            printf("  branch %d starts:", b);
            for (int k = 0; k < K; ++k) printf(" %7.1f", (h_t[b * K + k] - t0) / 100.0);
            printf("\n");
        }
    };
    // eager
    for (int rep = 0; rep < 3; ++rep) {
        enqueue();
        CK(hipStreamSynchronize(st[B]));
    }
    report("eager");
    // graph
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st[B], hipStreamCaptureModeThreadLocal));
    enqueue();
    CK(hipStreamEndCapture(st[B], &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipGraphLaunch(ge, st[B]));
        CK(hipStreamSynchronize(st[B]));
    }
    report("graph");
    return 0;
}
