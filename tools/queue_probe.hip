// dev tool (round 5): which HIP streams share a hardware queue? Streams are created in a fixed order (normal and high priority), a
// ~200 us spin kernel of ONE workgroup is launched on every pair, and the pair's wall time says whether the two ran side by side
// (~200 us) or one behind the other (~400 us). Prints the matrix: '.' = concurrent, 'S' = serialised.
// build: hipcc --offload-arch=gfx950 -O2 tools/queue_probe.hip -o tools/bin/queue_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void spin(long long cycles, unsigned *sink) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {
    }
    if (sink && threadIdx.x == 1234567) *sink = 1;
}
int main(int argc, char **argv) {
    const int nn = argc > 1 ? atoi(argv[1]) : 8, nh = argc > 2 ? atoi(argv[2]) : 4;
    int lo = 0, hi = 0;
    hipDeviceGetStreamPriorityRange(&lo, &hi);
    printf("priority range: least %d greatest %d\n", lo, hi);
    std::vector<hipStream_t> st;
    std::vector<char> kind;
    for (int i = 0; i < nn; ++i) {
        hipStream_t s;
        hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        st.push_back(s), kind.push_back('n');
    }
    for (int i = 0; i < nh; ++i) {
        hipStream_t s;
        hipStreamCreateWithPriority(&s, hipStreamNonBlocking, hi);
        st.push_back(s), kind.push_back('h');
    }
    const long long cyc = 20000; // wall_clock64 ticks at 100 MHz: 200 us
    for (auto s : st) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, 1000, nullptr);
    hipDeviceSynchronize();
    const int n = (int)st.size();
    printf("      ");
    for (int j = 0; j < n; ++j) printf("%c%-2d", kind[j], j);
    printf("\n");
    for (int i = 0; i < n; ++i) {
        printf("%c%-2d   ", kind[i], i);
        for (int j = 0; j < n; ++j) {
            if (j == i) {
                printf(" - ");
                continue;
            }
            hipDeviceSynchronize();
            auto t0 = std::chrono::steady_clock::now();
            hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st[i], cyc, nullptr);
            hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st[j], cyc, nullptr);
            hipStreamSynchronize(st[i]);
            hipStreamSynchronize(st[j]);
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            printf(" %c ", us > 330 ? 'S' : '.');
        }
        printf("\n");
    }
    // how many spin kernels run side by side when every stream gets one?
    for (int k = 2; k <= n; ++k) {
        hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < k; ++i) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st[i], cyc, nullptr);
        hipDeviceSynchronize();
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        printf("first %2d streams, one kernel each: %.0f us\n", k, us);
    }
    return 0;
}
