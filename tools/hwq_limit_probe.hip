// dev tool (round 6): how many hardware (HSA) queues of ONE process run side by side on this GPU? Every stream below has a queue of
// its own (hipExtStreamCreateWithCUMask, full mask). k one-workgroup spin kernels of 500 us, one per stream, launched together:
// wall time ~500 us while the k queues run concurrently, a multiple once the scheduler has to take turns.
// build: hipcc --offload-arch=gfx950 -O2 tools/hwq_limit_probe.hip -o tools/bin/hwq_limit_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void spin(long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {
    }
}
int main() {
    const int N = 24;
    std::vector<uint32_t> mask(8, 0xffffffffu);
    std::vector<hipStream_t> ordinary, st;  // (`ordinary`: only created, to fill the runtime's pool)
    for (int i = 0; i < 8; ++i) { // fill the runtime's shared pool first (see profiles/r06_pipeline_phase.txt)
        hipStream_t s;
        if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return 1;
        ordinary.push_back(s);
    }
    for (int i = 0; i < N; ++i) {
        hipStream_t s;
        if (hipExtStreamCreateWithCUMask(&s, 8, mask.data()) != hipSuccess) { printf("stream %d refused\n", i); break; }
        st.push_back(s);
    }
    for (auto s : st) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, 1000);
    (void)hipDeviceSynchronize();
    for (int k = 1; k <= (int)st.size(); ++k) {
        double best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipDeviceSynchronize();
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < k; ++i) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st[i], 50000); // 500 us at 100 MHz
            for (int i = 0; i < k; ++i) (void)hipStreamSynchronize(st[i]);
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            if (us < best) best = us;
        }
        printf("k = %2d dedicated queues busy: %7.0f us\n", k, best);
    }
    return 0;
}
