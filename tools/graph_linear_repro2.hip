// Stand-alone probe #2 (round 5): a closer mimic of the library's linear part-A graph -- several memset nodes, kernels and D2H copy nodes
// to pinned memory in ONE linear capture on a high-priority stream, two eager runs first, an H2D copy on the stream before every replay,
// and between replays what tools/diag_linear.py's checksum pass does (hipDeviceSynchronize + a few dozen synchronous pageable D2H copies).
// Checks after every replay that the big memset really zeroed its buffer (the graph's own kernels leave rows [0, rows) = 1, the rest 0).
//   build: hipcc --offload-arch=gfx950 -O2 -o tools/bin/graph_linear_repro2 tools/graph_linear_repro2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("ERR %s line %d: %s\n", #x, __LINE__, hipGetErrorString(e_)); exit(2);} } while (0)
__global__ void fill(unsigned *p, unsigned rows, const unsigned *z) {
    unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < rows) p[i] = 1u + (z[i % 64] & 0u);
}
__global__ void count_nz(const unsigned *p, unsigned n, unsigned *cnt) {
    unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && p[i]) atomicAdd(cnt, 1u);
}
__global__ void touch(unsigned *p, unsigned n) {
    unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] += 1u;
}
int main(int argc, char **argv) {
    const unsigned n = argc > 1 ? (unsigned)atoi(argv[1]) : 27648u, rows = n / 3;
    const int n_fill_nodes = argc > 2 ? atoi(argv[2]) : 4;
    unsigned *buf, *cnt, *small[8], *z, *h_z, *h_out;
    CK(hipMalloc(&buf, n * 4));
    CK(hipMalloc(&cnt, 256));
    CK(hipMalloc(&z, 16384));
    for (auto &s : small) CK(hipMalloc(&s, 4096));
    CK(hipHostMalloc((void **)&h_z, 16384, hipHostMallocDefault));
    CK(hipHostMalloc((void **)&h_out, 4096, hipHostMallocDefault));
    for (int i = 0; i < 4096; ++i) h_z[i] = i;
    hipStream_t s;
    int lo, hi;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    CK(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, hi));
    auto body = [&] {
        CK(hipMemsetAsync(buf, 0, n * 4, s));
        hipLaunchKernelGGL(fill, dim3((rows + 255) / 256), dim3(256), 0, s, buf, rows, z);
        CK(hipMemsetAsync(cnt, 0, 4, s));
        hipLaunchKernelGGL(count_nz, dim3((n + 255) / 256), dim3(256), 0, s, buf, n, cnt);
        for (int k = 0; k < n_fill_nodes; ++k) {
            CK(hipMemsetAsync(small[k % 8], 0, k % 2 ? 576 : 144, s));
            hipLaunchKernelGGL(touch, dim3(1), dim3(64), 0, s, small[k % 8], 36u);
            CK(hipMemcpyAsync(h_out + 64 * (k % 8), small[k % 8], 144, hipMemcpyDeviceToHost, s));
        }
        CK(hipMemcpyAsync(h_out + 1000, cnt, 4, hipMemcpyDeviceToHost, s));
    };
    for (int r = 0; r < 2; ++r) { // two eager runs, like a proof slot
        CK(hipMemcpyAsync(z, h_z, 16000, hipMemcpyHostToDevice, s));
        body();
        CK(hipStreamSynchronize(s));
    }
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    body();
    hipGraph_t g;
    hipGraphExec_t ex;
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
    CK(hipGraphDestroy(g));
    std::vector<unsigned> host(n);
    auto replay = [&](const char *what) {
        CK(hipMemcpyAsync(z, h_z, 16000, hipMemcpyHostToDevice, s));
        h_out[1000] = 0xdeadbeef;
        CK(hipGraphLaunch(ex, s));
        CK(hipStreamSynchronize(s));
        const unsigned got = h_out[1000];
        bool small_ok = true;
        for (int k = 0; k < n_fill_nodes && k < 8; ++k)
            for (int i = 0; i < 36; ++i) small_ok = small_ok && h_out[64 * k + i] == 1u;
        printf("%-60s non-zero words after memset + fill: %u (expected %u) %s; small buffers %s\n", what, got, rows, got == rows ? "ok" : "BAD",
               small_ok ? "ok" : "BAD");
        fflush(stdout);
    };
    auto checksum_pass = [&] { // what mg_diag_slot_sums does
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(host.data(), z, 16000, hipMemcpyDeviceToHost));
        for (int r = 0; r < 3; ++r) CK(hipMemcpy(host.data(), buf + r * (n / 3), n / 3 * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(host.data(), buf, n * 4, hipMemcpyDeviceToHost));
        for (int k = 0; k < 30; ++k) CK(hipMemcpy(host.data(), small[k % 8], k % 3 ? 4096 : 4, hipMemcpyDeviceToHost));
    };
    printf("n = %u words, %d small memset / kernel / D2H groups; DEBUG_CLR_GRAPH_PACKET_CAPTURE=%s\n", n, n_fill_nodes,
           getenv("DEBUG_CLR_GRAPH_PACKET_CAPTURE") ? getenv("DEBUG_CLR_GRAPH_PACKET_CAPTURE") : "default");
    replay("first replay");
    replay("second replay, nothing in between");
    checksum_pass();
    replay("after a checksum pass (sync + pageable D2H copies)");
    checksum_pass();
    replay("after another");
    {
        unsigned *big;
        CK(hipMalloc(&big, 256 << 20));
        CK(hipMemset(big, 0x77, 256 << 20));
        CK(hipDeviceSynchronize());
        replay("after hipMalloc + hipMemset(0x77) of 256 MB on the null stream");
        CK(hipFree(big));
        replay("after hipFree");
    }
    for (int i = 0; i < 20; ++i) replay("again");
    return 0;
}
