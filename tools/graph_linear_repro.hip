// Stand-alone probe (round 5) for the wrong-C defect of linear captured graphs (profiles/r05_linear_graph_defect.txt):
// does a LINEAR captured graph (one stream: the runtime pre-builds its AQL packets at instantiation,
// DEBUG_CLR_GRAPH_PACKET_CAPTURE) replay its memset / kernel nodes correctly after other work has gone through the runtime?
//   build: hipcc --offload-arch=gfx950 -O2 -o tools/bin/graph_linear_repro tools/graph_linear_repro.hip
//   run:   tools/bin/graph_linear_repro            (and again with DEBUG_CLR_GRAPH_PACKET_CAPTURE=0)
// Graph under test: memset(buf, 0, N) -> fill<<<>>>(buf, rows) [writes 1 into the first `rows` words] -> acc<<<>>>(buf, out)
// [out[i] += buf[i]] ; expected after every replay: out[i] - out_before[i] = (i < rows).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("ERR %s line %d: %s\n", #x, __LINE__, hipGetErrorString(e_)); exit(2);} } while (0)

struct Big { unsigned *p[12]; unsigned n; }; // a by-value struct argument like spmv3's
__global__ void fill(Big b, unsigned rows) {
    unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < rows) b.p[3][i] = 1u;
}
__global__ void acc(const unsigned *buf, unsigned *out, unsigned n) {
    unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] += buf[i];
}
__global__ void scribble(unsigned *p, unsigned n, unsigned v) {
    unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

struct Test {
    size_t n; // words
    unsigned rows;
    unsigned *buf = nullptr, *out = nullptr;
    hipStream_t s = nullptr;
    hipGraphExec_t ex = nullptr;
    std::vector<unsigned> before, after;
    void build(bool forked) {
        CK(hipMalloc(&buf, n * 4));
        CK(hipMalloc(&out, n * 4));
        CK(hipMemset(out, 0, n * 4));
        CK(hipMemset(buf, 0xff, n * 4));
        if (getenv("REPRO_PRIO")) {
            int lo, hi;
            CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
            CK(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, hi));
        } else
            CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        if (getenv("REPRO_EAGER")) // the library runs a slot twice with plain launches before it captures
            for (int r = 0; r < 2; ++r) {
                CK(hipMemsetAsync(buf, 0, n * 4, s));
                Big b0{};
                b0.p[3] = buf;
                hipLaunchKernelGGL(fill, dim3((rows + 255) / 256), dim3(256), 0, s, b0, rows);
                hipLaunchKernelGGL(acc, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, buf, out, (unsigned)n);
                CK(hipStreamSynchronize(s));
            }
        hipStream_t s2 = nullptr;
        hipEvent_t e1, e2;
        CK(hipEventCreateWithFlags(&e1, hipEventDisableTiming));
        CK(hipEventCreateWithFlags(&e2, hipEventDisableTiming));
        if (forked) CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        CK(hipMemsetAsync(buf, 0, n * 4, s));
        Big b{};
        b.p[3] = buf;
        b.n = (unsigned)n;
        hipLaunchKernelGGL(fill, dim3((rows + 255) / 256), dim3(256), 0, s, b, rows);
        if (forked) { // a dummy second branch makes the graph non-linear (the runtime then launches node by node)
            CK(hipEventRecord(e1, s));
            CK(hipStreamWaitEvent(s2, e1, 0));
            hipLaunchKernelGGL(scribble, dim3(1), dim3(64), 0, s2, out + n - 64, 0u, 0u);
            CK(hipEventRecord(e2, s2));
            CK(hipStreamWaitEvent(s, e2, 0));
        }
        hipLaunchKernelGGL(acc, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, buf, out, (unsigned)n);
        hipGraph_t g;
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
        CK(hipGraphDestroy(g));
        before.assign(n, 0);
        after.assign(n, 0);
    }
    // one replay; returns the number of wrong words
    size_t replay() {
        CK(hipMemcpy(before.data(), out, n * 4, hipMemcpyDeviceToHost));
        CK(hipGraphLaunch(ex, s));
        CK(hipStreamSynchronize(s));
        CK(hipMemcpy(after.data(), out, n * 4, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (size_t i = 0; i < n; ++i) bad += (after[i] - before[i]) != (i < rows ? 1u : 0u);
        return bad;
    }
    // the same without any synchronous runtime call between replays (events only)
    size_t replay_quiet(int times) {
        std::vector<unsigned> *h = nullptr;
        (void)h;
        unsigned *pin0 = nullptr, *pin1 = nullptr;
        CK(hipHostMalloc((void **)&pin0, n * 4, hipHostMallocDefault));
        CK(hipHostMalloc((void **)&pin1, n * 4, hipHostMallocDefault));
        size_t bad = 0;
        for (int t = 0; t < times; ++t) {
            CK(hipMemcpyAsync(pin0, out, n * 4, hipMemcpyDeviceToHost, s));
            CK(hipGraphLaunch(ex, s));
            CK(hipMemcpyAsync(pin1, out, n * 4, hipMemcpyDeviceToHost, s));
            CK(hipStreamSynchronize(s));
            for (size_t i = 0; i < n; ++i) bad += (pin1[i] - pin0[i]) != (i < rows ? 1u : 0u);
        }
        CK(hipHostFree(pin0));
        CK(hipHostFree(pin1));
        return bad;
    }
};

int main(int argc, char **argv) {
    const int forked = argc > 1 ? atoi(argv[1]) : 0;
    const size_t sizes[] = {1, 36, 27648, 1769472}; // words: 4 B (a counter), 144 B (a point), 110 KB, 7 MB (a PrivateTransfer a | b | c)
    std::vector<Test> tests;
    for (size_t n : sizes) {
        Test t;
        t.n = n;
        t.rows = (unsigned)(n > 8 ? n / 3 : 1);
        t.build(forked != 0);
        tests.push_back(t);
    }
    auto round = [&](const char *what) {
        printf("%-64s", what);
        for (Test &t : tests) printf(" n=%zu:%s", t.n, t.replay() ? "BAD" : "ok");
        printf("\n");
        fflush(stdout);
    };
    printf("graph = memset -> fill -> %sacc, one exec per size; packet capture %s\n", forked ? "(dummy branch) -> " : "",
           getenv("DEBUG_CLR_GRAPH_PACKET_CAPTURE") ? getenv("DEBUG_CLR_GRAPH_PACKET_CAPTURE") : "default");
    round("first replay");
    round("second replay (sync D2H copies in between)");
    unsigned *other = nullptr;
    CK(hipMalloc(&other, 64 << 20));
    round("after hipMalloc(64 MB)");
    CK(hipMemset(other, 0xAB, 64 << 20));
    CK(hipDeviceSynchronize());
    round("after hipMemset(other, 0xAB) on the null stream");
    CK(hipMemset(other, 0, 4));
    round("after hipMemset(other, 0, 4 B) on the null stream");
    hipLaunchKernelGGL(scribble, dim3(1024), dim3(256), 0, 0, other, 1u << 18, 7u);
    CK(hipDeviceSynchronize());
    round("after a kernel on the null stream");
    {
        hipStream_t s3;
        CK(hipStreamCreateWithFlags(&s3, hipStreamNonBlocking));
        CK(hipMemsetAsync(other, 0x5A, 1 << 20, s3));
        CK(hipStreamSynchronize(s3));
        round("after hipMemsetAsync(other, 0x5A) on a third stream");
        CK(hipMemsetAsync(other, 0x11, 4, tests[0].s));
        CK(hipStreamSynchronize(tests[0].s));
        round("after hipMemsetAsync(other, 0x11, 4 B) on test 0's own stream");
    }
    std::vector<unsigned> host(1 << 20, 3u);
    CK(hipMemcpy(other, host.data(), 4 << 20, hipMemcpyHostToDevice));
    round("after a pageable H2D hipMemcpy");
    // the same kinds of operation MANY times: a ring of blit constants / kernel arguments would wrap
    for (int i = 0; i < 200; ++i) CK(hipMemcpy(host.data(), other + i, 4 * (1 + i % 7), hipMemcpyDeviceToHost));
    round("after 200 small pageable D2H hipMemcpy");
    for (int i = 0; i < 200; ++i) CK(hipMemcpy(other + i, host.data(), 4 * (1 + i % 7), hipMemcpyHostToDevice));
    round("after 200 small pageable H2D hipMemcpy");
    for (int i = 0; i < 200; ++i) CK(hipMemset(other + 64 * i, 0x30 + i % 64, 4 * (1 + i % 50)));
    CK(hipDeviceSynchronize());
    round("after 200 small hipMemset (non-zero patterns), null stream");
    for (int i = 0; i < 200; ++i) CK(hipMemsetAsync(other + 64 * i, 0x30 + i % 64, 4 * (1 + i % 50), tests[2].s));
    CK(hipStreamSynchronize(tests[2].s));
    round("after 200 small hipMemsetAsync (non-zero) on test 2's stream");
    for (int i = 0; i < 200; ++i) CK(hipMemsetAsync(other + 64 * i, 0, 4 * (1 + i % 50), tests[2].s));
    CK(hipStreamSynchronize(tests[2].s));
    round("after 200 small hipMemsetAsync (zero) on test 2's stream");
    for (int i = 0; i < 200; ++i) CK(hipMemset(other + 64 * i, 0x30 + i % 64, (1 << 20) + 4 * i));
    CK(hipDeviceSynchronize());
    round("after 200 x 1 MB hipMemset (non-zero), null stream");
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(scribble, dim3(4), dim3(256), 0, 0, other, 1000u, (unsigned)i);
    CK(hipDeviceSynchronize());
    round("after 200 kernels on the null stream");
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(scribble, dim3(4), dim3(256), 0, tests[2].s, other, 1000u, (unsigned)i);
    CK(hipStreamSynchronize(tests[2].s));
    round("after 200 kernels on test 2's stream");
    {
        unsigned *pin = nullptr;
        CK(hipHostMalloc((void **)&pin, 1 << 20, hipHostMallocDefault));
        for (int i = 0; i < 200; ++i) CK(hipMemcpyAsync(other + 64 * i, pin, 4096, hipMemcpyHostToDevice, tests[2].s));
        CK(hipStreamSynchronize(tests[2].s));
        round("after 200 pinned H2D hipMemcpyAsync on test 2's stream");
        for (int i = 0; i < 200; ++i) CK(hipMemcpyAsync(pin, other + 64 * i, 4096, hipMemcpyDeviceToHost, tests[2].s));
        CK(hipStreamSynchronize(tests[2].s));
        round("after 200 pinned D2H hipMemcpyAsync on test 2's stream");
        CK(hipHostFree(pin));
    }
    {   // another graph is instantiated: does that "heal" the first ones?
        Test t;
        t.n = 4096;
        t.rows = 100;
        t.build(false);
        round("after instantiating one more linear graph");
        printf("   (the new graph itself: %s)\n", t.replay() ? "BAD" : "ok");
        round("after replaying the new graph");
    }
    CK(hipFree(other));
    round("after hipFree");
    printf("%-64s", "20 quiet replays each (async copies on the same stream only)");
    for (Test &t : tests) printf(" n=%zu:%s", t.n, t.replay_quiet(20) ? "BAD" : "ok");
    printf("\n");
    return 0;
}
