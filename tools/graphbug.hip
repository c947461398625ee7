// dev probe: does hipGraphLaunch of a multi-stream captured graph crash depending on how many streams the
// process created before? usage: graphbug <fillers> [branches]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void k(int *p) { atomicAdd(p, 1); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d: %s\n", #x, __LINE__, hipGetErrorString(e)); exit(2);} } while (0)
int main(int argc, char **argv) {
    int fillers = argc > 1 ? atoi(argv[1]) : 0, nb = argc > 2 ? atoi(argv[2]) : 5;
    int destroy = argc > 3 ? atoi(argv[3]) : 0; // 0: keep fillers; 1..4: destroy fillers with index % 4 == destroy-1
    int prio = argc > 4 ? atoi(argv[4]) : 0;    // 1: the launch stream is a high-priority stream
    int *d; CK(hipMalloc(&d, 4)); CK(hipMemset(d, 0, 4));
    std::vector<hipStream_t> fill(fillers);
    for (auto &s : fill) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    if (destroy) for (int i = 0; i < fillers; ++i) if (i % 4 == destroy - 1) CK(hipStreamDestroy(fill[i]));
    hipStream_t main_s;
    if (prio) { int lo, hi; CK(hipDeviceGetStreamPriorityRange(&lo, &hi)); CK(hipStreamCreateWithPriority(&main_s, hipStreamNonBlocking, hi)); }
    else CK(hipStreamCreateWithFlags(&main_s, hipStreamNonBlocking));
    std::vector<hipStream_t> br(nb); std::vector<hipEvent_t> done(nb);
    for (int i = 0; i < nb; ++i) { CK(hipStreamCreateWithFlags(&br[i], hipStreamNonBlocking)); CK(hipEventCreateWithFlags(&done[i], hipEventDisableTiming)); }
    hipEvent_t ready; CK(hipEventCreateWithFlags(&ready, hipEventDisableTiming));
    CK(hipStreamBeginCapture(main_s, hipStreamCaptureModeThreadLocal));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, main_s, d);
    CK(hipEventRecord(ready, main_s));
    for (int i = 0; i < nb; ++i) {
        CK(hipStreamWaitEvent(br[i], ready, 0));
        for (int j = 0; j < 4; ++j) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, br[i], d);
        CK(hipEventRecord(done[i], br[i]));
    }
    for (int j = 0; j < 4; ++j) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, main_s, d);
    for (int i = 0; i < nb; ++i) CK(hipStreamWaitEvent(main_s, done[i], 0));
    hipGraph_t g; CK(hipStreamEndCapture(main_s, &g));
    hipGraphExec_t ex; CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0)); CK(hipGraphDestroy(g));
    for (int r = 0; r < 3; ++r) { CK(hipGraphLaunch(ex, main_s)); CK(hipStreamSynchronize(main_s)); }
    int h = 0; CK(hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost));
    printf("fillers=%d branches=%d destroy=%d prio=%d OK count=%d\n", fillers, nb, destroy, prio, h);
    return 0;
}
