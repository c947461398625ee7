// Which hardware queue does a HIP stream live on?  (round 5)
//
// The HIP runtime multiplexes a process's streams onto a few hardware queues per priority level (GPU_MAX_HW_QUEUES, default 4), in
// an order that depends on every stream the process has ever created, and two kernels whose streams share a queue run one behind
// the other. A single proof is three chains (witness map + h MSM | combined a, b_g1, l MSM | G2 MSM); two proofs in flight are
// six, and whether two LONG chains of different proofs landed on one queue decided whether two host threads gained 0 % or 35 %
// over one (tools/queue_probe.hip prints the sharing matrix of a process; profiles/r05_hw_queues.txt). The runtime has no query
// for it, so it is measured: a kernel on stream A waits (bounded) for a word that a kernel on stream B writes. If B's kernel
// cannot start before A's has ended, the two share a queue.
#include "engine.h"

namespace mg {
namespace {
__global__ void queue_probe_wait(const unsigned *flag, unsigned token, unsigned *res, long long ticks) {
    const long long t0 = wall_clock64(); // 100 MHz
    unsigned saw = 0;
    do {
        if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == token) {
            saw = 1;
            break;
        }
        __builtin_amdgcn_s_sleep(8);
    } while (wall_clock64() - t0 < ticks);
    __atomic_store_n(res, saw ? token : ~token, __ATOMIC_RELEASE);
}
__global__ void queue_probe_set(unsigned *flag, unsigned token) { __atomic_store_n(flag, token, __ATOMIC_RELEASE); }
} // namespace

// 1: the two streams share a hardware queue, 0: their kernels run side by side, -1: HIP error. mem: two words of page-locked host
// memory (flag, result) owned by the caller; token_counter: the caller's running token. Must not run beside a stream capture (the
// caller holds HeavyOp): it synchronises the two streams.
int streams_share_queue(hipStream_t a, hipStream_t b, unsigned *mem, unsigned *token_counter) {
    // a "shared" verdict must repeat twice, the last time with a LONGER bound (0.5, 0.5, 2 ms): a host stall between the two launches, or a
    // busy GPU on which B's one-thread kernel waits for a wave slot behind other ranks' / this process's own proofs, fakes one
    // (independent queues reported as shared collapse the classes for the life of the process)
    long long ticks = 50000LL; // 100 MHz: 0.5 ms
    for (int attempt = 0; attempt < 3; ++attempt) {
        if (attempt == 2) ticks *= 4; // (a true "shared" verdict costs 3 ms: ~50 ms for the first context of a device)
        const unsigned token = ++*token_counter;
        hipLaunchKernelGGL(queue_probe_wait, dim3(1), dim3(1), 0, a, mem, token, mem + 1, ticks);
        hipLaunchKernelGGL(queue_probe_set, dim3(1), dim3(1), 0, b, mem, token);
        if (hipStreamSynchronize(a) != hipSuccess || hipStreamSynchronize(b) != hipSuccess) return -1;
        const unsigned res = __atomic_load_n(mem + 1, __ATOMIC_ACQUIRE);
        if (res == token) return 0;
        if (res != ~token) return -1;
    }
    return 1;
}
} // namespace mg
