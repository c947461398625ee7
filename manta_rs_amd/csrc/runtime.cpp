// Common host runtime of the MI355X Groth16 hot path: error reporting, grow-only device buffers,
// the per-engine MSM workspace pool and the (curve, group) engine registry.
#include "engine.h"
#include "tuning.h"
#include <cstddef>
#include <cstring>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>

// (The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues, default 4; streams that share one
// serialise. The library leaves the variable alone: measured on MI355X, PrivateTransfer shape, 6 queues against 4 give
// +6 % batched proofs/s and -4 % sequential latency in a process that only proves (tools/hw_queues_sweep.sh) but -20 % for
// two threads of single proofs and nothing for the batched stream inside bench.py's process; 8 queues halve everything.)

namespace mg {

static thread_local std::string g_last_error;

void set_last_hip_error(hipError_t e, const char *expr, const char *file, int line) {
    char buf[512];
    std::snprintf(buf, sizeof(buf), "HIP error %d (%s) at %s:%d: %s", (int)e, hipGetErrorString(e), file, line, expr);
    g_last_error = buf;
}
void set_last_error_text(const char *text) { g_last_error = text ? text : ""; }
const char *last_error_string() { return g_last_error.c_str(); }

static bool g_kernel_timing = false;
static thread_local float g_last_acc_ms = 0.f;
void set_kernel_timing(bool on) { g_kernel_timing = on; }
bool kernel_timing() { return g_kernel_timing; }
void set_last_accumulate_ms(float ms) { g_last_acc_ms = ms; }
float last_accumulate_ms() { return g_last_acc_ms; }
static thread_local float g_last_acc_mhz = 0.f;
void set_last_accumulate_mhz(float mhz) { g_last_acc_mhz = mhz; }
float last_accumulate_mhz() { return g_last_acc_mhz; }
static thread_local float g_last_ntt_ms[4] = {0, 0, 0, 0}, g_last_prove_ms[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
void set_last_ntt_ms(const float v[4]) { std::memcpy(g_last_ntt_ms, v, sizeof(g_last_ntt_ms)); }
void get_last_ntt_ms(float v[4]) { std::memcpy(v, g_last_ntt_ms, sizeof(g_last_ntt_ms)); }
static thread_local float g_last_pass_host_ms[3] = {0, 0, 0};
void set_last_pass_host_ms(const float v[3]) { std::memcpy(g_last_pass_host_ms, v, sizeof(g_last_pass_host_ms)); }
void get_last_pass_host_ms(float v[3]) { std::memcpy(v, g_last_pass_host_ms, sizeof(g_last_pass_host_ms)); }
void set_last_prove_ms(const float v[10]) { std::memcpy(g_last_prove_ms, v, sizeof(g_last_prove_ms)); }
void get_last_prove_ms(float v[10]) { std::memcpy(v, g_last_prove_ms, sizeof(g_last_prove_ms)); }

int DevBuf::reserve(size_t bytes) {
    if (bytes <= cap) return MG_OK;
    if (p) {
        hipFree(p);
        p = nullptr;
        cap = 0;
    }
    size_t want = bytes + bytes / 8 + 256; // slack so that near-equal sizes do not thrash
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {
        p = nullptr;
        set_last_hip_error(e, "hipMalloc(DevBuf)", __FILE__, __LINE__);
        return e == hipErrorOutOfMemory ? MG_ERR_OOM : MG_ERR_HIP;
    }
    cap = want;
    return MG_OK;
}
void DevBuf::release() {
    if (p) hipFree(p);
    p = nullptr;
    cap = 0;
}

// Proof-slot streams (the streams graphs are captured from and launched on) are HIGH-PRIORITY streams, pooled
// for the life of the process and never destroyed. Both properties work around one defect of the HIP runtime
// this library is loaded next to (libamdhip64 of ROCm 7.0, hip::Graph::UpdateStreams): when a multi-branch
// graph is launched, the exec's internal parallel streams that share a hardware queue with the launch stream
// are skipped, but only ONE spare stream exists -- if two of them alias the launch stream's queue the loop
// reads past the vector and hipGraphLaunch segfaults (seen about once in ten processes after contexts had
// come and gone; stream destruction unbalances the queue use counts and made it 7 in 8). The exec's internal
// streams are normal-priority; a high-priority launch stream lives in the other hardware-queue pool and can
// never alias them.
std::shared_mutex &capture_mutex() {
    static std::shared_mutex mu;
    return mu;
}
static thread_local int tl_heavy_depth = 0;
HeavyOp::HeavyOp() {
    if (tl_heavy_depth++ == 0) capture_mutex().lock_shared();
}
HeavyOp::~HeavyOp() {
    if (--tl_heavy_depth == 0) capture_mutex().unlock_shared();
}
static std::mutex g_stream_mu;
static std::vector<hipStream_t> g_stream_pools[MAX_DEVICES]; // a stream belongs to the device it was created on
static std::map<hipStream_t, int> g_stream_dev;
int current_device() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= MAX_DEVICES) return 0;
    return d;
}
hipStream_t stream_pool_get() {
    const int dev = current_device();
    {
        std::lock_guard<std::mutex> g(g_stream_mu);
        std::vector<hipStream_t> &pool = g_stream_pools[dev];
        if (!pool.empty()) {
            hipStream_t s = pool.back();
            pool.pop_back();
            return s;
        }
    }
    hipStream_t s = nullptr;
    int lo = 0, hi = 0;
    if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) return nullptr;
    if (hipStreamCreateWithPriority(&s, hipStreamNonBlocking, hi) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> g(g_stream_mu);
    g_stream_dev[s] = dev;
    return s;
}
void stream_pool_put(hipStream_t s) {
    if (!s) return;
    std::lock_guard<std::mutex> g(g_stream_mu);
    auto it = g_stream_dev.find(s);
    g_stream_pools[it == g_stream_dev.end() ? 0 : it->second].push_back(s);
}
// The same for the NORMAL-priority streams of MSM workspaces (the branch streams of a proof slot's forked capture) and every other
// short-lived stream of the library: pooled, never destroyed. Round 5, tools/soak.py with a context recycled every 2 s: the engine's
// idle pool overflowed, workspaces were deleted, their streams destroyed -- and the process died within a minute with
// "free(): corrupted unsorted chunks" (or answered MG_ERROR_HIP under a debugger's timing): the stream-destruction defect of
// hip::Graph::UpdateStreams described above, now reached through the workspaces. No hipStreamDestroy is left on any product path.
static std::vector<hipStream_t> g_nstream_pools[MAX_DEVICES];
static std::map<hipStream_t, int> g_nstream_dev;
hipStream_t stream_pool_get_normal() {
    const int dev = current_device();
    {
        std::lock_guard<std::mutex> g(g_stream_mu);
        std::vector<hipStream_t> &pool = g_nstream_pools[dev];
        if (!pool.empty()) {
            hipStream_t s = pool.back();
            pool.pop_back();
            return s;
        }
    }
    hipStream_t s = nullptr;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> g(g_stream_mu);
    g_nstream_dev[s] = dev;
    return s;
}
// ---- the calling thread's setup stream (engine.h): one pooled normal-priority non-blocking stream per device it has touched
namespace {
struct ThreadSetupStreams {
    hipStream_t s[MAX_DEVICES] = {};
    ~ThreadSetupStreams() {
        for (hipStream_t x : s)
            if (x) stream_pool_put_normal(x); // (idle: every user waits for its own work before it returns)
    }
};
} // namespace
hipStream_t setup_stream() {
    static thread_local ThreadSetupStreams tl;
    const int dev = current_device();
    if (!tl.s[dev]) tl.s[dev] = stream_pool_get_normal();
    return tl.s[dev];
}
// ---- tuning (tuning.h): compiled-in defaults, the environment applied ONCE through one table, then mg_set_tuning --------------
Tuning tuning_defaults() {
    Tuning t{};
    t.struct_size = (uint32_t)sizeof(Tuning);
    t.graph_mode = GRAPH_MODE_SINGLE;
    t.graph_mode_batch = -1;
    t.prove_streams = 6;
    t.linear_chains = 3;
    t.coalesce_inflight = 2;
    t.coalesce_gather_us = 100; // (0 / 40 / 80 / 150 / 300 us -> six threads 1 609 / 1 621 / 1 784 / 1 850 / 1 620 proofs/s)
    t.batch_inflight = 3;
    t.queue_aware = 1;
    t.msm_dedicated_queues = 1;
    t.window_bits_narrow = t.window_bits_wide = t.window_bits_h = t.window_bits_g2 = 0;
    t.full_table_bytes = -1;
    return t;
}
int normalize_tuning(Tuning &t) {
    auto in = [](int v, int lo, int hi) { return v >= lo && v <= hi; };
    if (!in(t.graph_mode, 0, 2) || !in(t.graph_mode_batch, -1, 2)) return MG_ERR_ARG;
    if (!(t.prove_streams >= 3 && t.prove_streams <= 6)) return MG_ERR_ARG; // (1 = the linear part A of the round-4 defect: -DMG_DIAG only)
    if (!in(t.linear_chains, 0, 3) || !in(t.coalesce_inflight, 0, 4) || !in(t.coalesce_gather_us, 0, 2000)) return MG_ERR_ARG;
    if (!in(t.batch_inflight, 1, 16) || !in(t.queue_aware, 0, 1) || !in(t.msm_dedicated_queues, 0, 2)) return MG_ERR_ARG;
    if (t.window_bits_narrow && !in(t.window_bits_narrow, 2, 20)) return MG_ERR_ARG;
    if (t.window_bits_wide && !in(t.window_bits_wide, 6, 16)) return MG_ERR_ARG;
    if (t.window_bits_h && !in(t.window_bits_h, 2, 20)) return MG_ERR_ARG;
    if (t.window_bits_g2 && !in(t.window_bits_g2, 4, 18)) return MG_ERR_ARG;
    if (t.full_table_bytes < -1) return MG_ERR_ARG;
    t.struct_size = (uint32_t)sizeof(Tuning);
    return MG_OK;
}
namespace {
enum EnvKind { ENV_INT, ENV_GRAPH, ENV_GB };
struct EnvField {
    const char *name;
    EnvKind kind;
    size_t off;
};
#define MG_TF(f) offsetof(Tuning, f)
const EnvField ENV_TABLE[] = {
    {"MANTA_GRAPH", ENV_GRAPH, MG_TF(graph_mode)},
    {"MANTA_GRAPH_BATCH", ENV_GRAPH, MG_TF(graph_mode_batch)},
    {"MANTA_PROVE_STREAMS", ENV_INT, MG_TF(prove_streams)},
    {"MANTA_Z3_LINEAR", ENV_INT, MG_TF(linear_chains)},
    {"MANTA_COALESCE", ENV_INT, MG_TF(coalesce_inflight)},
    {"MANTA_COALESCE_GATHER_US", ENV_INT, MG_TF(coalesce_gather_us)},
    {"MANTA_BATCH_INFLIGHT", ENV_INT, MG_TF(batch_inflight)},
    {"MANTA_QUEUE_AWARE", ENV_INT, MG_TF(queue_aware)},
    {"MANTA_MSM_DEDICATED_QUEUES", ENV_INT, MG_TF(msm_dedicated_queues)},
    {"MANTA_PROVE_C", ENV_INT, MG_TF(window_bits_narrow)},
    {"MANTA_PROVE_CW", ENV_INT, MG_TF(window_bits_wide)},
    {"MANTA_PROVE_CH", ENV_INT, MG_TF(window_bits_h)},
    {"MANTA_PROVE_CG2", ENV_INT, MG_TF(window_bits_g2)},
    {"MANTA_FULL_TABLE_GB", ENV_GB, MG_TF(full_table_bytes)},
};
#undef MG_TF
const char *const ENV_NAMES[] = {"MANTA_GRAPH", "MANTA_GRAPH_BATCH", "MANTA_PROVE_STREAMS", "MANTA_Z3_LINEAR", "MANTA_COALESCE",
                                 "MANTA_COALESCE_GATHER_US", "MANTA_BATCH_INFLIGHT", "MANTA_QUEUE_AWARE", "MANTA_MSM_DEDICATED_QUEUES",
                                 "MANTA_PROVE_C", "MANTA_PROVE_CW", "MANTA_PROVE_CH", "MANTA_PROVE_CG2", "MANTA_FULL_TABLE_GB",
                                 "MANTA_RCCL_LIB" /* prover.cpp: where librccl.so is */, nullptr};
std::mutex g_tuning_mu;
Tuning g_tuning;
bool g_tuning_init = false;
// the ONE place the shipped library reads MANTA_* tuning variables; a value the field's range refuses is ignored (the default stays)
void tuning_from_env_locked() {
    g_tuning = tuning_defaults();
    for (const EnvField &f : ENV_TABLE) {
        const char *e = std::getenv(f.name);
        if (!e || !*e) continue;
        Tuning t = g_tuning;
        char *p = (char *)&t + f.off;
        if (f.kind == ENV_GB) {
            const double gb = std::atof(e);
            *(int64_t *)p = gb >= 0 ? (int64_t)(gb * 1e9) : -1;
        } else if (f.kind == ENV_GRAPH) {
            int v = GRAPH_MODE_SINGLE;
            if (!std::strcmp(e, "off") || !std::strcmp(e, "0")) v = GRAPH_MODE_OFF;
            else if (!std::strcmp(e, "split") || !std::strcmp(e, "2")) v = GRAPH_MODE_SPLIT;
            *(int32_t *)p = v;
        } else {
            *(int32_t *)p = (int32_t)std::atoi(e);
        }
        if (normalize_tuning(t) == MG_OK) g_tuning = t;
    }
    g_tuning_init = true;
}
} // namespace
const Tuning &tuning() {
    // (a snapshot per thread: contexts copy it when they are created; set_tuning may run beside readers)
    static thread_local Tuning snap;
    std::lock_guard<std::mutex> g(g_tuning_mu);
    if (!g_tuning_init) tuning_from_env_locked();
    snap = g_tuning;
    return snap;
}
int set_tuning(const Tuning &t_in) {
    Tuning t = t_in;
    const int rc = normalize_tuning(t);
    if (rc) return rc;
    std::lock_guard<std::mutex> g(g_tuning_mu);
    g_tuning = t;
    g_tuning_init = true; // (the environment no longer applies: the host has spoken)
    return MG_OK;
}
const char *const *tuning_env_names() { return ENV_NAMES; }

static std::atomic<int> g_graph_clients{0};
GraphClient::GraphClient() { g_graph_clients.fetch_add(1, std::memory_order_relaxed); }
GraphClient::~GraphClient() { g_graph_clients.fetch_sub(1, std::memory_order_relaxed); }
int graph_clients_alive() { return g_graph_clients.load(std::memory_order_relaxed); }
// ---- streams on hardware queues of their own (engine.h MsmWorkspace::solo)
static std::vector<hipStream_t> g_dstream_pools[MAX_DEVICES];
static std::map<hipStream_t, int> g_dstream_dev;
static bool g_dstream_refused[MAX_DEVICES] = {};
hipStream_t stream_pool_get_dedicated() {
    if (!tuning().msm_dedicated_queues) return nullptr; // (0: never -- e.g. hosts that need non-blocking semantics against their own NULL-stream work)
    const int dev = current_device();
    {
        std::lock_guard<std::mutex> g(g_stream_mu);
        if (g_dstream_refused[dev]) return nullptr;
        std::vector<hipStream_t> &pool = g_dstream_pools[dev];
        if (!pool.empty()) {
            hipStream_t s = pool.back();
            pool.pop_back();
            return s;
        }
    }
    // The runtime's pool of shared normal-priority hardware queues must be FULL before the first dedicated queue exists: a process
    // that has created fewer ordinary streams than the pool holds (the runtime's GPU_MAX_HW_QUEUES, default 4) otherwise finds the dedicated queue
    // counted as a pool member, and later ordinary streams -- the NULL stream included -- are multiplexed onto it (measured: 316
    // Mscalar/s in a process with no foreign stream, 379-387 with two or more: profiles/r06_pipeline_phase.txt). Eight ordinary
    // streams are created once per device and kept in the library's normal-priority pool.
    {
        static std::mutex prime_mu;
        static bool primed[MAX_DEVICES] = {};
        std::lock_guard<std::mutex> g(prime_mu);
        if (!primed[dev]) {
            primed[dev] = true;
            const int nq = 8; // (GPU_MAX_HW_QUEUES is 4 by default; a host that raised it to 8 is covered too)
            std::vector<hipStream_t> fill;
            for (int i = 0; i < nq; ++i) {
                hipStream_t f = nullptr;
                if (hipStreamCreateWithFlags(&f, hipStreamNonBlocking) != hipSuccess) break;
                fill.push_back(f);
            }
            (void)hipGetLastError();
            std::lock_guard<std::mutex> g2(g_stream_mu);
            for (hipStream_t f : fill) g_nstream_dev[f] = dev, g_nstream_pools[dev].push_back(f);
        }
    }
    int cus = 0;
    hipStream_t s = nullptr;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) {
        std::vector<uint32_t> mask((size_t)(cus + 31) / 32, 0u); // every CU: the point is the queue, not the mask
        for (int b = 0; b < cus; ++b) mask[(size_t)b / 32] |= 1u << (b % 32);
        if (hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()) != hipSuccess) s = nullptr;
    }
    std::lock_guard<std::mutex> g(g_stream_mu);
    if (!s) {
        (void)hipGetLastError();
        g_dstream_refused[dev] = true; // (not asked again: the MSMs use their ordinary streams)
        return nullptr;
    }
    g_dstream_dev[s] = dev;
    return s;
}
void stream_pool_put_dedicated(hipStream_t s) {
    if (!s) return;
    std::lock_guard<std::mutex> g(g_stream_mu);
    auto it = g_dstream_dev.find(s);
    g_dstream_pools[it == g_dstream_dev.end() ? 0 : it->second].push_back(s);
}
// ---- queue-aware stream sets (see queues.hip) -------------------------------------------------------------------------------
namespace {
struct QueueClasses {
    std::vector<hipStream_t> rep;               // one stream of every hardware queue seen so far
    std::vector<std::vector<hipStream_t>> idle; // pooled streams by queue
    std::map<hipStream_t, int> cls;
};
struct DevQueues {
    QueueClasses pr[2]; // 0: normal priority, 1: high priority (the two levels have hardware queues of their own)
    std::vector<char> set_used;
    unsigned *mem = nullptr, token = 0;
    bool failed = false, ready = false;
    int probes = 0;
};
constexpr int PROBE_RETRIES = 4;
DevQueues g_q[MAX_DEVICES];
std::mutex g_q_mu;
bool queue_aware_on() { return tuning().queue_aware != 0; }
hipStream_t new_stream(int pr) {
    hipStream_t s = nullptr;
    int lo = 0, hi = 0;
    if (pr) {
        if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess || hipStreamCreateWithPriority(&s, hipStreamNonBlocking, hi) != hipSuccess)
            return nullptr;
    } else if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess)
        return nullptr;
    return s;
}
// creates one stream of priority level pr, finds its queue and pools it; returns the class or -1
int add_classified_stream(DevQueues &q, int pr) {
    QueueClasses &c = q.pr[pr];
    hipStream_t s = new_stream(pr);
    if (!s) return -1;
    int found = -1;
    for (size_t k = 0; k < c.rep.size() && found < 0; ++k) {
        const int r = streams_share_queue(c.rep[k], s, q.mem, &q.token);
        if (r < 0) return -1; // (the stream is leaked: the library destroys none)
        if (r == 1) found = (int)k;
    }
    if (found < 0) { // the first stream seen on a queue stays the library's own: later probes run on it, never inside a user's work
        c.rep.push_back(s);
        c.idle.emplace_back();
        return (int)c.rep.size() - 1;
    }
    c.cls[s] = found;
    c.idle[found].push_back(s);
    return found;
}
hipStream_t take_in_class(DevQueues &q, int pr, int want) {
    QueueClasses &c = q.pr[pr];
    const int n = (int)c.rep.size();
    if (n == 0) return nullptr;
    want %= n;
    for (int tries = 0; c.idle[want].empty() && tries < 4 * n + 4; ++tries)
        if (add_classified_stream(q, pr) < 0) return nullptr;
    if (c.idle[want].empty()) return nullptr;
    hipStream_t s = c.idle[want].back();
    c.idle[want].pop_back();
    return s;
}
void give_back(DevQueues &q, int pr, hipStream_t s) {
    if (!s) return;
    auto it = q.pr[pr].cls.find(s);
    if (it != q.pr[pr].cls.end()) q.pr[pr].idle[it->second].push_back(s);
}
} // namespace
// the hardware queues of the current device are known (probed on the first call: ~20 ms); false: disabled or the probe failed
static bool sets_ready_locked(DevQueues &q) {
    if (q.failed) return false;
    if (!q.ready) {
        // twelve streams per level: with the runtime's least-used-queue rule that is three per hardware queue, one of them kept for probes
        if (!q.mem) {
            q.failed = hipHostMalloc((void **)&q.mem, 64) != hipSuccess;
            if (!q.failed) q.mem[0] = q.mem[1] = 0;
        }
        for (int pr = 0; pr < 2 && !q.failed; ++pr)
            for (int i = 0; i < 12 && !q.failed; ++i) q.failed = add_classified_stream(q, pr) < 0;
        // fewer than three high-priority queues: a slot cannot have three of its own
        if (!q.failed) q.failed = q.pr[1].rep.size() < 3 || q.pr[0].rep.empty();
        if (q.failed) {
            (void)hipGetLastError();
            // not latched for the life of the process: a probe that ran on a busy GPU (fewer than three high-priority classes
            // seen) is repeated by a later context, at most PROBE_RETRIES times (the streams of a failed probe stay in the classes
            // they were put in; the library destroys none)
            if (++q.probes < PROBE_RETRIES) q.failed = false;
            return false;
        }
        q.ready = true;
    }
    return true;
}
bool stream_sets_ready() {
    if (!queue_aware_on()) return false;
    const int dev = current_device();
    HeavyOp probes_synchronise_streams; // (not beside a capture)
    std::lock_guard<std::mutex> g(g_q_mu);
    return sets_ready_locked(g_q[dev]);
}
int stream_queue_counts(int *normal, int *high) {
    const int dev = current_device();
    std::lock_guard<std::mutex> g(g_q_mu);
    if (!g_q[dev].ready) return 0;
    *normal = (int)g_q[dev].pr[0].rep.size(), *high = (int)g_q[dev].pr[1].rep.size();
    return 1;
}
bool stream_set_acquire(StreamSet &out, bool z3_high) {
    if (!queue_aware_on()) return false;
    const int dev = current_device();
    HeavyOp probes_synchronise_streams;
    std::lock_guard<std::mutex> g(g_q_mu);
    DevQueues &q = g_q[dev];
    if (!sets_ready_locked(q)) return false;
    size_t id = 0;
    while (id < q.set_used.size() && q.set_used[id]) ++id;
    if (id == q.set_used.size()) q.set_used.push_back(0);
    const int i = (int)id;
    hipStream_t a = take_in_class(q, 1, 2 * i), b = take_in_class(q, 1, 2 * i + 1);
    hipStream_t c = z3_high ? take_in_class(q, 1, 2 * i + 2) : take_in_class(q, 0, i);
    if (!a || !b || !c) {
        give_back(q, 1, a), give_back(q, 1, b), give_back(q, z3_high ? 1 : 0, c);
        (void)hipGetLastError();
        return false;
    }
    q.set_used[id] = 1;
    out.main = a, out.g2 = b, out.z3 = c, out.id = i, out.dev = dev, out.z3_high = z3_high;
    return true;
}
void stream_set_release(StreamSet &s) {
    if (s.id < 0) return;
    std::lock_guard<std::mutex> g(g_q_mu);
    DevQueues &q = g_q[s.dev];
    give_back(q, 1, s.main), give_back(q, 1, s.g2), give_back(q, s.z3_high ? 1 : 0, s.z3);
    if ((size_t)s.id < q.set_used.size()) q.set_used[s.id] = 0;
    s = StreamSet();
}
void stream_pool_put_normal(hipStream_t s) {
    if (!s) return;
    std::lock_guard<std::mutex> g(g_stream_mu);
    auto it = g_nstream_dev.find(s);
    g_nstream_pools[it == g_nstream_dev.end() ? 0 : it->second].push_back(s);
}

MsmWorkspace::~MsmWorkspace() {
    DevBuf *all[] = {&keys_in, &keys_out, &vals_in, &vals_out, &sort_tmp, &buckets, &pkeys[0], &pkeys[1],
                     &ppts[0], &ppts[1], &redA,     &redS,    &misc, &count, &front, &extra, &folded, &scratch};
    for (DevBuf *b : all) b->release();
    if (h_stage) hipHostFree(h_stage);
    if (h_flag) hipHostFree(h_flag);
    if (h_clk) hipHostFree(h_clk);
    if (solo) { // (drained, then pooled: never destroyed)
        (void)hipStreamSynchronize(solo);
        stream_pool_put_dedicated(solo);
    }
    if (d_token) hipFree(d_token);
    if (done) hipEventDestroy(done);
    if (t0) hipEventDestroy(t0);
    if (t1) hipEventDestroy(t1);
    if (side_fork) hipEventDestroy(side_fork);
    if (side_join) hipEventDestroy(side_join);
    if (stream) { // (drained, then pooled: never destroyed)
        (void)hipStreamSynchronize(stream);
        stream_pool_put_normal(stream);
    }
}

MsmWorkspace *GroupEngine::ws_acquire() {
    {
        std::lock_guard<std::mutex> g(ws_mu_);
        if (!ws_free_.empty()) {
            MsmWorkspace *w = ws_free_.back();
            ws_free_.pop_back();
            return w;
        }
    }
    MsmWorkspace *w = new MsmWorkspace();
    w->device = current_device();
    if (!(w->stream = stream_pool_get_normal()) ||
        hipEventCreateWithFlags(&w->done, hipEventDisableTiming) != hipSuccess ||
        hipEventCreate(&w->t0) != hipSuccess || hipEventCreate(&w->t1) != hipSuccess) {
        delete w;
        return nullptr;
    }
    return w;
}
// The idle pool is bounded: beyond MAX_IDLE_WS a released workspace is destroyed (its grow-only buffers are
// freed), so HBM held for past MSM sizes / dropped contexts does not accumulate.
void GroupEngine::ws_release(MsmWorkspace *w) {
    if (!w) return;
    w->use_solo = false;
    {
        std::lock_guard<std::mutex> g(ws_mu_);
        if (ws_free_.size() < MAX_IDLE_WS) {
            ws_free_.push_back(w);
            return;
        }
    }
    int prev = 0;
    hipGetDevice(&prev);
    hipSetDevice(w->device);
    delete w;
    hipSetDevice(prev);
}

// one engine per (device, curve, group): an engine's workspaces, streams and tables live on the device that was
// current when it was first asked for
GroupEngine *get_engine(int curve, int group) {
    static std::mutex mu;
    static GroupEngine *tab[MAX_DEVICES][2][2] = {};
    if (curve < 0 || curve > 1 || group < 1 || group > 2) return nullptr;
    const int dev = current_device();
    std::lock_guard<std::mutex> g(mu);
    GroupEngine *&e = tab[dev][curve][group - 1];
    if (!e) {
        if (curve == 0)
            e = group == 1 ? make_engine_bn254_g1() : make_engine_bn254_g2();
        else
            e = group == 1 ? make_engine_bls381_g1() : make_engine_bls381_g2();
    }
    return e;
}

} // namespace mg
