// Groth16 prover, part 1: the RCCL entry points loaded on demand and ProveWs, the proof slot (streams, workspaces, captured graphs).
// Included by prover.cpp only (one translation unit: the anonymous namespace is intended).
#pragma once

namespace mg {
namespace {

// RCCL behind the C ABI (mg_ctx_opts.exchange = MG_EXCHANGE_RCCL): the library is dlopen'ed the first time a context asks for
// it -- a process that already holds one (PyTorch ships its own librccl.so.1) gets THAT copy, two RCCL runtimes in one
// process would each claim the devices -- and only the six entry points below are used. MANTA_RCCL_LIB names another file.
struct Rccl {
    void *h = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    static Rccl *get() {
        static Rccl *inst = [] () -> Rccl * {
            Rccl *r = new Rccl();
            const char *names[] = {std::getenv("MANTA_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
            for (const char *n : names)
                if (n && !r->h) r->h = dlopen(n, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD); // already in the process?
            for (const char *n : names)
                if (n && !r->h) r->h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (!r->h) {
                delete r;
                return nullptr;
            }
            r->CommInitAll = (decltype(r->CommInitAll))dlsym(r->h, "ncclCommInitAll");
            r->CommDestroy = (decltype(r->CommDestroy))dlsym(r->h, "ncclCommDestroy");
            r->AllGather = (decltype(r->AllGather))dlsym(r->h, "ncclAllGather");
            r->GroupStart = (decltype(r->GroupStart))dlsym(r->h, "ncclGroupStart");
            r->GroupEnd = (decltype(r->GroupEnd))dlsym(r->h, "ncclGroupEnd");
            r->GetErrorString = (decltype(r->GetErrorString))dlsym(r->h, "ncclGetErrorString");
            if (!r->CommInitAll || !r->CommDestroy || !r->AllGather || !r->GroupStart || !r->GroupEnd || !r->GetErrorString) {
                delete r;
                return nullptr;
            }
            return r;
        }();
        return inst;
    }
};

// One in-flight proof (or batch of proofs): device scratch for the witness map, its five MSM workspaces (each
// with its own stream), a pinned copy of z, and -- after two eager runs that size every buffer -- captured
// hipGraphs of the GPU side (~90 launches: the prover is launch-bound at manta-pay circuit sizes, and
// concurrent host threads stop contending on the runtime). Default ("single"): two graphs, the G2 MSM alone on
// its stream and everything else (witness map, four G1 MSMs forked and joined) on the slot's main stream, so
// that the host can take the G1 results and assemble A and C while the G2 MSM -- the longest chain -- is still
// running. "split": six single-stream graphs with eager event fork/join (no multi-branch graph at all; 15 %
// slower). The launch streams are high-priority pooled streams: see stream_pool_get() for the runtime defect
// that makes this necessary for multi-branch graphs.
struct ProveWs {
    DevBuf z, a; // a holds the three work vectors a | b | c back to back (one allocation, one memset)
    hipStream_t stream = nullptr;             // witness map (and the launch stream of the main graph); the G1 MSMs join back into it
    hipStream_t side[2] = {nullptr, nullptr}; // [0]: the G2 MSM (the longest chain); [1]: a, b_g1, l in MANTA_PROVE_STREAMS=3 mode
    hipEvent_t z_ready = nullptr, h_ready = nullptr, fork = nullptr;
    MsmWorkspace *mw[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    GroupEngine *me[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    void *h_z = nullptr; // pinned staging of the assignment
    size_t h_z_cap = 0;
    hipGraphExec_t g_all = nullptr; // "single" mode: witness map + the four G1 MSMs, forked and joined on `stream`
    hipGraphExec_t g_g2 = nullptr;  // "single" mode: the G2 MSM, alone on its own stream
    hipGraphExec_t g_wm = nullptr;                                          // witness map body (main stream)
    hipGraphExec_t g_msm[5] = {nullptr, nullptr, nullptr, nullptr, nullptr}; // MSM i on its stream
    bool graphs_ready = false;
    u32 k = 1; // proofs per pass (the slot's buffers and its captured graph are sized for exactly this batch)
    // this slot runs the three G1 MSMs over the assignment (a, b_g1, l) as ONE pass of the MSM pipeline over the concatenated
    // query (ProverImpl::z3_bs_full_) on mw[0]; mw[1] and mw[3] stay idle. Fixed for the slot's lifetime (its graphs capture it).
    bool z3 = false;
    // round 5: a z3 slot replays THREE linear graphs -- (witness map + h MSM) on `stream`, the combined a | b_g1 | l MSM on side[1],
    // the G2 MSM on side[0] -- instead of a forked part A: a captured multi-branch graph starts its branches one after the other
    // (the combined MSM began 210-290 us into the proof) and its hipGraphLaunch costs ~110 us of host time against 15-30 us for a
    // linear one. Round 4 built exactly this and withdrew it because C came out wrong next to other contexts: that was the memset
    // node of the witness map in a packet-captured linear graph (profiles/r05_linear_graph_defect.txt), gone now. MANTA_Z3_LINEAR=0:
    // the forked graph (A/B).
    bool linear3 = false;
    int flavour = 0; // lin_flavour(): 0 forked graph, 1 linear3 of a lone proof, 2 / 3 linear3 beside other passes (combined / G2 MSM on the normal-priority stream)
    StreamSet sset; // linear3 slots: three streams on three different hardware queues (runtime.cpp); id < 0: plain pooled streams
    bool poisoned = false; // a stream capture of this slot failed: its streams are not trusted again (dropped, never pooled)
    std::vector<const uint64_t *> z_parts; // this pass's assignments as k separate host buffers (coalesced calls), else empty
    int device = 0;
    u64 gen = 0, last_use = 0; // circuit generation the slot belongs to; LRU stamp for the idle-slot cap
    int eager_runs = 0, capture_tries = 0;
    bool no_graph = false;
    // kernel timing (bench.py's per-phase split of a single proof): timing events, created on first use; a timed pass is
    // enqueued eagerly -- [0] before the upload of z, [1] after it, [2] witness map done, [3 + 2i], [4 + 2i] around MSM i on
    // its stream, [13] part A joined, [14] G2 MSM done
    hipEvent_t tev[15] = {};
    bool timed = false;
    void drop_graphs() {
        if (g_all) hipGraphExecDestroy(g_all);
        if (g_g2) hipGraphExecDestroy(g_g2);
        g_all = g_g2 = nullptr;
        if (g_wm) hipGraphExecDestroy(g_wm);
        g_wm = nullptr;
        for (int i = 0; i < 5; ++i) {
            if (g_msm[i]) hipGraphExecDestroy(g_msm[i]);
            g_msm[i] = nullptr;
        }
        graphs_ready = false;
    }
    ~ProveWs() {
        // hipFree / hipHostFree / hipGraphExecDestroy / hipEventDestroy beside another thread's stream capture invalidate that
        // capture (error 901): like every allocating path, a slot's destruction takes the shared side of the capture lock
        // (evicted, stale-generation and poisoned slots are deleted from proving threads)
        HeavyOp not_beside_a_capture;
        int prev = 0;
        hipGetDevice(&prev);
        hipSetDevice(device);
        // nothing of this slot may still be tracked by the runtime when its graph execs, events and buffers go (tools/soak.py,
        // round 5: a heap corruption inside the process after ~5 minutes of contexts being recycled under load)
        if (stream) (void)hipStreamSynchronize(stream);
        for (hipStream_t sd : side)
            if (sd) (void)hipStreamSynchronize(sd);
        for (int i = 0; i < 5; ++i)
            if (mw[i] && mw[i]->stream) (void)hipStreamSynchronize(mw[i]->stream);
        (void)hipGetLastError();
        drop_graphs();
        for (int i = 0; i < 5; ++i)
            if (mw[i]) {
                mw[i]->run_on = nullptr;
                mw[i]->in_graph_slot = false;
                mw[i]->notify = false;
                if (poisoned) { // its stream may have joined the invalidated capture: abandoned (leaked on purpose), never pooled
                    mw[i]->stream = nullptr;
                    delete mw[i];
                } else {
                    me[i]->ws_release(mw[i]);
                }
            }
        z.release();
        a.release();
        if (h_z) hipHostFree(h_z);
        if (z_ready) hipEventDestroy(z_ready);
        if (h_ready) hipEventDestroy(h_ready);
        if (fork) hipEventDestroy(fork);
        for (auto &e : tev)
            if (e) hipEventDestroy(e);
        if (sset.id >= 0) {
            if (poisoned) sset.main = sset.g2 = sset.z3 = nullptr; // (abandoned, the set id is free again)
            stream_set_release(sset);
        } else if (!poisoned) { // never destroyed: see stream_pool_get(); a poisoned slot's streams are abandoned (leaked on purpose)
            stream_pool_put(stream);
            stream_pool_put(side[0]);
            stream_pool_put(side[1]);
        }
        hipSetDevice(prev);
    }
};

static inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    __asm__ __volatile__("yield");
#endif
}

enum GraphMode { GRAPH_OFF = GRAPH_MODE_OFF, GRAPH_SINGLE = GRAPH_MODE_SINGLE, GRAPH_SPLIT = GRAPH_MODE_SPLIT };

} // namespace
} // namespace mg
