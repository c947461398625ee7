// Groth16 prover on one MI355X: device-resident proving key + R1CS, witness map on the GPU,
// five MSMs on five HIP streams, serial assembly on the host.
//
// Replaces ark-groth16 ^0.3.0 `create_proof` (prover.rs) + `R1CStoQAP::witness_map` (r1cs_to_qap.rs),
// reached from manta-crypto/src/arkworks/groth16.rs:597; restated from SURVEY.md section 3.2 / App. B.1:
//   h     = witness_map(z)
//   h_acc = MSM(h_query, h)              l_acc = MSM(l_query, z[P..])
//   g_a   = r*delta_g1 + a_query[0] + MSM(a_query[1..], z[1..]) + alpha_g1
//   g1_b  = s*delta_g1 + b_g1_query[0] + MSM(b_g1_query[1..], z[1..]) + beta_g1      (only if r != 0)
//   g2_b  = s*delta_g2 + b_g2_query[0] + MSM(b_g2_query[1..], z[1..]) + beta_g2
//   g_c   = s*g_a + r*g1_b - (r s)*delta_g1 + l_acc + h_acc
//   proof = (g_a, g2_b, g_c) as arkworks canonical compressed bytes.
// An unsatisfied witness is not an error (ark-groth16 only debug_asserts it): a non-verifying proof
// comes back, exactly like the reference in release builds.
#include "prover.h"
#include "tuning.h"
// The six RCCL entry points this file calls, declared here from NCCL's stable C ABI (ncclResult_t 0 = success, ncclUint64 = 5):
// librccl.so is loaded on demand (Rccl::get) and never linked, and its header is not a build dependency of a single-GPU host
// (advisor r4). Only decltype() of the prototypes is used: nothing below references the symbols themselves.
extern "C" {
typedef struct ncclComm *ncclComm_t;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclUint64 = 5 } ncclDataType_t;
ncclResult_t ncclCommInitAll(ncclComm_t *comms, int ndev, const int *devlist);
ncclResult_t ncclCommDestroy(ncclComm_t comm);
ncclResult_t ncclAllGather(const void *sendbuff, void *recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm, hipStream_t stream);
ncclResult_t ncclGroupStart(void);
ncclResult_t ncclGroupEnd(void);
const char *ncclGetErrorString(ncclResult_t result);
}
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <shared_mutex>
#include <stdexcept>
#include <thread>
#include <vector>

namespace mg {

FrEngine *get_ntt_engine(int curve) { // one per (device, curve): twiddle and scale tables are device memory
    static std::mutex mu;
    static FrEngine *tab[MAX_DEVICES][2] = {};
    if (curve < 0 || curve > 1) return nullptr;
    const int dev = current_device();
    std::lock_guard<std::mutex> g(mu);
    if (!tab[dev][curve]) tab[dev][curve] = curve == 0 ? make_fr_engine_bn254() : make_fr_engine_bls381();
    return tab[dev][curve];
}

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

class ProverImpl : public Prover {
  public:
    // what the deployment decided for THIS context (mg_ctx_opts.tuning, else the process-wide values when it was created): tuning.h
    Tuning tn_ = tuning();
    GraphClient counted_; // (stand-alone MSMs leave their dedicated queues alone while this context lives: engine.h)
    // streams of a forked pass. (1 = part A as ONE linear chain, the topology of the round-4 wrong-C defect: diagnosis builds only)
    int prove_streams() const {
#ifdef MG_DIAG
        if (ab_knob("MANTA_PROVE_STREAMS", 0) == 1) return 1;
#endif
        return tn_.prove_streams;
    }
    // Replay the GPU side of a pass as hipGraphs: 0 off, 1 single (default; two graphs, the G2 chain alone so the host can assemble A
    // and C while it still runs), 2 split (six single-stream graphs, eager event fork / join). Batched passes (k >= 4 proofs) may take
    // another topology than single proofs (graph_mode_batch; measured within noise: profiles/r05_batched_ab.txt).
    GraphMode graph_mode() const { return (GraphMode)tn_.graph_mode; }
    GraphMode graph_mode_for(u32 k) const { return k >= 4 && tn_.graph_mode_batch >= 0 ? (GraphMode)tn_.graph_mode_batch : graph_mode(); }
    int coalesce_gather_us() const { return tn_.coalesce_gather_us; }
    int coalesce_inflight() const { return tn_.coalesce_inflight; }
    int batch_inflight() const { return tn_.batch_inflight < (int)MAX_IDLE_SLOTS ? tn_.batch_inflight : (int)MAX_IDLE_SLOTS; }
    int curve_ = 0;
    int dev_ = 0;                      // the HIP device this (shard of the) context lives on
    u32 shard_ = 0, n_shards_ = 1;     // range shard g of G: every MSM of a proof covers the g-th contiguous slice of its query
    // task placement (SURVEY.md 8(e) last row; prover_create_task): bit i set = this context computes MSM i (a, b_g1, b_g2, l, h)
    // in full; the others are some other rank's. Only the partials interface works on such a context; it launches eagerly.
    u32 task_mask_ = 0x1f;
    bool does(int i) const { return (task_mask_ >> i) & 1u; }
    // a context from prover_create_shard with more than one shard: it holds slice g of every query and nothing of the other
    // slices (they are other processes'), so a whole proof cannot come out of it -- only partials_launch / assemble work
    bool lone_range_shard() const { return n_shards_ > 1 && peers_.empty() && shard_owner_ == nullptr; }
    ProverImpl *shard_owner_ = nullptr; // in-process peers: the shard-0 object that owns this one
    std::vector<ProverImpl *> peers_;  // shard 0 only: shards 1 .. G-1 (owned); a pass runs on all of them, shard 0 assembles
    FrEngine *fr_ = nullptr;
    GroupEngine *g1_ = nullptr, *g2_ = nullptr;
    u64 V_ = 0, P_ = 0, h_len_ = 0, m_ = 0;
    unsigned log_d_ = 0;
    bool have_r1cs_ = false;
    bool sets_ok_ = false; // queue-aware stream sets available on this device (runtime.cpp)
    BaseSet *a_bs_ = nullptr, *b1_bs_ = nullptr, *b2_bs_ = nullptr, *h_bs_ = nullptr, *l_bs_ = nullptr;
    BaseSet *h_bs_wide_ = nullptr; // the h query again with wider windows, for batched passes (nullptr: same as h_bs_)
    // the z queries again with 10-bit windows for batched passes (fewer mixed additions; single proofs want the
    // short bucket reduce of narrow windows, above all on the G2 chain); nullptr: same as the narrow set
    BaseSet *a_bs_wide_ = nullptr, *b1_bs_wide_ = nullptr, *b2_bs_wide_ = nullptr, *l_bs_wide_ = nullptr;
    // the five queries once more as FULL tables (every multiple of every window: the MSM is one plain sum), for passes of ONE
    // proof -- their latency chain loses the sort, the merge into buckets and the bucket reduce; nullptr: bucket tables
    BaseSet *a_bs_full_ = nullptr, *b1_bs_full_ = nullptr, *b2_bs_full_ = nullptr, *l_bs_full_ = nullptr, *h_bs_full_ = nullptr;
    // Round 4: a_query | b_g1_query | l_query (padded to the a query's indexing) as ONE full table (BaseSet::n_sets = 3). The three
    // MSMs share the scalar vector z, so a single proof runs them as one digit kernel, one accumulate launch and one chain of
    // merge levels with three bucket keys instead of three chains on three streams: a captured multi-branch graph starts its
    // branches one after the other (tools/ubench_graph_branches.hip: 4 branches progress like 3, a fifth waits for a whole
    // branch), which left the third of these MSMs starting 630 us into a 880 us proof (profiles/r04_proof_timeline_*). Replaces
    // the three separate full tables of an unsharded context (same HBM); MANTA_Z3=0 keeps them apart.
    BaseSet *z3_bs_full_ = nullptr;
    HostPoint alpha_g1_, beta_g1_, delta_g1_, beta_g2_, delta_g2_, a0_, b1_0_, b2_0_;
    HostPoint a0_alpha_, b10_beta_, b20_beta_; // constant terms of g_a, g1_b, g2_b folded once
    void *delta1_tab_ = nullptr, *delta2_tab_ = nullptr; // fixed-base tables for r*delta, s*delta, rs*delta
    DevCsr A_, B_, C_;
    std::vector<u32> h_query_host_; // kept until the domain size is known (set_r1cs), then re-laid
    std::mutex mu_;
    // proofs hold it shared for the length of a pass, set_r1cs exclusively: replacing the circuit waits for the passes in
    // flight and no pass ever sees a half-replaced one (mantagpu.h: prove is re-entrant on one context)
    mutable std::shared_mutex shape_mu_;
    u64 gen_ = 0; // bumped by every set_r1cs; a slot remembers the generation it was sized and captured for
    std::map<u32, std::vector<ProveWs *>> ws_free_; // idle proof slots, by batch size
    std::set<u32> no_graph_keys_; // slot kinds whose capture failed for a deterministic reason: their slots stay eager (mu_)
    static constexpr int CAPTURE_TRIES = 8; // passes that run eagerly because the capture lock was busy before build_graphs waits for it
    size_t idle_slots_ = 0;
    u64 lru_tick_ = 0;
    static constexpr size_t MAX_IDLE_SLOTS = 16; // (eight batch sizes of coalesced calls x two passes in flight) per context: beyond it the least recently used idle slot is destroyed

    ~ProverImpl() override {
        HeavyOp no_capture_meanwhile;
        exchange_destroy();
        for (ProverImpl *q : peers_) delete q;
        hipSetDevice(dev_);
        // (h_bs_ is created by set_r1cs)
        if (a_bs_) g1_->bases_destroy(a_bs_);
        if (b1_bs_) g1_->bases_destroy(b1_bs_);
        if (h_bs_) g1_->bases_destroy(h_bs_);
        if (h_bs_wide_) g1_->bases_destroy(h_bs_wide_);
        if (l_bs_) g1_->bases_destroy(l_bs_);
        if (b2_bs_) g2_->bases_destroy(b2_bs_);
        if (z3_bs_full_) g1_->bases_destroy(z3_bs_full_);
        if (a_bs_full_) g1_->bases_destroy(a_bs_full_);
        if (b1_bs_full_) g1_->bases_destroy(b1_bs_full_);
        if (l_bs_full_) g1_->bases_destroy(l_bs_full_);
        if (h_bs_full_) g1_->bases_destroy(h_bs_full_);
        if (b2_bs_full_) g2_->bases_destroy(b2_bs_full_);
        if (a_bs_wide_) g1_->bases_destroy(a_bs_wide_);
        if (b1_bs_wide_) g1_->bases_destroy(b1_bs_wide_);
        if (l_bs_wide_) g1_->bases_destroy(l_bs_wide_);
        if (b2_bs_wide_) g2_->bases_destroy(b2_bs_wide_);
        if (delta1_tab_) g1_->hp_table_free(delta1_tab_);
        if (delta2_tab_) g2_->hp_table_free(delta2_tab_);
        free_csr(A_);
        free_csr(B_);
        free_csr(C_);
        for (auto &kv : ws_free_)
            for (ProveWs *w : kv.second) delete w;
    }
    static void free_csr(DevCsr &M) {
        if (M.row_ptr) hipFree(M.row_ptr);
        if (M.col) hipFree(M.col);
        if (M.val) hipFree(M.val);
        M = DevCsr();
    }
    u64 domain_size() const override { return have_r1cs_ ? (u64)1 << log_d_ : 0; }
    void table_bytes(u64 out2[2]) const override {
        out2[0] = out2[1] = 0;
        for (const BaseSet *b : {a_bs_, b1_bs_, b2_bs_, l_bs_, h_bs_, a_bs_wide_, b1_bs_wide_, b2_bs_wide_, l_bs_wide_, h_bs_wide_})
            if (b) out2[0] += b->bytes;
        for (const BaseSet *b : {a_bs_full_, b1_bs_full_, b2_bs_full_, l_bs_full_, h_bs_full_, z3_bs_full_})
            if (b) out2[1] += b->bytes;
        for (const ProverImpl *q : peers_) {
            u64 t[2];
            q->table_bytes(t);
            out2[0] += t[0], out2[1] += t[1];
        }
    }

    // window bits for precomputed tables, by MSM length (HBM is plentiful: trade table size for fewer
    // buckets to fold and no doubling chain -- tuned on MI355X)
    int pre_c_for(u64 n) const {
        if (tn_.window_bits_narrow > 0) return tn_.window_bits_narrow; // tuning override: ONE width for every bucket table of the key
        // Measured on MI355X for the PrivateTransfer shape (n = 35k / 65k): c = 6..8 -> 2.0 ms per proof,
        // c = 9..13 -> 2.5-2.7 ms, c = 14 -> 3.0 ms. Few buckets keep the latency-bound bucket reduce short
        // (B = 128: two tiles); the extra windows only add perfectly parallel mixed additions.
        if (n <= (1u << 17)) return 8;
        if (n <= (1u << 19)) return 12;
        return 17; // 255 = 15 x 17, 254 < 15 x 17: fifteen windows on both curves (digits_kernel negates scalars above r / 2)
    }

    // FULL tables for the queries single proofs run on (mg_bases_create with a negative width: every multiple of every window
    // tabulated, the MSM is one plain sum -- no sort, no merge into buckets, no bucket reduce on the latency chain of a proof).
    // They are bought with HBM, and a signer holds three contexts (`MultiProvingContext`, manta-accounting/src/transfer/
    // canonical.rs:561-588), so the budget is a property of the CONTEXT (mg_ctx_opts.full_table_bytes; default a tenth of the
    // device's HBM; 0 = bucket tables only) and covers its five tables together. MANTA_FULL_TABLE_GB overrides it (GB per
    // context), MANTA_FULL_C fixes the width. A context sharded over several entries of one device splits the budget.
    int64_t full_budget_ = 0;   // bytes for this shard's five full tables
    int full_c_plan_[5] = {0, 0, 0, 0, 0}; // planned widths: a, b_g1, b_g2, l, h (0 = none)
    static u64 full_cost(GroupEngine *g, u64 n, int c) {
        return ((u64)((g->scalar_bits() + c - 1) / c) << (c - 1)) * n * (u64)g->base_record_bytes();
    }
    static bool full_fits_index(GroupEngine *g, u64 n, int c) { return (((u64)((g->scalar_bits() + c - 1) / c) << (c - 1)) * n) < ((u64)1 << 31); }
    // widths of the five tables under `budget`: the widest uniform width c in 4 .. 8 whose five tables fit together, then single
    // queries one step wider while they fit, the longest chains first (b_g2, h, a, b_g1, l). n[i] = entries of query i on this shard.
    void plan_full_tables(const u64 n[5], int64_t budget, int out[5], bool tie_abl = false) const {
        for (int i = 0; i < 5; ++i) out[i] = 0;
        static const int fixed = [] {
            const int v = ab_knob("MANTA_FULL_C", 0);
            return v >= 2 && v <= 12 ? v : 0;
        }();
        if (budget <= 0) return;
        GroupEngine *ge[5] = {g1_, g1_, g2_, g1_, g1_};
        auto total = [&](const int c[5]) {
            u64 t = 0;
            for (int i = 0; i < 5; ++i)
                if (c[i] && n[i]) t += full_cost(ge[i], n[i], c[i]);
            return t;
        };
        auto ok = [&](const int c[5]) {
            for (int i = 0; i < 5; ++i)
                if (c[i] && n[i] && !full_fits_index(ge[i], n[i], c[i])) return false;
            return total(c) <= (u64)budget;
        };
        int c[5];
        const int hi = fixed ? fixed : 8, lo = fixed ? fixed : 4;
        int u = 0;
        for (int w = hi; w >= lo && !u; --w) {
            for (int i = 0; i < 5; ++i) c[i] = w;
            if (ok(c)) u = w;
        }
        if (!u) return;
        for (int i = 0; i < 5; ++i) c[i] = u;
        if (!fixed) {
            if (tie_abl) { // a, b_g1 and l share one concatenated table (z3_bs_full_): one width for the three
                for (int i : {2, 4}) {
                    if (c[i] >= 8) continue;
                    ++c[i];
                    if (!ok(c)) --c[i];
                }
                if (c[0] < 8) {
                    ++c[0], ++c[1], ++c[3];
                    if (!ok(c)) --c[0], --c[1], --c[3];
                }
            } else {
                static const int order[5] = {2, 4, 0, 1, 3};
                for (int step = 0; step < 5; ++step) {
                    const int i = order[step];
                    if (c[i] >= 8) continue;
                    ++c[i];
                    if (!ok(c)) --c[i];
                }
            }
        }
        for (int i = 0; i < 5; ++i) out[i] = n[i] ? c[i] : 0;
    }
    // the budget of this shard: the context option, else the tuning (MANTA_FULL_TABLE_GB / mg_set_tuning), else a tenth of the HBM; never more
    // than 40 % of what is free on the device right now, split between the shards of this context that share the device
    int64_t resolve_full_budget(int64_t opt_bytes, int shards_on_this_device) const {
        size_t free_b = 0, total_b = 0;
        const bool have = hipMemGetInfo(&free_b, &total_b) == hipSuccess;
        // (the context's own option first, then the tuning's budget -- MANTA_FULL_TABLE_GB lands there --, then a tenth of the device)
        double b = opt_bytes >= 0 ? (double)opt_bytes : (tn_.full_table_bytes >= 0 ? (double)tn_.full_table_bytes : (have ? (double)total_b / 10.0 : 24e9));
        if (have && b > 0.4 * (double)free_b) b = 0.4 * (double)free_b;
        if (shards_on_this_device > 1) b /= shards_on_this_device;
        return b > 0 ? (int64_t)b : 0;
    }

    // contiguous slice of an n-entry query owned by this shard
    size_t shard_lo(size_t n) const { return n * shard_ / n_shards_; }
    size_t shard_hi(size_t n) const { return n * (shard_ + 1) / n_shards_; }

    // allow_z3 = false: the context is driven through the partials interface (mg_ctx_create_shard, an RCCL exchange) -- its passes
    // fold every MSM by its own index, wants_z3() is false for them, and a combined table would only take the separate tables'
    // HBM and leave a / b_g1 / l on the slower bucket tables (advisor r4)
    int init(int curve, const mg_pk_view *pk, int device, u32 shard = 0, u32 n_shards = 1, int64_t full_table_bytes = -1,
             int shards_on_this_device = 1, bool allow_z3 = true) {
        curve_ = curve;
        dev_ = device;
        shard_ = shard;
        n_shards_ = n_shards;
        MG_HIP(hipSetDevice(dev_));
        fr_ = get_ntt_engine(curve);
        g1_ = get_engine(curve, 1);
        g2_ = get_engine(curve, 2);
        if (!fr_ || !g1_ || !g2_) return MG_ERR_ARG;
        sets_ok_ = stream_sets_ready(); // (the caller holds HeavyOp; the first context of a device probes its hardware queues)
        V_ = pk->n_vars;
        P_ = pk->n_inputs;
        h_len_ = pk->h_len;
        if (V_ < 2 || P_ < 1 || P_ >= V_ || h_len_ < 1) return MG_ERR_ARG;
        if (!pk->alpha_g1 || !pk->beta_g1 || !pk->delta_g1 || !pk->beta_g2 || !pk->delta_g2 || !pk->a_query ||
            !pk->b_g1_query || !pk->b_g2_query || !pk->h_query || !pk->l_query)
            return MG_ERR_ARG;
        const size_t w1 = (size_t)g1_->affine_words(), w2 = (size_t)g2_->affine_words();
        g1_->hp_from_affine(&alpha_g1_, (const u32 *)pk->alpha_g1);
        g1_->hp_from_affine(&beta_g1_, (const u32 *)pk->beta_g1);
        g1_->hp_from_affine(&delta_g1_, (const u32 *)pk->delta_g1);
        g2_->hp_from_affine(&beta_g2_, (const u32 *)pk->beta_g2);
        g2_->hp_from_affine(&delta_g2_, (const u32 *)pk->delta_g2);
        g1_->hp_from_affine(&a0_, (const u32 *)pk->a_query);
        g1_->hp_from_affine(&b1_0_, (const u32 *)pk->b_g1_query);
        g2_->hp_from_affine(&b2_0_, (const u32 *)pk->b_g2_query);
        a0_alpha_ = a0_;
        g1_->hp_add(&a0_alpha_, &alpha_g1_);
        b10_beta_ = b1_0_;
        g1_->hp_add(&b10_beta_, &beta_g1_);
        b20_beta_ = b2_0_;
        g2_->hp_add(&b20_beta_, &beta_g2_);
        delta1_tab_ = g1_->hp_table_create(&delta_g1_);
        delta2_tab_ = g2_->hp_table_create(&delta_g2_);
        int rc;
        if (n_shards_ > 1 && (V_ - P_ < n_shards_ || V_ - 1 < n_shards_)) return MG_ERR_ARG; // every shard owns >= 1 entry
        // this shard's slices of the z queries (entries 1 .. V-1 of a / b_g1 / b_g2) and of the l query
        const size_t zlo = shard_lo(V_ - 1), zn = shard_hi(V_ - 1) - zlo, llo = shard_lo(V_ - P_), ln = shard_hi(V_ - P_) - llo;
        const u32 *aq = (const u32 *)pk->a_query + (1 + zlo) * w1, *b1q = (const u32 *)pk->b_g1_query + (1 + zlo) * w1;
        const u32 *b2q = (const u32 *)pk->b_g2_query + (1 + zlo) * w2, *lq = (const u32 *)pk->l_query + llo * w1;
        const int c_z = pre_c_for(V_ - 1);
        const bool proof_sized = V_ - 1 <= (1u << 17) && tn_.window_bits_narrow == 0;
        if (proof_sized) {
            full_budget_ = resolve_full_budget(full_table_bytes, shards_on_this_device);
            u64 D = 1; // the domain the h query was made for: len(h_query) = D - 1 (ark setup) or D (MPC keys)
            while (D < h_len_) D <<= 1;
            const u64 nq[5] = {zn, zn, zn, ln, (u64)(D * (shard_ + 1) / n_shards_ - D * shard_ / n_shards_)};
            plan_full_tables(nq, full_budget_, full_c_plan_, allow_z3 && n_shards_ == 1 && task_mask_ == 0x1f && ab_knob("MANTA_Z3", 1) != 0);
        }
        const int f_z1a = -full_c_plan_[0], f_z1b = -full_c_plan_[1], f_z2 = -full_c_plan_[2], f_l = -full_c_plan_[3];
        if ((rc = g1_->bases_create(aq, zn, false, c_z, &a_bs_, true))) return rc;
        if ((rc = g1_->bases_create(b1q, zn, false, c_z, &b1_bs_, true))) return rc;
        // The G2 MSM is the latency-critical chain of a single proof: 6-bit windows (32 buckets: one tile, no second
        // reduce level) shorten it by four dependent additions (measured +4 % proofs/s); the extra windows only
        // add parallel mixed additions.
        const bool small = V_ - 1 <= (1u << 17) && tn_.window_bits_narrow == 0;
        // large keys (2^20 variables, BASELINE configs[2]): the 2^16 Fp2 buckets of a 17-bit window made the G2 bucket reduce a
        // 4.4 ms chain of latency-bound kernels next to a 0.7 ms accumulate (profiles/r04_config2_timeline.txt); 13-bit windows
        // -- 4 096 buckets, 20 windows instead of 15 -- trade a third more mixed additions for a sixteenth of the buckets
        int c_g2 = small ? 6 : (c_z > 13 ? 13 : c_z);
        if (tn_.window_bits_g2) c_g2 = tn_.window_bits_g2;
        if ((rc = g2_->bases_create(b2q, zn, false, c_g2, &b2_bs_, true))) return rc;
        if ((rc = g1_->bases_create(lq, ln, false, pre_c_for(V_ - P_), &l_bs_, true))) return rc;
        // (an optimisation: a table that does not fit any more is left out, the bucket tables above serve its MSM)
        auto try_full = [&](GroupEngine *g, const u32 *q, size_t cnt, int f, BaseSet **dst) -> int {
            if (!f) return MG_OK;
            const int r = g->bases_create(q, cnt, false, f, dst, true);
            if (r == MG_ERR_OOM) {
                *dst = nullptr;
                (void)hipGetLastError();
                return MG_OK;
            }
            return r;
        };
        if ((rc = try_full(g2_, b2q, zn, f_z2, &b2_bs_full_))) return rc; // the G2 chain first: the longest of a proof
        static const bool z3_on = ab_knob("MANTA_Z3", 1) != 0;
        const int c_z3 = std::min(full_c_plan_[0], std::min(full_c_plan_[1], full_c_plan_[3]));
        if (z3_on && allow_z3 && n_shards_ == 1 && task_mask_ == 0x1f && c_z3 >= 2 && 3 * (u64)zn * ((u64)((g1_->scalar_bits() + c_z3 - 1) / c_z3) << (c_z3 - 1)) < ((u64)1 << 31)) {
            // a | b_g1 | l as one table over the scalars z[1 .. V): l_query[i] belongs to z[P + i] = scalar P - 1 + i of that range
            std::vector<u32> cat((size_t)3 * zn * w1, 0u);
            std::memcpy(&cat[0], aq, zn * w1 * 4);
            std::memcpy(&cat[zn * w1], b1q, zn * w1 * 4);
            std::memcpy(&cat[(2 * zn + (size_t)(P_ - 1)) * w1], lq, ln * w1 * 4);
            const int r = g1_->bases_create(cat.data(), 3 * zn, false, -c_z3, &z3_bs_full_, true, 3);
            if (r == MG_ERR_OOM) {
                z3_bs_full_ = nullptr;
                (void)hipGetLastError();
            } else if (r) {
                return r;
            }
        }
        if (!z3_bs_full_) {
        if ((rc = try_full(g1_, aq, zn, f_z1a, &a_bs_full_))) return rc;
        if ((rc = try_full(g1_, b1q, zn, f_z1b, &b1_bs_full_))) return rc;
        if ((rc = try_full(g1_, lq, ln, f_l, &l_bs_full_))) return rc;
        }
        if (small) { // batched passes are throughput-bound: wider windows = fewer mixed additions (c = 10: +7 % measured over c = 8)
            int cw = 11; // (with three passes in flight: 10 / 11 / 12 -> 3 405-3 606 / 3 688-3 729 / 3 517-3 548 proofs/s, two runs each)
            if (tn_.window_bits_wide) cw = tn_.window_bits_wide; // tuning override
            if ((rc = g1_->bases_create(aq, zn, false, cw, &a_bs_wide_, true))) return rc;
            if ((rc = g1_->bases_create(b1q, zn, false, cw, &b1_bs_wide_, true))) return rc;
            if ((rc = g2_->bases_create(b2q, zn, false, cw, &b2_bs_wide_, true))) return rc;
            if ((rc = g1_->bases_create(lq, ln, false, cw, &l_bs_wide_, true))) return rc;
        }
        // h_query is stored in the bit-reversed order the witness map leaves h in; that order depends on
        // the domain size, known once the R1CS arrives (set_r1cs)
        h_query_host_.assign((const u32 *)pk->h_query, (const u32 *)pk->h_query + (size_t)h_len_ * w1);
        return MG_OK;
    }

    // Structural checks of one matrix as it arrives over the ABI (O(m + nnz) on the host): the device kernels loop
    // k = row_ptr[i] .. row_ptr[i+1] and gather z[col[k]] without further checks, so nothing malformed may pass here.
    static int validate_csr(const mg_csr *src, u64 m, u64 n_vars) {
        if (!src || !src->row_ptr || (src->nnz && (!src->col || !src->val))) return MG_ERR_ARG;
        if (src->nnz >= ((u64)1 << 32)) return MG_ERR_ARG;
        if (src->row_ptr[0] != 0 || src->row_ptr[m] != src->nnz) return MG_ERR_ARG;
        for (u64 i = 0; i < m; ++i)
            if (src->row_ptr[i] > src->row_ptr[i + 1]) return MG_ERR_ARG; // monotone => every entry <= row_ptr[m] = nnz
        for (u64 k = 0; k < src->nnz; ++k)
            if (src->col[k] >= n_vars) return MG_ERR_ARG;
        return MG_OK;
    }
    static int upload_csr(const mg_csr *src, u64 m, DevCsr &dst) { // dst is empty on entry; freed by the caller on failure
        dst.nnz = src->nnz;
        MG_HIP(hipMalloc((void **)&dst.row_ptr, (m + 1) * 4));
        MG_HIP(hipMalloc((void **)&dst.col, (src->nnz ? src->nnz : 1) * 4));
        MG_HIP(hipMalloc((void **)&dst.val, (src->nnz ? src->nnz : 1) * 32));
        MG_HIP(memcpy_sync(dst.row_ptr, src->row_ptr, (m + 1) * 4, hipMemcpyHostToDevice));
        if (src->nnz) {
            MG_HIP(memcpy_sync(dst.col, src->col, src->nnz * 4, hipMemcpyHostToDevice));
            MG_HIP(memcpy_sync(dst.val, src->val, src->nnz * 32, hipMemcpyHostToDevice));
        }
        return MG_OK;
    }

    // Replaces the circuit. All-or-nothing, on every shard at once: the exclusive locks of ALL shards are taken in the order in
    // which a pass takes its shared ones (shard 0, then the peers) -- so no pass is in flight on any of them; then, in two
    // phases, every shard first validates the three matrices, uploads them into temporaries and builds the h-query tables of
    // a new domain size -- a failure on any shard (say, out of memory on one device) frees the temporaries everywhere and
    // leaves the previous circuit, if any, fully usable -- and only when all of them have succeeded is every shard switched
    // over: a proof never runs with some shards on the new matrices and others on the old.
    struct StagedR1cs {
        DevCsr A, B, C;
        BaseSet *h = nullptr, *h_wide = nullptr, *h_full = nullptr;
        bool new_domain = false;
        unsigned lg = 0;
        u64 m = 0;
    };
    std::mutex set_mu_; // one set_r1cs at a time per context (shard 0's)
    int set_r1cs(const mg_csr *a, const mg_csr *b, const mg_csr *c, u64 m) override {
        int rc = MG_OK, prev = 0;
        MG_HIP(hipGetDevice(&prev));
        std::lock_guard<std::mutex> one_at_a_time(set_mu_);
        std::vector<ProverImpl *> all{this};
        all.insert(all.end(), peers_.begin(), peers_.end());
        // The exclusive locks are taken BEFORE staging as well: staging uploads with synchronous copies, builds window tables
        // on the default stream and ends in hipDeviceSynchronize, and the HIP runtime fails the graph capture of a proof
        // slot on another thread when that happens meanwhile (seen on MI355X: mg_groth16_prove returning a HIP error while a
        // circuit was being staged). With every shard locked no pass is in flight, none starts, nothing is capturing.
        std::vector<std::unique_lock<std::shared_mutex>> locks;
        for (ProverImpl *q : all) locks.emplace_back(q->shape_mu_);
        // (after the shape locks, never before: a pass that holds a shape lock shared may be waiting for the capture lock)
        HeavyOp no_capture_meanwhile;
        std::vector<StagedR1cs> st(all.size());
        for (size_t g = 0; g < all.size() && !rc; ++g) rc = all[g]->stage_r1cs(a, b, c, m, st[g]);
        if (rc) {
            for (size_t g = 0; g < all.size(); ++g) all[g]->discard_staged(st[g]);
            hipSetDevice(prev);
            return rc;
        }
        for (size_t g = 0; g < all.size(); ++g) all[g]->commit_staged(st[g]);
        hipSetDevice(prev);
        return MG_OK;
    }
    u64 n_vars() const override { return V_; }
    u64 n_inputs() const override { return P_; }
    u32 n_shards() const override { return n_shards_; }
    void discard_staged(StagedR1cs &st) {
        hipSetDevice(dev_);
        free_csr(st.A), free_csr(st.B), free_csr(st.C);
        if (st.h) g1_->bases_destroy(st.h);
        if (st.h_wide) g1_->bases_destroy(st.h_wide);
        if (st.h_full) g1_->bases_destroy(st.h_full);
        st.h = st.h_wide = st.h_full = nullptr;
    }
    // phase 1 on this shard: nothing visible to a proof is touched
    int stage_r1cs(const mg_csr *a, const mg_csr *b, const mg_csr *c, u64 m, StagedR1cs &st) {
        MG_HIP(hipSetDevice(dev_));
        if (m == 0 || m + P_ > ((u64)1 << 32)) return MG_ERR_ARG;
        unsigned lg = 0;
        while (((u64)1 << lg) < m + P_) ++lg; // GeneralEvaluationDomain::new(m + P) -> next power of two
        if ((int)lg > fr_->two_adicity()) return MG_ERR_DOMAIN;
        int rc;
        if ((rc = validate_csr(a, m, V_)) || (rc = validate_csr(b, m, V_)) || (rc = validate_csr(c, m, V_))) return rc;
        st.lg = lg;
        st.m = m;
        if ((rc = upload_csr(a, m, st.A)) || (rc = upload_csr(b, m, st.B)) || (rc = upload_csr(c, m, st.C))) return rc;
        st.new_domain = !h_bs_ || lg != log_d_; // (h_bs_ / log_d_ only change under set_mu_, which the caller holds)
        if (st.new_domain) { // (re)build the h-query base set for this domain
            const size_t D = (size_t)1 << lg, w1 = (size_t)g1_->affine_words();
            // this shard's slice [h_lo, h_hi) of the bit-reversed positions; entries beyond len(h_query) stay
            // infinity: h[D-1] = 0 anyway
            const size_t lo = shard_lo(D), hi = shard_hi(D);
            std::vector<u32> perm((hi - lo) * w1, 0u);
            for (size_t p = lo; p < hi; ++p) {
                size_t src = 0;
                for (unsigned bb = 0; bb < lg; ++bb) src |= ((p >> bb) & 1) << (lg - 1 - bb);
                if (src < h_len_) std::memcpy(&perm[(p - lo) * w1], &h_query_host_[src * w1], w1 * 4);
            }
            // The h MSM is the one with dense, uniform scalars -- half of all the mixed additions of a proof at
            // c = 8. Wider windows halve them, but lengthen its bucket reduce: measured on PrivateTransfer,
            // c_h = 8/10/12/14/16 -> 2033 / 2202 / 2219 / 2363 / 2287 proofs/s batched (k = 32); for single proofs the
            // reduce chain matters more (with the cooperative reduce: c_h = 8/10/12 -> 839 / 859 / 862 proofs/s).
            // The tables are small (80 MB), so single proofs and batches each get their own width.
            int ch = pre_c_for(D), ch_wide = ch;
            if (lg >= 16 && lg <= 17) ch = 12; // dense 2^16 scalars: a third fewer mixed additions, 32 reduce tiles (+3 %)
            if (lg <= 17) ch_wide = (int)lg - 2 < 8 ? 8 : ((int)lg - 2 > 14 ? 14 : (int)lg - 2);
            if (tn_.window_bits_h) ch = ch_wide = tn_.window_bits_h;
            // the h table gets what the budget has left after the four z / l tables: the planned width when the domain is the
            // one the key was made for, else the widest that still fits
            int f_h = 0;
            if (lg <= 17 && !tn_.window_bits_h && full_budget_ > 0) {
                int64_t left = full_budget_;
                for (const BaseSet *b : {a_bs_full_, b1_bs_full_, b2_bs_full_, l_bs_full_, z3_bs_full_}) // (z3 replaces a / b_g1 / l: advisor r4)
                    if (b) left -= (int64_t)b->bytes;
                for (int cc = full_c_plan_[4] ? std::max(full_c_plan_[4], 4) : 0; cc >= 4 && !f_h; --cc)
                    if (full_fits_index(g1_, hi - lo, cc) && (int64_t)full_cost(g1_, hi - lo, cc) <= left) f_h = -cc;
            }
            rc = g1_->bases_create(perm.data(), hi - lo, false, ch, &st.h);
            if (!rc && ch_wide != ch) rc = g1_->bases_create(perm.data(), hi - lo, false, ch_wide, &st.h_wide);
            if (!rc && f_h && g1_->bases_create(perm.data(), hi - lo, false, f_h, &st.h_full) != MG_OK) {
                st.h_full = nullptr; // optional: the bucket tables serve
                (void)hipGetLastError();
            }
            if (rc) return rc;
        }
        return MG_OK;
    }
    // phase 2 on this shard; the caller holds the exclusive shape lock of every shard
    void commit_staged(StagedR1cs &st) {
        hipSetDevice(dev_);
        std::lock_guard<std::mutex> g(mu_);
        free_csr(A_), free_csr(B_), free_csr(C_);
        A_ = st.A, B_ = st.B, C_ = st.C;
        st.A = st.B = st.C = DevCsr();
        if (st.new_domain) {
            if (h_bs_) g1_->bases_destroy(h_bs_);
            if (h_bs_wide_) g1_->bases_destroy(h_bs_wide_);
            if (h_bs_full_) g1_->bases_destroy(h_bs_full_);
            h_bs_ = st.h, h_bs_wide_ = st.h_wide, h_bs_full_ = st.h_full;
            st.h = st.h_wide = st.h_full = nullptr;
        }
        // pooled proof slots hold captured graphs and buffers sized for the previous shape: drop them (slots of
        // another generation that are still in flight cannot exist -- the exclusive locks waited for them)
        for (auto &kv : ws_free_)
            for (ProveWs *w : kv.second) delete w;
        ws_free_.clear();
        idle_slots_ = 0;
        ++gen_;
        m_ = st.m;
        log_d_ = st.lg;
        have_r1cs_ = true;
    }

    // How a SINGLE proof's slot replays (ProveWs::linear3). 0: the forked graph. 1: three linear graphs -- witness map + h | a|b_g1|l
    // | G2 -- on three high-priority streams: the shortest chain for a LONE proof (company 0: no other pass of this context in
    // flight). 2 / 3: the same beside other passes, with ONE chain on a normal-priority stream -- the combined MSM beside a batched
    // pass (company 2), the G2 MSM beside single proofs only (company 1: two host threads). The streams come from
    // stream_set_acquire (runtime.cpp): three DIFFERENT hardware queues per slot, and the two slots that two host threads keep in
    // flight share none -- before, which chains of the two proofs met on one queue was decided by the order in which the process
    // had created its streams, and two threads ran at 976 or 1 364 proofs/s from process to process (profiles/r05_hw_queues.txt).
    // A normal-priority chain costs a lone proof 18 %: flavour 1 keeps all three high. Which chain yields beside others is measured:
    // two threads 1 412-1 435 proofs/s with the combined MSM normal, 1 511-1 531 with the G2 MSM normal; six threads (singles beside
    // coalesced passes) 1 794-1 898 against 1 624-1 700.
    // MANTA_Z3_LINEAR: 0 never linear, 1 lone proofs only (round 5's first version), 2 no flavour 3, 3 (default) all of the above.
    int lin_flavour(u32 k, bool z3, int company) const {
        const int z3_linear = tn_.linear_chains;
        if (!(z3 && k == 1 && prove_streams() == 6 && graph_mode_for(k) == GRAPH_SINGLE)) return 0;
        if (company == 0) return z3_linear >= 1 ? 1 : 0;
        if (!(z3_linear >= 2 && sets_ok_)) return 0; // (linear graphs beside others need queues of their own: -18 % without)
        return company == 1 && z3_linear >= 3 ? 3 : 2;
    }
    static u32 slot_key(u32 k, bool z3, int flavour = 0) { return k | (z3 ? 1u << 16 : 0u) | ((u32)flavour << 17); }
    ProveWs *ws_acquire(u32 k = 1, bool z3 = false, int company = 0) {
        u64 gen;
        {
            std::lock_guard<std::mutex> g(mu_);
            gen = gen_;
            auto it = ws_free_.find(slot_key(k, z3, lin_flavour(k, z3, company)));
            while (it != ws_free_.end() && !it->second.empty()) {
                ProveWs *w = it->second.back();
                it->second.pop_back();
                --idle_slots_;
                if (w->gen == gen_) return w;
                delete w; // sized / captured for a previous circuit
            }
        }
        HeavyOp creates_streams_events_workspaces; // (not beside another thread's capture: ADVICE r5)
        ProveWs *w = new ProveWs();
        w->k = k;
        w->z3 = z3;
        w->gen = gen;
        w->device = dev_;
        {
            std::lock_guard<std::mutex> g(mu_);
            w->no_graph = no_graph_keys_.count(slot_key(k, z3, lin_flavour(k, z3, company))) != 0; // (a capture of this kind failed for good)
        }
        static const int z3_high = ab_knob("MANTA_Z3_HIGH", -1); // A/B: the combined MSM's stream of every linear3 slot normal (0) / high (1) priority
        w->flavour = lin_flavour(k, z3, company);
        if (w->flavour && stream_set_acquire(w->sset, z3_high >= 0 ? z3_high != 0 : w->flavour == 1))
            w->stream = w->sset.main, w->side[0] = w->sset.g2, w->side[1] = w->sset.z3;
        if (w->sset.id >= 0 && w->flavour == 3 && !w->sset.z3_high) std::swap(w->side[0], w->side[1]); // the G2 chain takes the normal-priority stream
        if ((w->sset.id < 0 && (!(w->stream = stream_pool_get()) || !(w->side[0] = stream_pool_get()) ||
                                !(w->side[1] = stream_pool_get()))) ||
            hipEventCreateWithFlags(&w->z_ready, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&w->h_ready, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&w->fork, hipEventDisableTiming) != hipSuccess) {
            delete w;
            return nullptr;
        }
        GroupEngine *me[5] = {g1_, g1_, g2_, g1_, g1_}; // a, b_g1, b_g2, l, h
        for (int i = 0; i < 5; ++i) {
            w->me[i] = me[i];
            w->mw[i] = me[i]->ws_acquire();
            if (!w->mw[i]) {
                delete w;
                return nullptr;
            }
            w->mw[i]->in_graph_slot = graph_mode_for(k) == GRAPH_SINGLE; // (multi-branch capture: no front levels)
        }
        // z3 slots: the combined a | b_g1 | l MSM announces its end through a pinned flag (MsmWorkspace::notify), so that the host
        // can fold its three results into s A + r B1 -- the one long piece of host work of a proof, ~0.1 ms -- while the h chain is
        // still running (finish_pass_body). MANTA_Z3_EARLY=0: wait for all of part A first, as before (A/B).
        static const bool z3_early = ab_knob("MANTA_Z3_EARLY", 1) != 0;
        w->mw[0]->notify = z3 && z3_early;
        // Three streams per proof, not six: the G2 MSM is the critical path (~3x a G1 MSM), so the three
        // z-MSMs over G1 run back to back beside it and the h MSM follows the witness map on the main stream.
        // Fewer streams = fewer hardware queues per proof in flight (the runtime multiplexes streams onto
        // GPU_MAX_HW_QUEUES queues; streams that share one serialise). MANTA_PROVE_STREAMS=6 restores one
        // stream per MSM.
        w->mw[2]->run_on = w->side[0]; // the G2 MSM (the critical path) gets a high-priority stream of its own
        if (w->flavour) {
            w->linear3 = true;
            w->mw[0]->run_on = w->side[1]; // the combined MSM: a high-priority pooled stream of its own
            w->mw[4]->run_on = w->stream;  // the h MSM follows the witness map on the main stream
            for (int i = 0; i < 5; ++i) w->mw[i]->in_graph_slot = false; // (single-stream captures only: front levels allowed)
        } else
        if (prove_streams() == 3) {
            w->mw[0]->run_on = w->side[1];
            w->mw[1]->run_on = w->side[1];
            w->mw[2]->run_on = w->side[0];
            w->mw[3]->run_on = w->side[1];
            w->mw[4]->run_on = w->stream;
        } else if (prove_streams() == 1) {
            for (int i = 0; i < 5; ++i) w->mw[i]->run_on = w->stream;
        } else if (prove_streams() == 4) { // three branches beside the G2 chain: (a, b_g1) back to back | l | witness map + h
            w->mw[0]->run_on = w->side[1];
            w->mw[1]->run_on = w->side[1];
            w->mw[4]->run_on = w->stream;
        } else if (prove_streams() == 5) { // (a, l) back to back | b_g1 | witness map + h
            w->mw[0]->run_on = w->side[1];
            w->mw[3]->run_on = w->side[1];
            w->mw[4]->run_on = w->stream;
        }
        return w;
    }
    // Idle slots are cached per exact batch size (their buffers and graphs are sized for it) but the cache is
    // bounded: a slot of an outdated circuit generation is destroyed, and beyond MAX_IDLE_SLOTS the least recently
    // used idle slot goes -- its MSM workspaces return to the engine pool, which is bounded too (runtime.cpp), so a
    // service that varies k or creates and drops contexts does not accumulate HBM.
    void ws_release(ProveWs *w) {
        std::vector<ProveWs *> doomed;
        {
            std::lock_guard<std::mutex> g(mu_);
            if (w->gen != gen_ || w->poisoned) {
                doomed.push_back(w);
            } else {
                w->last_use = ++lru_tick_;
                ws_free_[slot_key(w->k, w->z3, w->flavour)].push_back(w);
                ++idle_slots_;
                while (idle_slots_ > MAX_IDLE_SLOTS) {
                    std::vector<ProveWs *> *from = nullptr;
                    size_t at = 0;
                    for (auto &kv : ws_free_)
                        for (size_t i = 0; i < kv.second.size(); ++i)
                            if (!from || kv.second[i]->last_use < (*from)[at]->last_use) from = &kv.second, at = i;
                    if (!from) break;
                    doomed.push_back((*from)[at]);
                    from->erase(from->begin() + (long)at);
                    --idle_slots_;
                }
            }
        }
        for (ProveWs *d : doomed) delete d;
    }

    // Witness map for the slot's w->k assignments (stored back to back, like the three work vectors: member q of
    // a batch lives V resp. D elements after member q-1); h ends up in w->a.
    int reserve_witness_map(ProveWs *w) {
        const size_t D = (size_t)1 << log_d_, k = w->k, ww = (size_t)fr_->work_words() * 4; // bytes per work element
        int rc;
        if ((rc = w->z.reserve(k * V_ * 32)) || (rc = w->a.reserve(3 * k * D * ww))) return rc;
        return MG_OK;
    }
    // everything after the upload of z, on w->stream (this is what the witness-map graph captures)
    int enqueue_witness_map_body(ProveWs *w) {
        const size_t D = (size_t)1 << log_d_, k = w->k, ww = (size_t)fr_->work_words(); // u32 per work element
        int rc;
        hipStream_t s = w->stream;
        u32 *a = w->a.as<u32>(), *b = a + k * D * ww, *c = b + k * D * ww, *zz = w->z.as<u32>();
        const size_t zs = (size_t)V_ * 8, ds = D * ww;
#ifdef MG_DIAG
        // diagnosis builds: MG_DIAG_MEMSET=1 puts the round-4 memset node back in front of the SpMV (the negative control of
        // test_captured_graphs_survive_other_contexts: with it a LINEAR part A must go wrong); MG_DIAG_WM_STOP cuts the witness map
        static const int diag_memset = ab_knob("MG_DIAG_MEMSET", 0);
        static const int diag_stop = ab_knob("MG_DIAG_WM_STOP", 0);
        if (diag_memset) MG_HIP(hipMemsetAsync(w->a.p, 0, 3 * k * D * ww * 4, s));
        if (diag_stop == 1) return MG_OK;
#endif
        // A z, B z, C z in the reduced-radix work form; the A vector also gets the input-consistency rows a[m + j] = z_j, and every
        // vector its zero rows up to the domain size (the all-zero words are 0 in the work form too): no memset node in front of it
#ifdef MG_DIAG
        if (diag_memset) { // round 4 exactly: the memset node zeroes, the SpMV writes its m + P rows only
            if ((rc = fr_->spmv3(A_, B_, C_, zz, a, b, c, m_, P_, s, (u32)k, zs, ds, 0))) return rc;
        } else
#endif
        if ((rc = fr_->spmv3(A_, B_, C_, zz, a, b, c, m_, P_, s, (u32)k, zs, ds, D))) return rc;
#ifdef MG_DIAG
        if (diag_stop == 2) return MG_OK;
#endif
        // ifft x3, coset fft x3, (ab - c)/Z, coset ifft -- fused; leaves h bit-reversed in `a`
        if ((rc = fr_->qap_quotient(a, b, c, log_d_, s, (u32)k))) return rc;
        return MG_OK;
    }
    // H2D(z), recording z_ready
    int upload_z(ProveWs *w, const uint64_t *z) {
        int rc = reserve_witness_map(w);
        if (rc) return rc;
        if (w->z_parts.size() == w->k) { // coalesced single calls: one copy per assignment, from where it lies
            for (u32 q = 0; q < w->k; ++q)
                MG_HIP(hipMemcpyAsync((char *)w->z.p + (size_t)q * V_ * 32, w->z_parts[q], (size_t)V_ * 32, hipMemcpyHostToDevice, w->stream));
        } else
            MG_HIP(hipMemcpyAsync(w->z.p, z, (size_t)w->k * V_ * 32, hipMemcpyHostToDevice, w->stream));
        MG_HIP(hipEventRecord(w->z_ready, w->stream));
        return MG_OK;
    }
    // witness map after upload_z; records h_ready at the end
    int launch_witness_map(ProveWs *w, bool use_graph = false) {
        int rc;
        if (use_graph) {
            MG_HIP(hipGraphLaunch(w->g_wm, w->stream));
        } else if ((rc = enqueue_witness_map_body(w))) {
            return rc;
        }
        MG_HIP(hipEventRecord(w->h_ready, w->stream));
        return MG_OK;
    }

    int witness_map_host(const uint64_t *z, uint64_t *h_out) override {
        DeviceGuard restore_callers_device;
        MG_HIP(hipSetDevice(dev_));
        std::shared_lock<std::shared_mutex> shape_lock(shape_mu_);
        if (!have_r1cs_) return MG_ERR_STATE;
        ProveWs *w = ws_acquire();
        if (!w) return MG_ERR_HIP;
        int rc = upload_z(w, z);
        if (!rc) rc = launch_witness_map(w);
        if (!rc) {
            const size_t D = (size_t)1 << log_d_;
            std::vector<uint64_t> tmp(D * 4);
            u32 *d_std = nullptr; // h leaves the pipeline in the work form: convert for the host
            hipError_t e = hipMalloc((void **)&d_std, D * 32);
            if (e == hipSuccess && fr_->work_to_std(w->a.as<u32>(), D, d_std, w->stream)) e = hipErrorUnknown;
            if (e == hipSuccess) e = hipMemcpyAsync(tmp.data(), d_std, D * 32, hipMemcpyDeviceToHost, w->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(w->stream);
            if (d_std) hipFree(d_std);
            if (e != hipSuccess) {
                set_last_hip_error(e, "witness_map_host", __FILE__, __LINE__);
                rc = MG_ERR_HIP;
            } else { // the device keeps h bit-reversed; the API returns natural order like witness_map
                for (size_t p = 0; p < D; ++p) {
                    size_t src = 0;
                    for (unsigned b = 0; b < log_d_; ++b) src |= ((p >> b) & 1) << (log_d_ - 1 - b);
                    std::memcpy(h_out + src * 4, &tmp[p * 4], 32);
                }
            }
        } else {
            hipStreamSynchronize(w->stream);
        }
        ws_release(w);
        return rc;
    }

    struct MsmArgs {
        const BaseSet *bs[5];
        const u32 *sc[5];
        size_t cnt[5], stride[5];
    };
    MsmArgs msm_args(const ProveWs *w) const {
        const size_t D = (size_t)1 << log_d_;
        const u32 *dz = w->z.as<u32>();
        // h and the h-query bases are both bit-reversed; bases beyond len(h_query) are infinity
        // (multi_scalar_mul zips to the shorter; the dropped coefficient h[D-1] is zero)
        // batched passes switch to the wide-window tables (fewer mixed additions, longer bucket reduce) from this many proofs
        // on: a pass of a few coalesced single calls is still a latency chain and keeps the narrow ones (MANTA_WIDE_MIN)
        static const u32 wide_min = [] {
            const int v = ab_knob("MANTA_WIDE_MIN", 4);
            return (u32)(v >= 1 && v <= 64 ? v : 4);
        }();
        const bool wide = w->k >= wide_min;
        // a range shard multiplies its contiguous slice of every query by the matching slice of the scalars
        const size_t zlo = shard_lo(V_ - 1), zn = shard_hi(V_ - 1) - zlo, llo = shard_lo(V_ - P_), ln = shard_hi(V_ - P_) - llo;
        const size_t hlo = shard_lo(D), hn = shard_hi(D) - hlo;
        const u32 *sz = dz + (1 + zlo) * 8;
        // passes of up to this many proofs run on the full tables where the key has them (MANTA_FULL_MAX_K; a batch sorts its
        // pairs by proof -- one radix pass -- and needs 32 additions per scalar where the wide bucket tables need 24)
        static const u32 full_max_k = [] {
            const int v = ab_knob("MANTA_FULL_MAX_K", 1);
            return (u32)(v >= 0 ? v : 1);
        }();
        const bool one = w->k <= full_max_k;
        auto pick = [&](BaseSet *full, BaseSet *wd, BaseSet *narrow) { return one && full ? full : (wide && wd ? wd : narrow); };
        return MsmArgs{{w->z3 ? z3_bs_full_ : pick(a_bs_full_, a_bs_wide_, a_bs_), pick(b1_bs_full_, b1_bs_wide_, b1_bs_), pick(b2_bs_full_, b2_bs_wide_, b2_bs_),
                        pick(l_bs_full_, l_bs_wide_, l_bs_), pick(h_bs_full_, h_bs_wide_, h_bs_)},
                       {sz, sz, sz, dz + ((size_t)P_ + llo) * 8, w->a.as<u32>() + hlo * (size_t)fr_->work_words()},
                       {zn, zn, zn, ln, hn},
                       {(size_t)V_ * 8, (size_t)V_ * 8, (size_t)V_ * 8, (size_t)V_ * 8, D * (size_t)fr_->work_words()}};
    }
    // does this slot launch MSM i (a, b_g1, b_g2, l, h)? -- not another rank's (task placement), not folded into the combined one
    bool runs(const ProveWs *w, int i) const { return does(i) && !(w->z3 && (i == 1 || i == 3)); }
    bool wants_z3(u32 k) const {
        static const u32 full_max_k = [] {
            const int v = ab_knob("MANTA_FULL_MAX_K", 1);
            return (u32)(v >= 0 ? v : 1);
        }();
        // (passes of one proof only: on full tables passes of 2-8 proofs are SLOWER than on the narrow bucket tables -- 2.65 against 1.9 ms
        // for two, six signer threads 1 300 against 1 650 proofs/s -- measured with MANTA_FULL_MAX_K = 4 / 8, round 4)
        return z3_bs_full_ && k == 1 && full_max_k >= 1 && peers_.empty() && !ex_;
    }
    static hipStream_t msm_stream(const ProveWs *w, int i) { return w->mw[i]->run_on ? w->mw[i]->run_on : w->mw[i]->stream; }

    // The GPU side of a pass is two independent pieces that only share the uploaded assignment:
    //   part A, on w->stream: witness map, then the four G1 MSMs (a, b_g1, l from z; h from the witness map)
    //           forked onto their streams with events and joined back;
    //   part B, on the G2 MSM's stream: the G2 MSM -- the longest chain of a proof.
    // The host waits for part A first and does the G1 half of the assembly (s*A + r*B1 is ~0.15 ms of host work)
    // while part B is still running. use_graphs replays the per-stream graphs of the "split" mode instead of
    // enqueuing kernels; the event structure is identical.
    static bool in_part_a(int i) { return i != 2; }
    int enqueue_msm(ProveWs *w, const MsmArgs &a, int i, bool use_graphs) {
        if (use_graphs) {
            hipStream_t ms = msm_stream(w, i);
            MG_HIP(hipGraphLaunch(w->g_msm[i], ms));
            MG_HIP(hipEventRecord(w->mw[i]->done, ms));
            w->mw[i]->pending = 1;
            return MG_OK;
        }
        // the z MSMs see witness scalars (mostly 0 / 1 / small): compact their zero digits; h is dense
        if (w->timed) MG_HIP(hipEventRecord(w->tev[3 + 2 * i], msm_stream(w, i)));
        const int rc = w->me[i]->msm_launch(a.bs[i], a.sc[i], a.cnt[i], i == 4 ? SCALARS_WORK : SCALARS_MONT, 0, w->mw[i], w->k, a.stride[i], i != 4);
        if (w->timed && !rc) MG_HIP(hipEventRecord(w->tev[4 + 2 * i], msm_stream(w, i)));
        return rc;
    }
    int enqueue_part_a(ProveWs *w, bool use_graphs) {
        int rc;
        const MsmArgs a = msm_args(w);
        MG_HIP(hipEventRecord(w->fork, w->stream)); // z is on the device (upload_z ran on this stream)
        if (w->timed) MG_HIP(hipEventRecord(w->tev[1], w->stream));
        if (does(4) && (rc = launch_witness_map(w, use_graphs))) return rc; // h is only needed by the h MSM
        if (w->timed) MG_HIP(hipEventRecord(w->tev[2], w->stream));
        for (int i = 0; i < 5; ++i) {
            if (!in_part_a(i) || !runs(w, i)) continue;
            hipStream_t ms = msm_stream(w, i);
            // (round 5, measured and dropped: for LARGE proofs -- 2^20 variables -- the a / b_g1 / l MSMs launched BEHIND the witness
            // map instead of beside it: the witness map falls from 5.9 to 3.8 ms and each of the three MSMs from 4-7 to 2-3 ms, but the
            // proof goes from 10.5 to 11.0 ms -- the chip is busy either way: profiles/r05_config2_ab.txt)
            if (ms != w->stream) MG_HIP(hipStreamWaitEvent(ms, i == 4 ? w->h_ready : w->fork, 0));
            if ((rc = enqueue_msm(w, a, i, use_graphs))) return rc;
        }
        for (int i = 0; i < 5; ++i) { // join (after every launch, so that no MSM on the main stream queues behind a wait)
            hipStream_t ms = msm_stream(w, i);
            if (in_part_a(i) && runs(w, i) && ms != w->stream) MG_HIP(hipStreamWaitEvent(w->stream, w->mw[i]->done, 0));
        }
        if (w->timed) MG_HIP(hipEventRecord(w->tev[13], w->stream));
        return MG_OK;
    }
    int enqueue_part_b(ProveWs *w, bool use_graphs) { return does(2) ? enqueue_msm(w, msm_args(w), 2, use_graphs) : MG_OK; }

    int enqueue_proof(ProveWs *w, const uint64_t *z_src, bool use_graphs) {
        if (w->timed) MG_HIP(hipEventRecord(w->tev[0], w->stream));
        int rc = upload_z(w, z_src);
        if (rc) return rc;
        hipStream_t g2s = msm_stream(w, 2);
        if (w->timed) { // eager launches with events between the phases
            if ((rc = enqueue_part_a(w, false))) return rc;
            if (g2s != w->stream) MG_HIP(hipStreamWaitEvent(g2s, w->z_ready, 0));
            if ((rc = enqueue_part_b(w, false))) return rc;
            MG_HIP(hipEventRecord(w->tev[14], g2s));
            return MG_OK;
        }
        if (w->linear3 && w->g_all && w->g_g2 && w->g_msm[0]) { // three linear graphs, each behind the upload
            hipStream_t z3s = msm_stream(w, 0);
            // launch order (MANTA_Z3_ORDER, three letters of a = witness map + h, b = G2, z = combined): the chain that bounds the
            // proof first -- each hipGraphLaunch is 10-20 us of host time, which the chains launched later start behind
            static const char *order = [] {
                const char *e = ab_knob_str("MANTA_Z3_ORDER", "abz");
                return std::strlen(e) == 3 ? e : "abz";
            }();
            for (int t = 0; t < 3; ++t) {
                if (order[t] == 'a') {
                    MG_HIP(hipGraphLaunch(w->g_all, w->stream));
                } else if (order[t] == 'b') {
                    if (g2s != w->stream) MG_HIP(hipStreamWaitEvent(g2s, w->z_ready, 0));
                    MG_HIP(hipGraphLaunch(w->g_g2, g2s));
                } else {
                    if (z3s != w->stream) MG_HIP(hipStreamWaitEvent(z3s, w->z_ready, 0));
                    MG_HIP(hipGraphLaunch(w->g_msm[0], z3s));
                }
            }
            for (int i = 0; i < 5; ++i) w->mw[i]->pending = runs(w, i) ? 1 : 0;
            return MG_OK;
        }
        if (w->g_all && w->g_g2) { // "single" mode replay
            // the G2 graph goes first: it is the longest chain and its launch is the cheaper of the two
            // (measured: 1.47 ms per PrivateTransfer proof against 1.68 with the other order)
            if (g2s != w->stream) MG_HIP(hipStreamWaitEvent(g2s, w->z_ready, 0));
            MG_HIP(hipGraphLaunch(w->g_g2, g2s));
            MG_HIP(hipGraphLaunch(w->g_all, w->stream));
            for (int i = 0; i < 5; ++i) w->mw[i]->pending = runs(w, i) ? 1 : 0;
            return MG_OK;
        }
        if ((rc = enqueue_part_a(w, use_graphs))) return rc;
        if (g2s != w->stream) MG_HIP(hipStreamWaitEvent(g2s, w->z_ready, 0));
        return enqueue_part_b(w, use_graphs);
    }

    // capture one single-stream segment into an executable graph
    // why the last failed capture_segment of this thread failed: true = the capture itself was invalidated / the stream cannot
    // capture (streams that joined it are not trusted again), false = a deterministic failure (instantiation, out of memory)
    static bool &capture_invalidated() {
        static thread_local bool v = false;
        return v;
    }
    template <class Fn> static bool capture_segment(hipStream_t s, hipGraphExec_t *out, Fn &&body) {
        capture_invalidated() = false;
        if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) != hipSuccess) {
            capture_invalidated() = true; // (still capturing / invalidated from an earlier failure)
            (void)hipGetLastError();
            return false;
        }
        const int rc = body();
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &st) != hipSuccess || st != hipStreamCaptureStatusActive) capture_invalidated() = true;
        hipGraph_t graph = nullptr;
        const hipError_t e = hipStreamEndCapture(s, &graph);
        if (e == hipErrorStreamCaptureInvalidated || e == hipErrorStreamCaptureUnjoined || e == hipErrorStreamCaptureUnmatched ||
            e == hipErrorStreamCaptureWrongThread || e == hipErrorStreamCaptureImplicit)
            capture_invalidated() = true;
        bool ok = !rc && e == hipSuccess && graph && hipGraphInstantiate(out, graph, nullptr, nullptr, 0) == hipSuccess;
        if (graph) hipGraphDestroy(graph);
        if (!ok) {
            *out = nullptr;
            (void)hipGetLastError();
        }
        return ok;
    }
    // every buffer has its final size (two eager runs): capture the witness map and the five MSMs
    bool build_graphs(ProveWs *w) {
        // exclusive side of the capture lock, but never WAITED for at once: a context creation (seconds of table precompute on the
        // shared side) or a stream of stand-alone MSM calls would stall this proving thread with its shape lock held (and libstdc++'s
        // shared_mutex prefers readers). The pass runs eagerly instead and the capture is retried on a later pass; after
        // CAPTURE_TRIES such passes it waits.
        std::unique_lock<std::shared_mutex> no_heavy_ops_meanwhile(capture_mutex(), std::try_to_lock);
        if (!no_heavy_ops_meanwhile.owns_lock()) {
            static const int tries = ab_knob("MANTA_CAPTURE_TRIES", CAPTURE_TRIES);
            if (++w->capture_tries < tries) return false;
            no_heavy_ops_meanwhile.lock();
        }
        w->capture_tries = 0;
        bool invalidated = false;
        const bool ok = build_graphs_locked(w, invalidated);
        if (!ok) {
            if (invalidated) {
                // streams that joined an invalidated capture are not trusted again: the slot is destroyed after this pass, its
                // streams and its workspaces' streams abandoned (~ProveWs)
                w->poisoned = true;
            } else {
                // a deterministic failure (instantiation error, out of memory): destroying the slot would only repeat two eager
                // passes, every hipMalloc and the failure on each call -- this kind of slot stays eager, here and in later slots
                std::lock_guard<std::mutex> g(mu_);
                no_graph_keys_.insert(slot_key(w->k, w->z3, w->flavour));
            }
        }
        return ok;
    }
    bool build_graphs_locked(ProveWs *w, bool &invalidated) {
        if (w->linear3) {
            const MsmArgs a = msm_args(w);
            w->mw[4]->capturing = true; // linear captures: nothing inside them waits on a `done` event
            bool ok = capture_segment(w->stream, &w->g_all, [&] {
                const int rc = enqueue_witness_map_body(w);
                return rc ? rc : enqueue_msm(w, a, 4, false);
            });
            w->mw[4]->capturing = false;
            if (ok) {
                w->mw[0]->capturing = true;
                ok = capture_segment(msm_stream(w, 0), &w->g_msm[0], [&] { return enqueue_msm(w, a, 0, false); });
                w->mw[0]->capturing = false;
            }
            if (ok) {
                w->mw[2]->capturing = true;
                ok = capture_segment(msm_stream(w, 2), &w->g_g2, [&] { return enqueue_part_b(w, false); });
                w->mw[2]->capturing = false;
            }
            for (int i = 0; i < 5; ++i) w->mw[i]->pending = 0;
            if (!ok) {
                invalidated = capture_invalidated(); // (of the segment that failed: the chain stops at the first failure)
                w->drop_graphs();
                w->no_graph = true;
            }
            w->graphs_ready = ok;
            return ok;
        }
        if (graph_mode_for(w->k) == GRAPH_SINGLE) {
            // (Round 4, measured and withdrawn: the combined MSM of a z3 slot captured as a LINEAR graph of its own and replayed next to
            // the G2 one started with the upload instead of 210-290 us into the proof and was worth 2-3 % of a sequential proof -- but
            // with other contexts' passes in flight on the GPU the proof's C element came out WRONG, on a pooled high-priority
            // stream as on the workspace's own (test_rccl_branch_with_a_one_rank_group caught it; the branch form below is right under
            // the same load). Three graphs per proof are not worth an unexplained dependency on the runtime's graph executor.)
            bool ok1 = capture_segment(w->stream, &w->g_all, [&] { return enqueue_part_a(w, false); });
            if (ok1) {
                w->mw[2]->capturing = true; // a linear capture: nothing waits on its `done` event
                ok1 = capture_segment(msm_stream(w, 2), &w->g_g2, [&] { return enqueue_part_b(w, false); });
                w->mw[2]->capturing = false;
            }
            for (int i = 0; i < 5; ++i) w->mw[i]->pending = 0;
            if (!ok1) {
                invalidated = capture_invalidated();
                w->drop_graphs();
                w->no_graph = true;
            }
            w->graphs_ready = ok1;
            return ok1;
        }
        bool ok = capture_segment(w->stream, &w->g_wm, [&] { return enqueue_witness_map_body(w); });
        const MsmArgs a = msm_args(w);
        for (int i = 0; ok && i < 5; ++i) {
            if (!runs(w, i)) continue;
            w->mw[i]->capturing = true; // no event records inside the capture: the replay path records `done`
            ok = capture_segment(msm_stream(w, i), &w->g_msm[i], [&] {
                return w->me[i]->msm_launch(a.bs[i], a.sc[i], a.cnt[i], i == 4 ? SCALARS_WORK : SCALARS_MONT, 0, w->mw[i], w->k, a.stride[i], i != 4);
            });
            w->mw[i]->capturing = false;
            w->mw[i]->pending = 0;
        }
        if (!ok) {
            invalidated = capture_invalidated();
            w->drop_graphs();
            w->no_graph = true;
        }
        w->graphs_ready = ok;
        return ok;
    }

    // ---- one proof. Concurrent callers on one context are COALESCED: while COALESCE_INFLIGHT passes are on the GPU, further
    // calls queue up, and the next caller to find a pass slot free takes everything queued (up to BATCH_CHUNK) as ONE batched
    // pass -- the wallet / ledger simulation of the reference drives one ProvingContext from six threads
    // (manta-pay/src/bin/simulation.rs:36-38, simulation/mod.rs:75-79), each proving one transfer at a time; measured on
    // MI355X, PrivateTransfer shape, single calls from 1 / 2 / 3 / 4 / 6 threads without coalescing: 964 / 1 254 / 1 130 /
    // 1 108 / 1 068 proofs/s (the passes only share the GPU's queues), against ~3 900 for explicit batches. A lone caller
    // is never delayed (it leads a pass of one at once); batch sizes are rounded up to a power of two by repeating the first
    // request (slots and captured graphs exist per batch size), the surplus proofs are dropped. Proof bytes do not depend
    // on how calls were grouped. Sharded contexts and MANTA_COALESCE=0 take the direct path.
    struct Req {
        const uint64_t *z, *r, *s;
        uint8_t *out;
        int rc = MG_OK;
        bool done = false;
    };
    std::mutex cq_mu_;
    std::condition_variable cq_cv_;
    std::deque<Req *> cq_;
    int cq_inflight_ = 0, cq_batched_inflight_ = 0; // passes of this context's coalescing queue on the GPU; those of more than one proof
    bool cq_gathering_ = false; // a leader is waiting for the callers of the pass that has just finished
    size_t cq_last_k_ = 1;      // size of the most recently finished pass
    int prove(const uint64_t *z, const uint64_t *r, const uint64_t *s, uint8_t *proof_out) override {
        if (!z || !r || !s || !proof_out) return MG_ERR_ARG;
        if (task_mask_ != 0x1f) return MG_ERR_STATE; // holds some of the MSMs only: partials_launch / assemble
        if (lone_range_shard()) return MG_ERR_STATE; // one slice of every query, the others live in other processes: same
        if (coalesce_inflight() == 0 || !peers_.empty() || ex_) return prove_pass(1, z, r, s, proof_out);
        Req me{z, r, s, proof_out};
        std::unique_lock<std::mutex> lk(cq_mu_);
        cq_.push_back(&me);
        if (cq_gathering_) cq_cv_.notify_all(); // a leader is collecting arrivals
        for (;;) {
            if (me.done) return me.rc;
            if (!cq_gathering_ && cq_inflight_ < coalesce_inflight() && !cq_.empty()) { // lead a pass: everything queued, oldest first
                // The callers of a pass that has just finished come back one after the other within a few tens of microseconds.
                // While ANOTHER pass keeps the GPU busy nothing is lost by letting them all arrive: without this the first one
                // back led a pass of one and the rest followed as a pass of two -- six signer threads then ran as passes of
                // 1 + 2 + 3 instead of 3 + 3. Only when a pass is in flight, only up to the size of the last finished pass, at
                // most coalesce_gather_us(): a lone caller, or callers that never overlapped, are not delayed.
                if (cq_inflight_ >= 1 && cq_.size() < cq_last_k_ && coalesce_gather_us() > 0) {
                    cq_gathering_ = true;
                    const auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(coalesce_gather_us());
                    while (cq_.size() < cq_last_k_ && cq_inflight_ >= 1 && // (the other pass may finish meanwhile: then go at once)
                           cq_cv_.wait_until(lk, deadline) != std::cv_status::timeout) {
                    }
                    cq_gathering_ = false;
                    if (me.done) { // (cannot happen while this thread gathers -- nobody else leads -- but stay safe)
                        cq_cv_.notify_all();
                        return me.rc;
                    }
                }
                std::vector<Req *> batch;
                while (!cq_.empty() && batch.size() < BATCH_CHUNK) {
                    batch.push_back(cq_.front());
                    cq_.pop_front();
                }
                // what this pass runs beside: 0 nothing of this context on the GPU right now, 1 single proofs only, 2 a batched pass
                const int company = cq_inflight_ == 0 ? 0 : (cq_batched_inflight_ == 0 ? 1 : 2);
                const bool batched = batch.size() > 1;
                ++cq_inflight_;
                if (batched) ++cq_batched_inflight_;
                lk.unlock();
                int rc; // nothing may escape here: the followers of this batch wait on cq_cv_ for their `done`
                try {
                    rc = prove_gathered(batch, company);
                } catch (const std::bad_alloc &) {
                    rc = MG_ERR_OOM;
                } catch (...) {
                    rc = MG_ERR_STATE;
                }
                lk.lock();
                for (Req *q : batch) q->rc = rc, q->done = true;
                --cq_inflight_;
                if (batched) --cq_batched_inflight_;
                cq_last_k_ = batch.size();
                cq_cv_.notify_all();
                continue;
            }
            cq_cv_.wait(lk);
        }
    }
    // one pass over the requests of `batch` (padded to a power of two with copies of the first one)
    int prove_gathered(const std::vector<Req *> &batch, int company = 0) {
        const size_t k = batch.size();
        if (k == 1) return prove_pass(1, batch[0]->z, batch[0]->r, batch[0]->s, batch[0]->out, nullptr, company);
        // pass sizes: exact up to 8 (a pass of k proofs costs ~0.65 + 0.42 k ms for the PrivateTransfer shape -- padding three
        // coalesced calls to four wastes a sixth of the pass; six signer threads produce passes of two to four), then multiples of
        // four: slots and their captured graphs exist per size, so the set of sizes stays small
        size_t kp = k <= 8 ? k : (k + 3) / 4 * 4;
        if (ab_knob("MANTA_COALESCE_POW2", 0) > 0) // round up to a power of two (the round-2 rule)
            for (kp = 1; kp < k;) kp <<= 1;
        const size_t pbytes = 2 * (size_t)g1_->point_bytes(true) + (size_t)g2_->point_bytes(true);
        std::vector<const uint64_t *> zl(kp);
        std::vector<uint64_t> rr(kp * 4), ss(kp * 4);
        std::vector<uint8_t> out(kp * pbytes);
        for (size_t q = 0; q < kp; ++q) {
            const Req *src = batch[q < k ? q : 0];
            zl[q] = src->z;
            std::memcpy(&rr[q * 4], src->r, 32);
            std::memcpy(&ss[q * 4], src->s, 32);
        }
        const int rc = prove_pass(kp, nullptr, rr.data(), ss.data(), out.data(), zl.data());
        if (!rc)
            for (size_t q = 0; q < k; ++q) std::memcpy(batch[q]->out, &out[q * pbytes], pbytes);
        return rc;
    }

    // k proofs of this circuit. Up to BATCH_CHUNK of them are ONE pass of the GPU pipeline (k = 1: a single proof): the
    // kernels are the same; every (assignment, window) pair is its own bucket segment and the NTT / SpMV grids get a
    // batch dimension, so a pass costs one chain of latency-bound launches instead of k. A longer batch is streamed
    // through as passes of BATCH_CHUNK with BATCH_INFLIGHT of them in flight on their own slots (library threads): the
    // witness-map head and the bucket-reduce tail of one pass then overlap the accumulate kernels of the others --
    // measured on MI355X for the PrivateTransfer shape, passes of 32: 2 280 proofs/s with one pass in flight, 3 350 with
    // two, 3 820 with three, 3 680 with four; passes of 16 or 64 are no better. (Splitting ONE pass of 32 into two
    // halves in flight was measured too and gains nothing: the smaller passes lose what the overlap wins.)
    static constexpr u64 BATCH_CHUNK = 32;
    int prove_batch(u64 k64, const uint64_t *z, const uint64_t *r, const uint64_t *s, uint8_t *proofs_out) override {
        if (k64 == 0 || k64 > 1024 || !z || !r || !s || !proofs_out) return MG_ERR_ARG;
        if (task_mask_ != 0x1f || lone_range_shard()) return MG_ERR_STATE;
        if (k64 <= BATCH_CHUNK) return prove_pass(k64, z, r, s, proofs_out);
        // (passes of exactly BATCH_CHUNK proofs plus one remainder: equalising the pass sizes -- 256 proofs as 9 x 29 instead
        // of 8 x 32 -- was measured and is slower: every distinct pass size needs its own workspaces and captured graphs)
        const u64 fl = (u64)batch_inflight(), per = BATCH_CHUNK;
        const u64 chunks = (k64 + per - 1) / per;
        const size_t pbytes = 2 * (size_t)g1_->point_bytes(true) + (size_t)g2_->point_bytes(true); // compressed A, B, C
        std::atomic<u64> next{0};
        std::atomic<int> first_rc{MG_OK};
        auto worker = [&]() noexcept {
            for (;;) {
                const u64 c = next.fetch_add(1);
                if (c >= chunks || first_rc.load() != MG_OK) return;
                const u64 lo = c * per, n = std::min(per, k64 - lo);
                stream_gate_acquire(); // at most `fl` streamed passes in flight per context, however many callers
                int rc;
                try {
                    rc = prove_pass(n, z + lo * V_ * 4, r + lo * 4, s + lo * 4, proofs_out + lo * pbytes);
                } catch (const std::bad_alloc &) {
                    rc = MG_ERR_OOM;
                } catch (...) {
                    rc = MG_ERR_STATE;
                }
                stream_gate_release();
                int ok = MG_OK;
                if (rc) first_rc.compare_exchange_strong(ok, rc);
            }
        };
        const int nthreads = (int)std::min<u64>(fl, chunks);
        {
            std::vector<std::thread> th;
            JoinAll guard{th};
            try {
                for (int t = 1; t < nthreads; ++t) th.emplace_back(worker);
            } catch (...) { // a helper thread could not be started: the caller works through the chunks alone
            }
            worker();
        }
        return first_rc.load();
    }
    // two callers streaming a batch each would otherwise put six passes in flight, which is slower than three (measured:
    // 2 x 256 proofs 3 465 proofs/s against 3 650-3 840 for one caller)
    std::mutex gate_mu_;
    std::condition_variable gate_cv_;
    int gate_busy_ = 0;
    void stream_gate_acquire() {
        std::unique_lock<std::mutex> lk(gate_mu_);
        gate_cv_.wait(lk, [&] { return gate_busy_ < batch_inflight(); });
        ++gate_busy_;
    }
    void stream_gate_release() {
        {
            std::lock_guard<std::mutex> lk(gate_mu_);
            --gate_busy_;
        }
        gate_cv_.notify_one();
    }
    int prove_pass(u64 k64, const uint64_t *z, const uint64_t *r, const uint64_t *s, uint8_t *proofs_out,
                   const uint64_t *const *z_list = nullptr, int company = 0) {
        DeviceGuard restore_callers_device; // the pass visits every shard's device
        // shared against set_r1cs on every shard for the length of the pass
        std::vector<std::shared_lock<std::shared_mutex>> locks;
        locks.emplace_back(shape_mu_);
        for (ProverImpl *q : peers_) locks.emplace_back(q->shape_mu_);
        if (!have_r1cs_) return MG_ERR_STATE;
        if (ex_) return prove_pass_rccl((u32)k64, z, r, s, proofs_out, z_list); // the partial points meet through RCCL
        // every shard gets the whole assignment (1.1 MB for PrivateTransfer) and recomputes the witness map -- cheaper
        // than broadcasting h -- then multiplies its slices; shard 0 launches last and assembles
        std::vector<Pass> pp(peers_.size());
        int rc = MG_OK;
        for (size_t g = 0; g < peers_.size() && !rc; ++g) rc = peers_[g]->launch_pass(pp[g], (u32)k64, z, r, s, nullptr);
        Pass p;
        const auto t_enq = std::chrono::steady_clock::now();
        if (!rc) rc = launch_pass(p, (u32)k64, z, r, s, proofs_out, z_list, true, company);
        p.enqueue_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_enq).count();
        return finish_pass(p, rc, &pp);
    }

    struct Pass {
        ProveWs *w = nullptr;
        u32 k = 0;
        const uint64_t *r = nullptr, *s = nullptr;
        uint8_t *out = nullptr;
        float enqueue_ms = 0.f;
        bool z3_folded = false; // the combined MSM's results were taken before the rest of part A (finish_pass_body)
    };

    static bool is_page_locked(const void *p) {
        hipPointerAttribute_t a;
        if (hipPointerGetAttributes(&a, p) != hipSuccess) {
            (void)hipGetLastError(); // ordinary pageable memory is "invalid value" to the runtime
            return false;
        }
        return a.type == hipMemoryTypeHost;
    }

    // stage z, enqueue (or replay) the GPU side of k proofs on a slot; returns without waiting
    // (z_list: the k assignments as separate buffers -- coalesced single calls -- gathered into the slot's staging copy)
    int launch_pass(Pass &p, u32 k, const uint64_t *z, const uint64_t *r, const uint64_t *s, uint8_t *out,
                    const uint64_t *const *z_list = nullptr, bool whole_proof = true, int company = 0) {
        MG_HIP(hipSetDevice(dev_));
        p.k = k, p.r = r, p.s = s, p.out = out;
        // (the partials interface folds every MSM on the device by its index: it keeps the five separate MSMs)
        ProveWs *w = p.w = ws_acquire(k, whole_proof && wants_z3(k), company);
        if (!w) return MG_ERR_HIP;
        int rc = MG_OK;
        const size_t zbytes = (size_t)k * V_ * 32;
        // the assignment is uploaded from where it is if the caller keeps it in page-locked memory
        // (mg_host_alloc), else through the slot's pinned staging copy
        const uint64_t *z_src = z;
        w->z_parts.clear();
        bool stage = !z_list && !is_page_locked(z);
        if (z_list) { // page-locked assignments are uploaded in place, the others through their part of the staging copy
            w->z_parts.assign(z_list, z_list + k);
            std::vector<char> pageable(k, 0);
            for (u32 q = 0; q < k; ++q) {
                bool seen = false;
                for (u32 t = 0; t < q && !seen; ++t)
                    if (z_list[t] == z_list[q]) pageable[q] = pageable[t], seen = true; // (padding repeats request 0)
                if (!seen) pageable[q] = !is_page_locked(z_list[q]);
                stage = stage || pageable[q];
            }
            if (stage) {
                if (w->h_z_cap < zbytes) {
                    HeavyOp pinned_allocation_not_beside_a_capture;
                    if (w->h_z) hipHostFree(w->h_z);
                    w->h_z = nullptr;
                    w->h_z_cap = 0;
                    if (hipHostMalloc(&w->h_z, zbytes, hipHostMallocDefault) != hipSuccess) return MG_ERR_OOM;
                    w->h_z_cap = zbytes;
                }
                for (u32 q = 0; q < k; ++q)
                    if (pageable[q]) {
                        std::memcpy((char *)w->h_z + (size_t)q * V_ * 32, z_list[q], (size_t)V_ * 32);
                        w->z_parts[q] = (const uint64_t *)((char *)w->h_z + (size_t)q * V_ * 32);
                    }
            }
            stage = false;
        }
        if (stage) {
            if (w->h_z_cap < zbytes) {
                HeavyOp pinned_allocation_not_beside_a_capture;
                if (w->h_z) hipHostFree(w->h_z);
                w->h_z = nullptr;
                w->h_z_cap = 0;
                if (hipHostMalloc(&w->h_z, zbytes, hipHostMallocDefault) != hipSuccess) return MG_ERR_OOM;
                w->h_z_cap = zbytes;
            }
            std::memcpy(w->h_z, z, zbytes);
            z_src = (const uint64_t *)w->h_z;
        }
        w->timed = false;
        if (task_mask_ != 0x1f) w->no_graph = true; // a subset of the MSMs: plain launches
        if (kernel_timing() && k == 1 && peers_.empty() && task_mask_ == 0x1f) {
            bool ok = true;
            for (auto &e : w->tev)
                if (!e) ok = ok && hipEventCreate(&e) == hipSuccess;
            w->timed = ok;
        }
        if (!w->graphs_ready && graph_mode_for(w->k) != GRAPH_OFF && !w->no_graph && w->eager_runs >= 2 && !w->timed) build_graphs(w);
        if (w->mw[0]->notify && w->mw[0]->h_flag) { // the combined MSM's end-of-chain token of THIS pass (z3 slots)
            *(volatile u32 *)w->mw[0]->h_flag = 0;
            std::atomic_thread_fence(std::memory_order_seq_cst);
        }
        if (w->timed) {
            HeavyOp eager_passes_allocate; // (see below)
            rc = enqueue_proof(w, z_src, false);
        } else if (w->graphs_ready) {
            rc = enqueue_proof(w, z_src, graph_mode_for(w->k) == GRAPH_SPLIT);
            if (rc) { // do not trust the graphs again; the failed pass is reported to the caller
                hipStreamSynchronize(w->stream);
                for (int i = 0; i < 5; ++i) hipStreamSynchronize(msm_stream(w, i));
                w->drop_graphs();
                w->no_graph = true;
            }
        } else {
            // An eager pass sizes the slot's buffers (hipMalloc / hipFree / hipHostMalloc inside msm_launch and the witness map):
            // like context creation it must not run while another thread captures (capture_mutex, engine.h) -- with contexts
            // recycled every 2 s the soak saw 70 000 calls fail on a slot whose first pass had met a capture
            HeavyOp eager_passes_allocate;
            rc = enqueue_proof(w, z_src, false);
            w->eager_runs++;
        }
        return rc;
    }

    // host side of a pass: blinding terms while the GPU works, wait, fold the MSM results, assemble and encode
    // wait for one part of a pass on this shard and fold its MSMs: res[i * k + q] = MSM i of proof q
    int collect_part(Pass &p, bool part_a, HostPoint *res) {
        ProveWs *w = p.w;
        if (!w) return MG_ERR_HIP;
        int rc = MG_OK;
        hipSetDevice(dev_);
        hipError_t e = hipStreamSynchronize(part_a ? w->stream : msm_stream(w, 2));
        if (e != hipSuccess) {
            set_last_hip_error(e, "prove: hipStreamSynchronize", __FILE__, __LINE__);
            rc = MG_ERR_HIP;
        }
        for (int i = 0; i < 5; ++i) {
            if (in_part_a(i) != part_a) continue;
            if (w->z3 && (i == 1 || i == 3)) continue; // part of the combined MSM on mw[0]
            if (w->z3 && i == 0 && p.z3_folded) { // finish_pass_body took its results when its chain ended
                // (a linear3 slot: the combined MSM's graph runs on a stream nothing joins; its end-of-chain token has been seen, but
                // the RUNTIME only retires a launch when its stream is waited on -- without this the stream is never synchronised
                // in the slot's whole life and its graph exec could be destroyed with launches the runtime still tracks)
                if (w->linear3) (void)hipStreamSynchronize(msm_stream(w, 0));
                continue;
            }
            if (w->z3 && i == 0 && w->mw[0]->pending) { // three results per proof: a, b_g1, l
                if (w->linear3) { // (its graph runs on a stream of its own, joined by nothing)
                    const hipError_t e3 = hipStreamSynchronize(msm_stream(w, 0));
                    if (e3 != hipSuccess && !rc) {
                        set_last_hip_error(e3, "prove: hipStreamSynchronize", __FILE__, __LINE__);
                        rc = MG_ERR_HIP;
                    }
                }
                std::vector<HostPoint> t3((size_t)3 * p.k);
                int rc2 = w->me[0]->msm_finish(w->mw[0], t3.data(), true);
                if (!rc) rc = rc2;
                for (u32 q = 0; q < p.k; ++q) {
                    res[(size_t)0 * p.k + q] = t3[(size_t)q * 3 + 0];
                    res[(size_t)1 * p.k + q] = t3[(size_t)q * 3 + 1];
                    res[(size_t)3 * p.k + q] = t3[(size_t)q * 3 + 2];
                }
                continue;
            }
            if (w->mw[i]->pending) {
                int rc2 = w->me[i]->msm_finish(w->mw[i], res + (size_t)i * p.k, true);
                if (!rc) rc = rc2;
            } else {
                hipStreamSynchronize(msm_stream(w, i));
                if (!rc) rc = MG_ERR_STATE;
            }
        }
        return rc;
    }
    void abandon_pass(Pass &p) { // a pass that will not be assembled: drain its streams, return the slot
        if (!p.w) return;
        hipSetDevice(dev_);
        hipStreamSynchronize(p.w->stream);
        for (int i = 0; i < 5; ++i) {
            hipStreamSynchronize(msm_stream(p.w, i));
            p.w->mw[i]->pending = 0;
        }
        ws_release(p.w);
        p.w = nullptr;
    }

    // The host side of a pass is ~0.15 ms per proof (two 254-bit scalar multiplications in one doubling chain, four table
    // multiplications, three serialisations with a field inversion each): nothing next to a single proof, but 5 ms of a
    // 32-proof pass whose GPU side is 8.6 ms. Batches spread it over up to four library threads.
    // (exception-safe: a worker that throws -- bad_alloc -- is caught in its own thread, every thread is joined, and the
    // failure is rethrown on the calling thread, where the C ABI turns it into a status code; a thread that cannot be
    // started just leaves its share to the caller)
    struct JoinAll {
        std::vector<std::thread> &th;
        ~JoinAll() {
            for (auto &t : th)
                if (t.joinable()) t.join();
        }
    };
    template <class Fn> static void for_each_proof(u32 k, Fn &&fn) {
        const u32 nt = k >= 4 ? 4u : k; // (a thread start is ~30 us against ~150 us of work per proof)
        if (nt == 1) {
            for (u32 q = 0; q < k; ++q) fn(q);
            return;
        }
        std::atomic<bool> failed{false};
        std::atomic<u32> next{0}; // proofs are handed out one at a time: threads that never started cost nothing
        auto body = [&]() noexcept {
            try {
                for (u32 q; (q = next.fetch_add(1)) < k;) fn(q);
            } catch (...) {
                failed.store(true);
            }
        };
        {
            std::vector<std::thread> th;
            JoinAll guard{th};
            try {
                for (u32 t = 1; t < nt; ++t) th.emplace_back(body);
            } catch (...) { // std::system_error: fewer helpers
            }
            body();
        }
        if (failed.load()) throw std::runtime_error("prove: host assembly failed");
    }

    // ---- the host side of a pass, shared by the single-process paths (finish_pass) and the process-per-GPU one (assemble)
    struct Blind {
        u64 rc4[4], sc4[4], rs4[4];
        HostPoint t_rd, t_sd, t_rsd, t_sd2;
    };
    // r*delta_g1, s*delta_g1, (r s)*delta_g1, s*delta_g2: fixed-base (64 table additions each)
    void compute_blinds(u32 k, const uint64_t *r, const uint64_t *s, Blind *bl) const {
        for_each_proof(k, [&](u32 q) {
            Blind &b = bl[q];
            u64 rs_m[4];
            fr_->fr_to_canonical(r + 4 * q, b.rc4);
            fr_->fr_to_canonical(s + 4 * q, b.sc4);
            fr_->fr_mul(r + 4 * q, s + 4 * q, rs_m);
            fr_->fr_to_canonical(rs_m, b.rs4);
            g1_->hp_table_mul(delta1_tab_, b.rc4, &b.t_rd);
            g1_->hp_table_mul(delta1_tab_, b.sc4, &b.t_sd);
            g1_->hp_table_mul(delta1_tab_, b.rs4, &b.t_rsd);
            g2_->hp_table_mul(delta2_tab_, b.sc4, &b.t_sd2);
        });
    }
    // res[i * k + q] = MSM i (a, b_g1, b_g2, l, h) of proof q; writes A and C of every proof
    // `pre` (single proofs on a z3 slot): g_a and s g_a + r g1_b - rs delta were computed by assemble_g1_early while the h chain ran
    struct EarlyG1 {
        HostPoint g_a, g_c;
    };
    void assemble_g1_early(const HostPoint *res /* k = 1 */, Blind &b, const uint64_t *rq, EarlyG1 *e, uint8_t *out) const {
        const bool r_zero = (rq[0] | rq[1] | rq[2] | rq[3]) == 0;
        e->g_a = res[0];
        g1_->hp_add(&e->g_a, &a0_alpha_);
        g1_->hp_add(&e->g_a, &b.t_rd);
        HostPoint g1_b;
        g1_->hp_set_inf(&g1_b);
        if (!r_zero) {
            g1_b = res[1];
            g1_->hp_add(&g1_b, &b10_beta_);
            g1_->hp_add(&g1_b, &b.t_sd);
        }
        g1_->hp_mul2(&e->g_a, b.sc4, &g1_b, b.rc4, &e->g_c);
        g1_->hp_neg(&b.t_rsd);
        g1_->hp_add(&e->g_c, &b.t_rsd);
        g1_->hp_add(&e->g_c, &res[3]);
        g1_->hp_serialize(&e->g_a, out, true); // A is final (its inversion too runs beside the h chain)
    }
    void assemble_g1_late(const HostPoint *res /* k = 1 */, EarlyG1 *e, uint8_t *out) const {
        const int b1 = g1_->point_bytes(true), b2 = g2_->point_bytes(true);
        g1_->hp_add(&e->g_c, &res[4]);
        g1_->hp_serialize(&e->g_c, out + b1 + b2, true);
    }
    void assemble_g1(u32 k, const HostPoint *res, Blind *bl, const uint64_t *r, uint8_t *proofs_out) const {
        const int b1 = g1_->point_bytes(true), b2 = g2_->point_bytes(true);
        for_each_proof(k, [&](u32 q) {
            Blind &b = bl[q];
            const uint64_t *rq = r + 4 * q;
            const bool r_zero = (rq[0] | rq[1] | rq[2] | rq[3]) == 0; // g1_b is not used iff r == 0 (App. B.1)
            HostPoint g_a = res[0 * (size_t)k + q];
            g1_->hp_add(&g_a, &a0_alpha_);
            g1_->hp_add(&g_a, &b.t_rd);
            HostPoint g1_b;
            g1_->hp_set_inf(&g1_b);
            if (!r_zero) {
                g1_b = res[1 * (size_t)k + q];
                g1_->hp_add(&g1_b, &b10_beta_);
                g1_->hp_add(&g1_b, &b.t_sd);
            }
            HostPoint g_c;
            g1_->hp_mul2(&g_a, b.sc4, &g1_b, b.rc4, &g_c); // s*g_a + r*g1_b, one doubling chain
            g1_->hp_neg(&b.t_rsd);
            g1_->hp_add(&g_c, &b.t_rsd);
            g1_->hp_add(&g_c, &res[3 * (size_t)k + q]);
            g1_->hp_add(&g_c, &res[4 * (size_t)k + q]);
            uint8_t *out = proofs_out + (size_t)q * (2 * b1 + b2);
            g1_->hp_serialize(&g_a, out, true);
            g1_->hp_serialize(&g_c, out + b1 + b2, true);
        });
    }
    void assemble_g2(u32 k, const HostPoint *res, const Blind *bl, uint8_t *proofs_out) const {
        const int b1 = g1_->point_bytes(true), b2 = g2_->point_bytes(true);
        for_each_proof(k, [&](u32 q) {
            HostPoint g2_b = res[2 * (size_t)k + q];
            g2_->hp_add(&g2_b, &b20_beta_);
            g2_->hp_add(&g2_b, &bl[q].t_sd2);
            g2_->hp_serialize(&g2_b, proofs_out + (size_t)q * (2 * b1 + b2) + b1, true);
        });
    }

    // ---- process-per-GPU sharding (manta_rs_amd/distributed.py ShardedProver): this context is
    // shard g of G in its own process. partials_launch runs the pass on this shard's slices and leaves the five partial MSM
    // results of every proof on the DEVICE, folded there (msm_fold_device), as arkworks-format XYZZ points in slots of
    // slot_words() u32 -- [q][a, b_g1, b_g2, l, h] -- and makes `consumer` (the stream of the collective) wait for them: no
    // host synchronisation between launch and all_gather. assemble() adds the n_parts gathered copies and finishes the proofs.
    struct PartialJob {
        Pass p;
        std::shared_lock<std::shared_mutex> shape_lock; // held until partials_finish: set_r1cs waits for the pass
    };
    size_t slot_words() const override { return (size_t)g2_->xyzz_words(); }
    int partials_launch(u64 k64, const uint64_t *z, uint64_t *d_out, void *consumer, void **job_out) override {
        if (k64 == 0 || k64 > BATCH_CHUNK || !z || !d_out || !job_out || !peers_.empty()) return MG_ERR_ARG;
        DeviceGuard restore_callers_device;
        std::shared_lock<std::shared_mutex> shape_lock(shape_mu_);
        return partials_launch_locked(k64, z, d_out, consumer, job_out, &shape_lock);
    }
    // the caller holds shape_mu_ shared; with `take` the job keeps that lock until partials_finish (the public entry point),
    // without it the caller keeps holding it for the length of the pass (the in-library exchange of prove_pass)
    int partials_launch_locked(u64 k64, const uint64_t *z, uint64_t *d_out, void *consumer, void **job_out,
                               std::shared_lock<std::shared_mutex> *take, const uint64_t *const *z_list = nullptr) {
        if (!have_r1cs_) return MG_ERR_STATE;
        PartialJob *job = new PartialJob();
        if (take) job->shape_lock = std::move(*take);
        int rc = launch_pass(job->p, (u32)k64, z, nullptr, nullptr, nullptr, z_list, false);
        ProveWs *w = job->p.w;
        if (rc || !w) {
            if (w) abandon_pass(job->p);
            delete job;
            return rc ? rc : MG_ERR_HIP;
        }
        const size_t sw = slot_words();
        // the fold of MSM i goes behind it: on its own stream, or -- when the pass was replayed from the two graphs of the
        // "single" mode -- on the stream its graph was launched on (the G1 MSMs are joined inside that graph)
        const bool replayed = w->graphs_ready && w->g_all && w->g_g2;
        auto fold_stream = [&](int i) { return replayed ? (i == 2 ? msm_stream(w, 2) : w->stream) : msm_stream(w, i); };
        for (int i = 0; i < 5 && !rc; ++i) {
            if (does(i)) {
                rc = w->me[i]->msm_fold_device(w->mw[i], (u32 *)d_out + (size_t)i * sw, 5 * sw, fold_stream(i));
            } else { // another rank's MSM: this rank contributes the point at infinity (all-zero XYZZ)
                for (u64 q = 0; q < k64 && !rc; ++q)
                    if (hipMemsetAsync((u32 *)d_out + (q * 5 + (u64)i) * sw, 0, sw * 4, fold_stream(i)) != hipSuccess) rc = MG_ERR_HIP;
            }
        }
        hipStream_t cs = (hipStream_t)consumer;
        if (!rc) {
            hipError_t e = hipSuccess;
            for (int i = 0; i < 5 && e == hipSuccess; ++i) {
                hipStream_t ms = fold_stream(i);
                e = hipEventRecord(w->mw[i]->done, ms);
                if (e == hipSuccess && cs != ms) e = hipStreamWaitEvent(cs, w->mw[i]->done, 0);
            }
            if (e != hipSuccess) {
                set_last_hip_error(e, "partials_launch: events", __FILE__, __LINE__);
                rc = MG_ERR_HIP;
            }
        }
        if (rc) {
            abandon_pass(job->p);
            delete job;
            return rc;
        }
        *job_out = job;
        return MG_OK;
    }
    int partials_finish(void *job_in) override {
        PartialJob *job = static_cast<PartialJob *>(job_in);
        if (!job) return MG_ERR_ARG;
        DeviceGuard restore_callers_device;
        abandon_pass(job->p); // waits for the slot's streams and returns it (nothing is folded on the host)
        delete job;
        return MG_OK;
    }
    int assemble(u64 k64, u32 n_parts, const uint64_t *parts, const uint64_t *r, const uint64_t *s, uint8_t *proofs_out) override {
        if (k64 == 0 || k64 > 1024 || n_parts == 0 || !parts || !r || !s || !proofs_out) return MG_ERR_ARG;
        const u32 k = (u32)k64;
        const size_t sw = slot_words();
        std::vector<HostPoint> res((size_t)5 * k);
        for (int i = 0; i < 5; ++i) {
            GroupEngine *ge = i == 2 ? g2_ : g1_;
            for (u32 q = 0; q < k; ++q) {
                HostPoint &acc = res[(size_t)i * k + q], t;
                ge->hp_set_inf(&acc);
                for (u32 g = 0; g < n_parts; ++g) {
                    ge->hp_from_xyzz(&t, (const u32 *)parts + (((size_t)g * k + q) * 5 + (size_t)i) * sw);
                    ge->hp_add(&acc, &t);
                }
            }
        }
        std::vector<Blind> bl(k);
        compute_blinds(k, r, s, bl.data());
        assemble_g1(k, res.data(), bl.data(), r, proofs_out);
        assemble_g2(k, res.data(), bl.data(), proofs_out);
        return MG_OK;
    }

    // ---- the exchange step of an in-process sharded context over RCCL (mg_ctx_opts.exchange = MG_EXCHANGE_RCCL; BASELINE
    // north_star "final RCCL reduce of partial EC points over xGMI", reached from the Rust host through mg_ctx_create_ex):
    // one communicator per shard from ncclCommInitAll over the context's device list; a pass folds its five partial points
    // per proof on every device (the same partials_launch the process-per-GPU path uses) straight into that device's send
    // buffer, ONE grouped ncclAllGather moves them (k x 5 x 256 B per device for BN254), shard 0's copy lands in pinned
    // memory and the usual assembly adds the G copies. RCCL has no user-defined reduction, hence gather + sum. Several
    // passes may be in flight (the streamed batches run three): each takes a buffer set from a small pool; the enqueue of
    // the grouped collective is serialised -- communicators want one order of operations on every rank.
    struct ExSet {
        std::vector<u32 *> d_send, d_recv; // per shard, on its device
        std::vector<hipStream_t> st;       // the stream the collective runs on, per shard
        u32 *h_recv = nullptr;             // pinned: shard 0's gathered copy
    };
    struct Exchange {
        Rccl *api = nullptr;
        std::vector<ncclComm_t> comm;
        std::vector<ProverImpl *> shard;
        std::mutex mu, pool_mu;
        std::condition_variable pool_cv;
        std::vector<ExSet *> idle;
        int made = 0;
        size_t words = 0; // u32 per shard and set: BATCH_CHUNK proofs x 5 slots
    };
    Exchange *ex_ = nullptr;
    static constexpr int EX_SETS = 4;
    bool nccl_ok(ncclResult_t r, const char *what) {
        if (r == ncclSuccess) return true;
        char buf[256];
        std::snprintf(buf, sizeof(buf), "RCCL: %s failed: %s", what, ex_ && ex_->api ? ex_->api->GetErrorString(r) : "?");
        set_last_error_text(buf);
        return false;
    }
    int exchange_init() {
        Rccl *api = Rccl::get();
        if (!api) {
            set_last_error_text("MG_EXCHANGE_RCCL: librccl.so.1 could not be loaded (set MANTA_RCCL_LIB)");
            return MG_ERR_STATE;
        }
        Exchange *x = new Exchange();
        x->api = api;
        x->shard.push_back(this);
        x->shard.insert(x->shard.end(), peers_.begin(), peers_.end());
        const int G = (int)x->shard.size();
        std::vector<int> devs(G);
        for (int g = 0; g < G; ++g) devs[g] = x->shard[g]->dev_;
        for (int g = 0; g < G; ++g)
            for (int t = 0; t < g; ++t)
                if (devs[t] == devs[g]) { // RCCL refuses a device twice in one clique; the host exchange serves such lists
                    delete x;
                    set_last_error_text("MG_EXCHANGE_RCCL: a device is listed twice");
                    return MG_ERR_ARG;
                }
        x->comm.assign(G, nullptr);
        x->words = (size_t)BATCH_CHUNK * 5 * slot_words();
        ex_ = x;
        if (!nccl_ok(api->CommInitAll(x->comm.data(), G, devs.data()), "ncclCommInitAll")) {
            x->comm.clear();
            exchange_destroy();
            return MG_ERR_HIP;
        }
        return MG_OK;
    }
    void exchange_destroy() {
        if (!ex_) return;
        for (ExSet *e : ex_->idle) ex_free(e);
        for (ncclComm_t c : ex_->comm)
            if (c) ex_->api->CommDestroy(c);
        delete ex_;
        ex_ = nullptr;
    }
    void ex_free(ExSet *e) {
        for (size_t g = 0; g < e->st.size(); ++g) {
            hipSetDevice(ex_->shard[g]->dev_);
            if (e->d_send[g]) hipFree(e->d_send[g]);
            if (e->d_recv[g]) hipFree(e->d_recv[g]);
            if (e->st[g]) (void)hipStreamSynchronize(e->st[g]), stream_pool_put_normal(e->st[g]); // (pooled, never destroyed)
        }
        if (e->h_recv) hipHostFree(e->h_recv);
        delete e;
    }
    ExSet *ex_acquire() {
        std::unique_lock<std::mutex> lk(ex_->pool_mu);
        for (;;) {
            if (!ex_->idle.empty()) {
                ExSet *e = ex_->idle.back();
                ex_->idle.pop_back();
                return e;
            }
            if (ex_->made < EX_SETS) break;
            ex_->pool_cv.wait(lk);
        }
        ++ex_->made;
        lk.unlock();
        const size_t G = ex_->shard.size();
        ExSet *e = new ExSet();
        e->d_send.assign(G, nullptr), e->d_recv.assign(G, nullptr), e->st.assign(G, nullptr);
        bool ok = true;
        for (size_t g = 0; g < G && ok; ++g) {
            ok = hipSetDevice(ex_->shard[g]->dev_) == hipSuccess && hipMalloc((void **)&e->d_send[g], ex_->words * 4) == hipSuccess &&
                 hipMalloc((void **)&e->d_recv[g], G * ex_->words * 4) == hipSuccess &&
                 (e->st[g] = stream_pool_get_normal()) != nullptr;
        }
        ok = ok && hipHostMalloc((void **)&e->h_recv, G * ex_->words * 4, hipHostMallocDefault) == hipSuccess;
        if (!ok) {
            (void)hipGetLastError();
            ex_free(e);
            std::lock_guard<std::mutex> g2(ex_->pool_mu);
            --ex_->made;
            ex_->pool_cv.notify_one();
            return nullptr;
        }
        return e;
    }
    void ex_release(ExSet *e) {
        {
            std::lock_guard<std::mutex> g(ex_->pool_mu);
            ex_->idle.push_back(e);
        }
        ex_->pool_cv.notify_one();
    }
    // one pass of k proofs on every shard with the RCCL exchange; the caller (prove_pass) holds every shard's shape lock
    int prove_pass_rccl(u32 k, const uint64_t *z, const uint64_t *r, const uint64_t *s, uint8_t *proofs_out,
                        const uint64_t *const *z_list) {
        const size_t G = ex_->shard.size(), words = (size_t)k * 5 * slot_words();
        ExSet *e = ex_acquire();
        if (!e) return MG_ERR_OOM;
        std::vector<void *> jobs(G, nullptr);
        int rc = MG_OK;
        for (size_t g = G; g-- > 0 && !rc;) // shard 0 last, like the host exchange
            rc = ex_->shard[g]->partials_launch_locked(k, z, (uint64_t *)e->d_send[g], e->st[g], &jobs[g], nullptr, z_list);
        if (!rc) {
            std::lock_guard<std::mutex> one_order(ex_->mu);
            bool ok = nccl_ok(ex_->api->GroupStart(), "ncclGroupStart");
            for (size_t g = 0; g < G && ok; ++g)
                ok = nccl_ok(ex_->api->AllGather(e->d_send[g], e->d_recv[g], words / 2, ncclUint64, ex_->comm[g], e->st[g]), "ncclAllGather");
            ok = nccl_ok(ex_->api->GroupEnd(), "ncclGroupEnd") && ok;
            if (!ok) rc = MG_ERR_HIP;
        }
        if (!rc) {
            hipSetDevice(dev_);
            hipError_t he = hipMemcpyAsync(e->h_recv, e->d_recv[0], G * words * 4, hipMemcpyDeviceToHost, e->st[0]);
            if (he == hipSuccess) he = hipStreamSynchronize(e->st[0]);
            if (he != hipSuccess) {
                set_last_hip_error(he, "prove_pass_rccl: gathered points to the host", __FILE__, __LINE__);
                rc = MG_ERR_HIP;
            }
        } else {
            for (size_t g = 0; g < G; ++g) {
                hipSetDevice(ex_->shard[g]->dev_);
                hipStreamSynchronize(e->st[g]);
            }
        }
        for (size_t g = 0; g < G; ++g) // (every rank received the gather: wait for the others' streams before their buffers are reused)
            if (g && !rc) {
                hipSetDevice(ex_->shard[g]->dev_);
                hipStreamSynchronize(e->st[g]);
            }
        for (size_t g = 0; g < G; ++g)
            if (jobs[g]) ex_->shard[g]->partials_finish(jobs[g]);
        hipSetDevice(dev_);
        if (!rc) rc = assemble(k, (u32)G, (const uint64_t *)e->h_recv, r, s, proofs_out);
        ex_release(e);
        return rc;
    }

#ifdef MG_DIAG
    // diagnosis builds: word sums of every device buffer of the most recently used one-proof slot (it sits in ws_free_), so that a
    // pass replayed from graphs can be compared buffer by buffer with the same pass in a good state / launched eagerly
    int diag_slot_sums(u64 *out, int cap, int eager_next) {
        hipSetDevice(dev_);
        hipDeviceSynchronize();
        ProveWs *w = nullptr;
        {
            std::lock_guard<std::mutex> g(mu_);
            for (auto &kv : ws_free_)
                for (ProveWs *q : kv.second)
                    if (q->k == 1 && (!w || q->last_use > w->last_use)) w = q;
        }
        if (!w) return -1;
        if (eager_next == 1) w->drop_graphs(), w->no_graph = true;  // from now on this slot launches eagerly
        if (eager_next == 2) w->drop_graphs(), w->no_graph = false, w->eager_runs = 2; // re-capture on the next pass
        int n = 0;
        auto sum = [&](const void *p, size_t bytes) {
            u64 acc = 0;
            if (p && bytes) {
                std::vector<u32> h(bytes / 4);
                hipMemcpy(h.data(), p, bytes / 4 * 4, hipMemcpyDeviceToHost);
                for (size_t i = 0; i < h.size(); ++i) acc = acc * 1000003ull + h[i];
            }
            if (n < cap) out[n] = acc;
            ++n;
        };
        const size_t D = (size_t)1 << log_d_, ww = (size_t)fr_->work_words() * 4;
        sum(w->z.p, V_ * 32);                       // 0 z
        sum(w->a.p, D * ww);                        // 1 h (a)
        sum((char *)w->a.p + D * ww, D * ww);       // 2 b
        sum((char *)w->a.p + 2 * D * ww, D * ww);   // 3 c
        {                                           // 4: non-zero words of a | b | c, 5: index of the first one
            std::vector<u32> h(3 * D * ww / 4);
            hipMemcpy(h.data(), w->a.p, h.size() * 4, hipMemcpyDeviceToHost);
            u64 nz = 0, first = ~0ull;
            for (size_t i = 0; i < h.size(); ++i)
                if (h[i]) {
                    if (!nz) first = i;
                    ++nz;
                }
            if (n < cap) out[n] = nz;
            ++n;
            if (n < cap) out[n] = first;
            ++n;
        }
        for (int i : {0, 2, 4}) {                   // 6.. : 10 per MSM (a / z3, b_g2, h)
            MsmWorkspace *m = w->mw[i];
            sum(m->count.p, m->count.p ? 4 : 0);
            sum(m->keys_in.p, m->keys_in.cap);
            sum(m->vals_in.p, m->vals_in.cap);
            sum(m->keys_out.p, m->keys_out.cap);
            sum(m->vals_out.p, m->vals_out.cap);
            sum(m->pkeys[0].p, m->pkeys[0].cap);
            sum(m->ppts[0].p, m->ppts[0].cap);
            sum(m->ppts[1].p, m->ppts[1].cap);
            sum(m->buckets.p, m->buckets.cap);
            sum(m->redS.p, m->redS.cap);
        }
        return n;
    }
#endif

    int finish_pass(Pass &p, int rc, std::vector<Pass> *peer_passes = nullptr) {
        ProveWs *w = p.w;
        if (!w || rc) {
            if (peer_passes)
                for (size_t g = 0; g < peers_.size(); ++g) peers_[g]->abandon_pass((*peer_passes)[g]);
            if (w) abandon_pass(p);
            return rc ? rc : MG_ERR_HIP;
        }
        try {
            return finish_pass_body(p, peer_passes);
        } catch (...) { // bad_alloc in the host assembly: drain and return every slot of the pass, then report upwards
            abandon_pass(p);
            if (peer_passes)
                for (size_t g = 0; g < peers_.size(); ++g) peers_[g]->abandon_pass((*peer_passes)[g]);
            throw;
        }
    }
    bool peer_passes_active(const std::vector<Pass> *peer_passes) const { return peer_passes && !peers_.empty(); }
    int finish_pass_body(Pass &p, std::vector<Pass> *peer_passes) {
        ProveWs *w = p.w;
        int rc = MG_OK;
        const u32 k = p.k;
        const uint64_t *r = p.r, *s = p.s;
        // ---- host work that does not depend on the MSMs runs while the GPU is busy: the blinding terms
        std::vector<Blind> bl(k);
        if (!rc) compute_blinds(k, r, s, bl.data());
        std::vector<HostPoint> res((size_t)5 * k), tmp; // res[i * k + q]: MSM i of proof q
        auto collect = [&](hipStream_t, bool part_a) { // wait for one part on every shard and fold its MSMs
            int rc2 = collect_part(p, part_a, res.data());
            if (!rc) rc = rc2;
            if (peer_passes && !peers_.empty()) { // the exchange step of the sharded path: partial points are summed here
                tmp.resize((size_t)5 * k);
                for (size_t g = 0; g < peers_.size(); ++g) {
                    rc2 = peers_[g]->collect_part((*peer_passes)[g], part_a, tmp.data());
                    if (!rc) rc = rc2;
                    for (int i = 0; i < 5 && !rc2; ++i) {
                        if (in_part_a(i) != part_a) continue;
                        GroupEngine *ge = i == 2 ? g2_ : g1_;
                        for (u32 q = 0; q < k; ++q) ge->hp_add(&res[(size_t)i * k + q], &tmp[(size_t)i * k + q]);
                    }
                }
                hipSetDevice(dev_);
            }
        };
        // ---- part A is back: the G1 side of the assembly runs while the G2 MSM finishes
        const auto t_wait0 = std::chrono::steady_clock::now();
        // A single proof on a z3 slot: the combined a | b_g1 | l MSM ends before the h chain does (witness map, then the one dense MSM
        // of a proof). Its end-of-chain token lands in pinned memory; the host folds the three results and runs s A + r B1 (0.1 ms, the
        // long piece of host work) while the GPU finishes h. The order of additions into C differs from assemble_g1's; the point is
        // the same and so are its bytes.
        EarlyG1 early;
        bool have_early = false, g2_early = false;
        if (!rc && k == 1 && w->z3 && w->mw[0]->notify && w->mw[0]->h_flag && w->mw[0]->pending && !peer_passes_active(peer_passes)) {
            volatile u32 *flag = w->mw[0]->h_flag;
            bool seen = false;
            // a short spin (the chain usually ends within tens of microseconds of the host getting here on dense witnesses), then the
            // core is offered to other runnable threads between polls: six signer threads on one context must not pin six cores for
            // the length of their GPU chains (advisor r4); a lone caller's sched_yield returns at once
            for (u32 spin = 0;; ++spin) {
                if (*flag) {
                    seen = true;
                    break;
                }
                // (a failed launch or a device fault never writes the token: every few microseconds ask the stream itself)
                if ((spin & 1023u) == 1023u && hipStreamQuery(w->linear3 ? msm_stream(w, 0) : w->stream) != hipErrorNotReady) break;
                cpu_relax();
                if (spin >= 2048u && (spin & 15u) == 15u) std::this_thread::yield();
            }
            if (!seen && *flag) seen = true; // (the stream reported complete: the token was written before that)
            std::atomic_thread_fence(std::memory_order_seq_cst);
            if (seen) {
                HostPoint t3[3];
                const int rc2 = w->me[0]->msm_finish(w->mw[0], t3, true);
                if (!rc2) {
                    res[0] = t3[0], res[1] = t3[1], res[3] = t3[2];
                    p.z3_folded = true;
                    assemble_g1_early(res.data(), bl[0], r, &early, p.out);
                    have_early = true;
                    // the G2 chain (a linear graph on a stream of its own) has usually ended by now: its element too is
                    // finished before the h chain is waited for
                    if (hipStreamQuery(msm_stream(w, 2)) == hipSuccess) {
                        collect(msm_stream(w, 2), false);
                        if (!rc) assemble_g2(k, res.data(), bl.data(), p.out);
                        g2_early = true;
                    } else {
                        (void)hipGetLastError(); // (hipErrorNotReady is not an error here)
                    }
                } else {
                    rc = rc2;
                }
            }
        }
        collect(w->stream, true); // every G1 MSM stream has been joined into it
        if (!rc && have_early) assemble_g1_late(res.data(), &early, p.out);
        else if (!rc) assemble_g1(k, res.data(), bl.data(), r, p.out);
        // ---- part B: the G2 element
        if (!g2_early) collect(msm_stream(w, 2), false);
        float phases[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        const bool timed = w->timed && !rc;
        if (timed) { // every event has completed: both parts were synchronised above
            hipEventElapsedTime(&phases[0], w->tev[0], w->tev[1]);
            hipEventElapsedTime(&phases[1], w->tev[1], w->tev[2]);
            for (int i = 0; i < 5; ++i)
                if (runs(w, i)) hipEventElapsedTime(&phases[2 + i], w->tev[3 + 2 * i], w->tev[4 + 2 * i]); // (z3: "msm_a" is a + b_g1 + l)
            hipEventElapsedTime(&phases[7], w->tev[0], w->tev[13]);
            hipEventElapsedTime(&phases[8], w->tev[0], w->tev[14]);
        }
        const auto t_host = std::chrono::steady_clock::now();
        const float wait_ms = std::chrono::duration<float, std::milli>(t_host - t_wait0).count();
        ws_release(w);
        p.w = nullptr;
        if (peer_passes)
            for (size_t g = 0; g < peers_.size(); ++g) {
                Pass &pg = (*peer_passes)[g];
                if (pg.w) peers_[g]->ws_release(pg.w);
                pg.w = nullptr;
            }
        if (rc) return rc;
        if (!g2_early) assemble_g2(k, res.data(), bl.data(), p.out);
        {
            const float hv[3] = {p.enqueue_ms, wait_ms, std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_host).count()};
            set_last_pass_host_ms(hv);
        }
        if (timed) {
            phases[9] = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_host).count();
            set_last_prove_ms(phases);
        }
        return MG_OK;
    }
};

} // namespace

// Every way a context comes into being goes through here (mg_ctx_create_ex; the older entry points fill a ProverOptions):
//   devices / n_devices  one process driving several GPUs: shard g owns the g-th contiguous slice of every query on devices[g]
//                        (a device may be listed more than once -- two shards then share it, which is how the path is tested on
//                        a 1-GPU box). Shard 0 is the object handed back; it owns the others.
//   shard / n_shards     one shard of a context in ITS OWN process (one process per GPU): slice `shard` of `n_shards` of every
//                        query on the current device, no peers -- the partial results meet through partials_launch / assemble
//                        and a collective (distributed.py).
//   task_mask            task placement (last row): the whole key, the MSMs of the mask computed in full.
//   full_table_bytes     HBM budget of the context's full tables (see resolve_full_budget)
//   exchange             how the partial points of an in-process sharded context meet: host-staged sum, or RCCL all_gather
int prover_create_ex(int curve, const mg_pk_view *pk, const ProverOptions &o, Prover **out) {
    if (!pk || !out) return MG_ERR_ARG;
    HeavyOp no_capture_meanwhile;
    if (o.task_mask > 0x1f || o.n_shards == 0 || o.shard >= o.n_shards || o.n_shards > 64) return MG_ERR_ARG;
    if (o.exchange != 0 && o.exchange != 1) return MG_ERR_ARG;
    const bool listed = o.devices && o.n_devices > 0;
    if (listed && (o.n_shards > 1 || o.task_mask != 0x1f)) return MG_ERR_ARG; // one placement at a time
    if (o.n_shards > 1 && o.task_mask != 0x1f) return MG_ERR_ARG;
    if (o.n_devices < 0 || o.n_devices > 64) return MG_ERR_ARG;
    int count = 0, prev = 0;
    MG_HIP(hipGetDeviceCount(&count));
    MG_HIP(hipGetDevice(&prev));
    if (!listed) {
        if (o.exchange != 0) return MG_ERR_ARG; // a collective needs a device list
        ProverImpl *p = new ProverImpl();
        if (o.tuning) p->tn_ = *o.tuning;
        p->task_mask_ = o.task_mask;
        const int rc = p->init(curve, pk, prev, o.shard, o.n_shards, o.full_table_bytes, 1, !o.partials_interface);
        if (rc) {
            delete p;
            return rc;
        }
        *out = p;
        return MG_OK;
    }
    for (int g = 0; g < o.n_devices; ++g)
        if (o.devices[g] < 0 || o.devices[g] >= count) return MG_ERR_ARG;
    ProverImpl *p0 = new ProverImpl();
    int rc = MG_OK;
    for (int g = o.n_devices - 1; g >= 0 && !rc; --g) { // shard 0 last: it ends up the current device's context
        ProverImpl *p = g == 0 ? p0 : new ProverImpl();
        if (o.tuning) p->tn_ = *o.tuning;
        if (g) p0->peers_.insert(p0->peers_.begin(), p), p->shard_owner_ = p0;
        int same = 0;
        for (int t = 0; t < o.n_devices; ++t) same += o.devices[t] == o.devices[g];
        rc = p->init(curve, pk, o.devices[g], (u32)g, (u32)o.n_devices, o.full_table_bytes, same, o.exchange == 0);
    }
    if (!rc && o.exchange == 1) rc = p0->exchange_init();
    hipSetDevice(prev);
    if (rc) {
        delete p0;
        return rc;
    }
    *out = p0;
    return MG_OK;
}

#ifdef MG_DIAG
extern "C" __attribute__((visibility("default"))) int mg_diag_slot_sums(void *ctx, uint64_t *out, int cap, int eager_next) {
    Prover *p = *(Prover **)ctx; // struct mg_ctx { Prover *p; }
    return static_cast<ProverImpl *>(p)->diag_slot_sums(out, cap, eager_next);
}
#endif

int prover_create(int curve, const mg_pk_view *pk, Prover **out) { return prover_create_ex(curve, pk, ProverOptions(), out); }
int prover_create_shard(int curve, const mg_pk_view *pk, u32 shard, u32 n_shards, Prover **out) {
    ProverOptions o;
    o.shard = shard, o.n_shards = n_shards;
    o.partials_interface = true; // one process per GPU: partials_launch / assemble (a world of one included)
    return prover_create_ex(curve, pk, o, out);
}
int prover_create_task(int curve, const mg_pk_view *pk, u32 task_mask, Prover **out) {
    ProverOptions o;
    o.task_mask = task_mask;
    return prover_create_ex(curve, pk, o, out);
}
int prover_create_sharded(int curve, const mg_pk_view *pk, const int *devices, int n_devices, Prover **out) {
    if (!devices || n_devices < 1) return MG_ERR_ARG;
    ProverOptions o;
    o.devices = devices, o.n_devices = n_devices;
    return prover_create_ex(curve, pk, o, out);
}

} // namespace mg
