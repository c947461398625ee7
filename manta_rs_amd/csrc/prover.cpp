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

// The prover is one class hierarchy split along its seams (round 6): prover_slot.h (slot), prover_key.h (ProverKey: key + circuit),
// prover_assembly.h (ProverAssembly: host assembly), prover_passes.h (ProverSlots: slots, enqueue, graph capture); below: ProverImpl --
// passes (launch / collect / finish), the coalescing queue, batches, the partials interface, the in-library RCCL exchange.
} // namespace mg
#include "prover_slot.h"
#include "prover_key.h"
#include "prover_assembly.h"
#include "prover_passes.h"
namespace mg {
namespace {

class ProverImpl : public ProverSlots {
  public:
    ~ProverImpl() override {
        HeavyOp no_capture_meanwhile;
        exchange_destroy();
        for (ProverKey *q : peers_) delete q;
        peers_.clear();
    }
    ProverImpl *peer(size_t g) const { return static_cast<ProverImpl *>(peers_[g]); }
    // ---- one proof. Concurrent callers on one context are COALESCED: while COALESCE_INFLIGHT passes are on the GPU, further
    // calls queue up, and the next caller to find a pass slot free takes everything queued (up to BATCH_CHUNK) as ONE batched
    // pass -- the wallet / ledger simulation of the reference drives one ProvingContext from six threads
    // (manta-pay/src/bin/simulation.rs:36-38, simulation/mod.rs:75-79), each proving one transfer at a time; without coalescing
    // their passes only share the GPU's queues (a quarter of the rate of explicit batches). A lone caller
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
    // witness-map head and the bucket-reduce tail of one pass then overlap the accumulate kernels of the others (three in flight
    // is the measured optimum; passes of 16 or 64 are no better).
    static constexpr u64 BATCH_CHUNK = 32;
    int prove_batch(u64 k64, const uint64_t *z, const uint64_t *r, const uint64_t *s, uint8_t *proofs_out) override {
        if (k64 == 0 || k64 > 1024 || !z || !r || !s || !proofs_out) return MG_ERR_ARG;
        if (task_mask_ != 0x1f || lone_range_shard()) return MG_ERR_STATE;
        if (k64 <= BATCH_CHUNK) return prove_pass(k64, z, r, s, proofs_out);
        // (passes of exactly BATCH_CHUNK proofs plus one remainder: every distinct pass size needs its own workspaces and graphs)
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
    // two callers streaming a batch each would otherwise put six passes in flight, which is slower than three
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
        for (ProverKey *q : peers_) locks.emplace_back(q->shape_mu_);
        if (!have_r1cs_) return MG_ERR_STATE;
        if (ex_) return prove_pass_rccl((u32)k64, z, r, s, proofs_out, z_list); // the partial points meet through RCCL
        // every shard gets the whole assignment (1.1 MB for PrivateTransfer) and recomputes the witness map -- cheaper
        // than broadcasting h -- then multiplies its slices; shard 0 launches last and assembles
        std::vector<Pass> pp(peers_.size());
        int rc = MG_OK;
        for (size_t g = 0; g < peers_.size() && !rc; ++g) rc = peer(g)->launch_pass(pp[g], (u32)k64, z, r, s, nullptr);
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
        for (size_t g = 0; g < peers_.size(); ++g) x->shard.push_back(peer(g));
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
        has_exchange_ = true;
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
        has_exchange_ = false;
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
                for (size_t g = 0; g < peers_.size(); ++g) peer(g)->abandon_pass((*peer_passes)[g]);
            if (w) abandon_pass(p);
            return rc ? rc : MG_ERR_HIP;
        }
        try {
            return finish_pass_body(p, peer_passes);
        } catch (...) { // bad_alloc in the host assembly: drain and return every slot of the pass, then report upwards
            abandon_pass(p);
            if (peer_passes)
                for (size_t g = 0; g < peers_.size(); ++g) peer(g)->abandon_pass((*peer_passes)[g]);
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
                    rc2 = peer(g)->collect_part((*peer_passes)[g], part_a, tmp.data());
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
                if (pg.w) peer(g)->ws_release(pg.w);
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
