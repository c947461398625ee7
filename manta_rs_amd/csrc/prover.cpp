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
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

namespace mg {

FrEngine *get_ntt_engine(int curve) {
    static std::mutex mu;
    static FrEngine *tab[2] = {nullptr, nullptr};
    if (curve < 0 || curve > 1) return nullptr;
    std::lock_guard<std::mutex> g(mu);
    if (!tab[curve]) tab[curve] = curve == 0 ? make_fr_engine_bn254() : make_fr_engine_bls381();
    return tab[curve];
}

namespace {

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
    int eager_runs = 0;
    bool no_graph = false;
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
        drop_graphs();
        for (int i = 0; i < 5; ++i)
            if (mw[i]) {
                mw[i]->run_on = nullptr;
                me[i]->ws_release(mw[i]);
            }
        z.release();
        a.release();
        if (h_z) hipHostFree(h_z);
        if (z_ready) hipEventDestroy(z_ready);
        if (h_ready) hipEventDestroy(h_ready);
        if (fork) hipEventDestroy(fork);
        stream_pool_put(stream); // never destroyed: see stream_pool_get()
        stream_pool_put(side[0]);
        stream_pool_put(side[1]);
    }
};

static int prove_streams() {
    static const int n = [] {
        const char *e = std::getenv("MANTA_PROVE_STREAMS");
        const int v = e ? std::atoi(e) : 6;
        return v == 1 || v == 3 ? v : 6;
    }();
    return n;
}

// MANTA_GRAPH = single (default): one captured graph for the whole proof (fork/join over all streams);
//               split: six single-stream graphs (witness map + one per MSM) with eager event fork/join;
//               off (or MANTA_NO_GRAPH): plain stream launches
enum GraphMode { GRAPH_OFF = 0, GRAPH_SINGLE = 1, GRAPH_SPLIT = 2 };
static GraphMode graph_mode() {
    static const GraphMode m = [] {
        if (std::getenv("MANTA_NO_GRAPH")) return GRAPH_OFF;
        const char *e = std::getenv("MANTA_GRAPH");
        if (!e) return GRAPH_SINGLE;
        if (!std::strcmp(e, "off")) return GRAPH_OFF;
        return std::strcmp(e, "split") ? GRAPH_SINGLE : GRAPH_SPLIT;
    }();
    return m;
}
static bool graphs_enabled() { return graph_mode() != GRAPH_OFF; }

class ProverImpl : public Prover {
  public:
    int curve_ = 0;
    FrEngine *fr_ = nullptr;
    GroupEngine *g1_ = nullptr, *g2_ = nullptr;
    u64 V_ = 0, P_ = 0, h_len_ = 0, m_ = 0;
    unsigned log_d_ = 0;
    bool have_r1cs_ = false;
    BaseSet *a_bs_ = nullptr, *b1_bs_ = nullptr, *b2_bs_ = nullptr, *h_bs_ = nullptr, *l_bs_ = nullptr;
    BaseSet *h_bs_wide_ = nullptr; // the h query again with wider windows, for batched passes (nullptr: same as h_bs_)
    // the z queries again with 10-bit windows for batched passes (fewer mixed additions; single proofs want the
    // short bucket reduce of narrow windows, above all on the G2 chain); nullptr: same as the narrow set
    BaseSet *a_bs_wide_ = nullptr, *b1_bs_wide_ = nullptr, *b2_bs_wide_ = nullptr, *l_bs_wide_ = nullptr;
    HostPoint alpha_g1_, beta_g1_, delta_g1_, beta_g2_, delta_g2_, a0_, b1_0_, b2_0_;
    HostPoint a0_alpha_, b10_beta_, b20_beta_; // constant terms of g_a, g1_b, g2_b folded once
    void *delta1_tab_ = nullptr, *delta2_tab_ = nullptr; // fixed-base tables for r*delta, s*delta, rs*delta
    DevCsr A_, B_, C_;
    std::vector<u32> h_query_host_; // kept until the domain size is known (set_r1cs), then re-laid
    std::mutex mu_;
    std::map<u32, std::vector<ProveWs *>> ws_free_; // idle proof slots, by batch size

    ~ProverImpl() override {
        // (h_bs_ is created by set_r1cs)
        if (a_bs_) g1_->bases_destroy(a_bs_);
        if (b1_bs_) g1_->bases_destroy(b1_bs_);
        if (h_bs_) g1_->bases_destroy(h_bs_);
        if (h_bs_wide_) g1_->bases_destroy(h_bs_wide_);
        if (l_bs_) g1_->bases_destroy(l_bs_);
        if (b2_bs_) g2_->bases_destroy(b2_bs_);
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

    // window bits for precomputed tables, by MSM length (HBM is plentiful: trade table size for fewer
    // buckets to fold and no doubling chain -- tuned on MI355X)
    static int pre_c_for(u64 n) {
        if (const char *e = std::getenv("MANTA_PROVE_C")) // tuning override (window bits of the pk tables)
            if (std::atoi(e) > 0) return std::atoi(e);
        // Measured on MI355X for the PrivateTransfer shape (n = 35k / 65k): c = 6..8 -> 2.0 ms per proof,
        // c = 9..13 -> 2.5-2.7 ms, c = 14 -> 3.0 ms. Few buckets keep the latency-bound bucket reduce short
        // (B = 128: two tiles); the extra windows only add perfectly parallel mixed additions.
        if (n <= (1u << 17)) return 8;
        if (n <= (1u << 19)) return 12;
        return 16;
    }

    int init(int curve, const mg_pk_view *pk) {
        curve_ = curve;
        fr_ = get_ntt_engine(curve);
        g1_ = get_engine(curve, 1);
        g2_ = get_engine(curve, 2);
        if (!fr_ || !g1_ || !g2_) return MG_ERR_ARG;
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
        const int c_z = pre_c_for(V_ - 1);
        if ((rc = g1_->bases_create((const u32 *)pk->a_query + w1, V_ - 1, false, c_z, &a_bs_, true))) return rc;
        if ((rc = g1_->bases_create((const u32 *)pk->b_g1_query + w1, V_ - 1, false, c_z, &b1_bs_, true))) return rc;
        // The G2 MSM is the latency-critical chain of a single proof: 6-bit windows (32 buckets: one tile, no second
        // reduce level) shorten it by four dependent additions (measured +4 % proofs/s); the extra windows only
        // add parallel mixed additions.
        const bool small = V_ - 1 <= (1u << 17) && !std::getenv("MANTA_PROVE_C");
        const int c_g2 = small ? 6 : c_z;
        if ((rc = g2_->bases_create((const u32 *)pk->b_g2_query + w2, V_ - 1, false, c_g2, &b2_bs_, true))) return rc;
        if ((rc = g1_->bases_create((const u32 *)pk->l_query, V_ - P_, false, pre_c_for(V_ - P_), &l_bs_, true))) return rc;
        if (small) { // batched passes are throughput-bound: 10-bit windows = 20 % fewer mixed additions (+7 % measured)
            const int cw = 10;
            if ((rc = g1_->bases_create((const u32 *)pk->a_query + w1, V_ - 1, false, cw, &a_bs_wide_, true))) return rc;
            if ((rc = g1_->bases_create((const u32 *)pk->b_g1_query + w1, V_ - 1, false, cw, &b1_bs_wide_, true))) return rc;
            if ((rc = g2_->bases_create((const u32 *)pk->b_g2_query + w2, V_ - 1, false, cw, &b2_bs_wide_, true))) return rc;
            if ((rc = g1_->bases_create((const u32 *)pk->l_query, V_ - P_, false, cw, &l_bs_wide_, true))) return rc;
        }
        // h_query is stored in the bit-reversed order the witness map leaves h in; that order depends on
        // the domain size, known once the R1CS arrives (set_r1cs)
        h_query_host_.assign((const u32 *)pk->h_query, (const u32 *)pk->h_query + (size_t)h_len_ * w1);
        return MG_OK;
    }

    static int upload_csr(const mg_csr *src, u64 m, u64 n_vars, DevCsr &dst) {
        if (!src->row_ptr || (src->nnz && (!src->col || !src->val))) return MG_ERR_ARG;
        if (src->row_ptr[m] != src->nnz) return MG_ERR_ARG;
        for (u64 k = 0; k < src->nnz; ++k)
            if (src->col[k] >= n_vars) return MG_ERR_ARG;
        free_csr(dst);
        dst.nnz = src->nnz;
        MG_HIP(hipMalloc((void **)&dst.row_ptr, (m + 1) * 4));
        MG_HIP(hipMalloc((void **)&dst.col, (src->nnz ? src->nnz : 1) * 4));
        MG_HIP(hipMalloc((void **)&dst.val, (src->nnz ? src->nnz : 1) * 32));
        MG_HIP(hipMemcpy(dst.row_ptr, src->row_ptr, (m + 1) * 4, hipMemcpyHostToDevice));
        if (src->nnz) {
            MG_HIP(hipMemcpy(dst.col, src->col, src->nnz * 4, hipMemcpyHostToDevice));
            MG_HIP(hipMemcpy(dst.val, src->val, src->nnz * 32, hipMemcpyHostToDevice));
        }
        return MG_OK;
    }

    int set_r1cs(const mg_csr *a, const mg_csr *b, const mg_csr *c, u64 m) override {
        std::lock_guard<std::mutex> g(mu_);
        if (m == 0 || m + P_ > ((u64)1 << 32)) return MG_ERR_ARG;
        unsigned lg = 0;
        while (((u64)1 << lg) < m + P_) ++lg; // GeneralEvaluationDomain::new(m + P) -> next power of two
        if ((int)lg > fr_->two_adicity()) return MG_ERR_DOMAIN;
        int rc;
        if ((rc = upload_csr(a, m, V_, A_)) || (rc = upload_csr(b, m, V_, B_)) || (rc = upload_csr(c, m, V_, C_)))
            return rc;
        if (!h_bs_ || lg != log_d_) { // (re)build the h-query base set for this domain
            const size_t D = (size_t)1 << lg, w1 = (size_t)g1_->affine_words();
            std::vector<u32> perm(D * w1, 0u); // entries beyond len(h_query) stay infinity: h[D-1] = 0 anyway
            for (size_t p = 0; p < D; ++p) {
                size_t src = 0;
                for (unsigned b = 0; b < lg; ++b) src |= ((p >> b) & 1) << (lg - 1 - b);
                if (src < h_len_) std::memcpy(&perm[p * w1], &h_query_host_[src * w1], w1 * 4);
            }
            if (h_bs_) g1_->bases_destroy(h_bs_);
            if (h_bs_wide_) g1_->bases_destroy(h_bs_wide_);
            h_bs_ = h_bs_wide_ = nullptr;
            // The h MSM is the one with dense, uniform scalars -- half of all the mixed additions of a proof at
            // c = 8. Wider windows halve them, but lengthen its bucket reduce: measured on PrivateTransfer,
            // c_h = 8/10/12/14/16 -> 2033 / 2202 / 2219 / 2363 / 2287 proofs/s batched (k = 32); for single proofs the
            // reduce chain matters more (with the cooperative reduce: c_h = 8/10/12 -> 839 / 859 / 862 proofs/s).
            // The tables are small (80 MB), so single proofs and batches each get their own width.
            int ch = pre_c_for(D), ch_wide = ch;
            if (lg >= 16 && lg <= 17) ch = 12; // dense 2^16 scalars: a third fewer mixed additions, 32 reduce tiles (+3 %)
            if (lg <= 17) ch_wide = (int)lg - 2 < 8 ? 8 : ((int)lg - 2 > 14 ? 14 : (int)lg - 2);
            if (const char *e = std::getenv("MANTA_PROVE_CH")) ch = ch_wide = std::atoi(e) > 0 ? std::atoi(e) : ch;
            if ((rc = g1_->bases_create(perm.data(), D, false, ch, &h_bs_))) return rc;
            if (ch_wide != ch && (rc = g1_->bases_create(perm.data(), D, false, ch_wide, &h_bs_wide_))) return rc;
        }
        // pooled proof slots hold captured graphs and buffers sized for the previous shape: drop them
        for (auto &kv : ws_free_)
            for (ProveWs *w : kv.second) delete w;
        ws_free_.clear();
        m_ = m;
        log_d_ = lg;
        have_r1cs_ = true;
        return MG_OK;
    }

    ProveWs *ws_acquire(u32 k = 1) {
        {
            std::lock_guard<std::mutex> g(mu_);
            auto it = ws_free_.find(k);
            if (it != ws_free_.end() && !it->second.empty()) {
                ProveWs *w = it->second.back();
                it->second.pop_back();
                return w;
            }
        }
        ProveWs *w = new ProveWs();
        w->k = k;
        if (!(w->stream = stream_pool_get()) || !(w->side[0] = stream_pool_get()) ||
            !(w->side[1] = stream_pool_get()) || hipEventCreateWithFlags(&w->z_ready, hipEventDisableTiming) != hipSuccess ||
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
        }
        // Three streams per proof, not six: the G2 MSM is the critical path (~3x a G1 MSM), so the three
        // z-MSMs over G1 run back to back beside it and the h MSM follows the witness map on the main stream.
        // Fewer streams = fewer hardware queues per proof in flight (the runtime multiplexes streams onto
        // GPU_MAX_HW_QUEUES queues; streams that share one serialise). MANTA_PROVE_STREAMS=6 restores one
        // stream per MSM.
        w->mw[2]->run_on = w->side[0]; // the G2 MSM (the critical path) gets a high-priority stream of its own
        if (prove_streams() == 3) {
            w->mw[0]->run_on = w->side[1];
            w->mw[1]->run_on = w->side[1];
            w->mw[2]->run_on = w->side[0];
            w->mw[3]->run_on = w->side[1];
            w->mw[4]->run_on = w->stream;
        } else if (prove_streams() == 1) {
            for (int i = 0; i < 5; ++i) w->mw[i]->run_on = w->stream;
        }
        return w;
    }
    void ws_release(ProveWs *w) {
        std::lock_guard<std::mutex> g(mu_);
        ws_free_[w->k].push_back(w);
    }

    // Witness map for the slot's w->k assignments (stored back to back, like the three work vectors: member q of
    // a batch lives V resp. D elements after member q-1); h ends up in w->a.
    int reserve_witness_map(ProveWs *w) {
        const size_t D = (size_t)1 << log_d_, k = w->k;
        int rc;
        if ((rc = w->z.reserve(k * V_ * 32)) || (rc = w->a.reserve(3 * k * D * 32))) return rc;
        return MG_OK;
    }
    // everything after the upload of z, on w->stream (this is what the witness-map graph captures)
    int enqueue_witness_map_body(ProveWs *w) {
        const size_t D = (size_t)1 << log_d_, k = w->k;
        int rc;
        hipStream_t s = w->stream;
        MG_HIP(hipMemsetAsync(w->a.p, 0, 3 * k * D * 32, s));
        u32 *a = w->a.as<u32>(), *b = a + k * D * 8, *c = b + k * D * 8, *zz = w->z.as<u32>();
        const size_t zs = (size_t)V_ * 8, ds = D * 8;
        if ((rc = fr_->spmv3(A_, B_, C_, zz, a, b, c, m_, s, (u32)k, zs, ds))) return rc;
        // input-consistency rows: a[m + j] = z_j for j < P (mpc.rs:299-312)
        MG_HIP(hipMemcpy2DAsync(a + (size_t)m_ * 8, D * 32, zz, V_ * 32, P_ * 32, k, hipMemcpyDeviceToDevice, s));
        // ifft x3, coset fft x3, (ab - c)/Z, coset ifft -- fused; leaves h bit-reversed in `a`
        if ((rc = fr_->qap_quotient(a, b, c, log_d_, s, (u32)k))) return rc;
        return MG_OK;
    }
    // H2D(z), recording z_ready
    int upload_z(ProveWs *w, const uint64_t *z) {
        int rc = reserve_witness_map(w);
        if (rc) return rc;
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
        if (!have_r1cs_) return MG_ERR_STATE;
        ProveWs *w = ws_acquire();
        if (!w) return MG_ERR_HIP;
        int rc = upload_z(w, z);
        if (!rc) rc = launch_witness_map(w);
        if (!rc) {
            const size_t D = (size_t)1 << log_d_;
            std::vector<uint64_t> tmp(D * 4);
            hipError_t e = hipMemcpyAsync(tmp.data(), w->a.p, D * 32, hipMemcpyDeviceToHost, w->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(w->stream);
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
        const bool wide = w->k >= 4;
        return MsmArgs{{wide && a_bs_wide_ ? a_bs_wide_ : a_bs_, wide && b1_bs_wide_ ? b1_bs_wide_ : b1_bs_,
                        wide && b2_bs_wide_ ? b2_bs_wide_ : b2_bs_, wide && l_bs_wide_ ? l_bs_wide_ : l_bs_,
                        wide && h_bs_wide_ ? h_bs_wide_ : h_bs_},
                       {dz + 8, dz + 8, dz + 8, dz + (size_t)P_ * 8, w->a.as<u32>()},
                       {(size_t)V_ - 1, (size_t)V_ - 1, (size_t)V_ - 1, (size_t)(V_ - P_), D},
                       {(size_t)V_ * 8, (size_t)V_ * 8, (size_t)V_ * 8, (size_t)V_ * 8, D * 8}};
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
        return w->me[i]->msm_launch(a.bs[i], a.sc[i], a.cnt[i], true, 0, w->mw[i], w->k, a.stride[i], i != 4);
    }
    int enqueue_part_a(ProveWs *w, bool use_graphs) {
        int rc;
        const MsmArgs a = msm_args(w);
        MG_HIP(hipEventRecord(w->fork, w->stream)); // z is on the device (upload_z ran on this stream)
        if ((rc = launch_witness_map(w, use_graphs))) return rc;
        for (int i = 0; i < 5; ++i) {
            if (!in_part_a(i)) continue;
            hipStream_t ms = msm_stream(w, i);
            if (ms != w->stream) MG_HIP(hipStreamWaitEvent(ms, i == 4 ? w->h_ready : w->fork, 0));
            if ((rc = enqueue_msm(w, a, i, use_graphs))) return rc;
        }
        for (int i = 0; i < 5; ++i) { // join (after every launch, so that no MSM on the main stream queues behind a wait)
            hipStream_t ms = msm_stream(w, i);
            if (in_part_a(i) && ms != w->stream) MG_HIP(hipStreamWaitEvent(w->stream, w->mw[i]->done, 0));
        }
        return MG_OK;
    }
    int enqueue_part_b(ProveWs *w, bool use_graphs) { return enqueue_msm(w, msm_args(w), 2, use_graphs); }

    int enqueue_proof(ProveWs *w, const uint64_t *z_src, bool use_graphs) {
        int rc = upload_z(w, z_src);
        if (rc) return rc;
        hipStream_t g2s = msm_stream(w, 2);
        if (w->g_all && w->g_g2) { // "single" mode replay
            // the G2 graph goes first: it is the longest chain and its launch is the cheaper of the two
            // (measured: 1.47 ms per PrivateTransfer proof against 1.68 with the other order)
            if (g2s != w->stream) MG_HIP(hipStreamWaitEvent(g2s, w->z_ready, 0));
            MG_HIP(hipGraphLaunch(w->g_g2, g2s));
            MG_HIP(hipGraphLaunch(w->g_all, w->stream));
            for (int i = 0; i < 5; ++i) w->mw[i]->pending = 1;
            return MG_OK;
        }
        if ((rc = enqueue_part_a(w, use_graphs))) return rc;
        if (g2s != w->stream) MG_HIP(hipStreamWaitEvent(g2s, w->z_ready, 0));
        return enqueue_part_b(w, use_graphs);
    }

    // capture one single-stream segment into an executable graph
    template <class Fn> static bool capture_segment(hipStream_t s, hipGraphExec_t *out, Fn &&body) {
        if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) != hipSuccess) return false;
        const int rc = body();
        hipGraph_t graph = nullptr;
        const hipError_t e = hipStreamEndCapture(s, &graph);
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
        if (graph_mode() == GRAPH_SINGLE) {
            bool ok1 = capture_segment(w->stream, &w->g_all, [&] { return enqueue_part_a(w, false); });
            if (ok1) {
                w->mw[2]->capturing = true; // a linear capture: nothing waits on its `done` event
                ok1 = capture_segment(msm_stream(w, 2), &w->g_g2, [&] { return enqueue_part_b(w, false); });
                w->mw[2]->capturing = false;
            }
            for (int i = 0; i < 5; ++i) w->mw[i]->pending = 0;
            if (!ok1) {
                w->drop_graphs();
                w->no_graph = true;
            }
            w->graphs_ready = ok1;
            return ok1;
        }
        bool ok = capture_segment(w->stream, &w->g_wm, [&] { return enqueue_witness_map_body(w); });
        const MsmArgs a = msm_args(w);
        for (int i = 0; ok && i < 5; ++i) {
            w->mw[i]->capturing = true; // no event records inside the capture: the replay path records `done`
            ok = capture_segment(msm_stream(w, i), &w->g_msm[i], [&] {
                return w->me[i]->msm_launch(a.bs[i], a.sc[i], a.cnt[i], true, 0, w->mw[i], w->k, a.stride[i], i != 4);
            });
            w->mw[i]->capturing = false;
            w->mw[i]->pending = 0;
        }
        if (!ok) {
            w->drop_graphs();
            w->no_graph = true;
        }
        w->graphs_ready = ok;
        return ok;
    }

    int prove(const uint64_t *z, const uint64_t *r, const uint64_t *s, uint8_t *proof_out) override {
        return prove_batch(1, z, r, s, proof_out);
    }

    // k proofs of this circuit in ONE pass of the GPU pipeline (k = 1: a single proof). The kernels are the same;
    // every (assignment, window) pair is its own bucket segment and the NTT / SpMV grids get a batch dimension,
    // so a batch costs one chain of latency-bound launches instead of k. (Splitting a batch into two passes in
    // flight, to assemble one half on the host while the GPU works on the other, was measured and gains nothing:
    // the smaller passes lose what the overlap wins. Two calling threads with a batch each do overlap.)
    int prove_batch(u64 k64, const uint64_t *z, const uint64_t *r, const uint64_t *s, uint8_t *proofs_out) override {
        if (!have_r1cs_) return MG_ERR_STATE;
        if (k64 == 0 || k64 > 1024 || !z || !r || !s || !proofs_out) return MG_ERR_ARG;
        Pass p;
        const int rc = launch_pass(p, (u32)k64, z, r, s, proofs_out);
        return finish_pass(p, rc);
    }

    struct Pass {
        ProveWs *w = nullptr;
        u32 k = 0;
        const uint64_t *r = nullptr, *s = nullptr;
        uint8_t *out = nullptr;
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
    int launch_pass(Pass &p, u32 k, const uint64_t *z, const uint64_t *r, const uint64_t *s, uint8_t *out) {
        p.k = k, p.r = r, p.s = s, p.out = out;
        ProveWs *w = p.w = ws_acquire(k);
        if (!w) return MG_ERR_HIP;
        int rc = MG_OK;
        const size_t zbytes = (size_t)k * V_ * 32;
        // the assignment is uploaded from where it is if the caller keeps it in page-locked memory
        // (mg_host_alloc), else through the slot's pinned staging copy
        const uint64_t *z_src = z;
        if (!is_page_locked(z)) {
            if (w->h_z_cap < zbytes) {
                if (w->h_z) hipHostFree(w->h_z);
                w->h_z = nullptr;
                w->h_z_cap = 0;
                if (hipHostMalloc(&w->h_z, zbytes, hipHostMallocDefault) != hipSuccess) return MG_ERR_OOM;
                w->h_z_cap = zbytes;
            }
            std::memcpy(w->h_z, z, zbytes);
            z_src = (const uint64_t *)w->h_z;
        }
        if (!w->graphs_ready && graphs_enabled() && !w->no_graph && w->eager_runs >= 2) build_graphs(w);
        if (w->graphs_ready) {
            rc = enqueue_proof(w, z_src, graph_mode() == GRAPH_SPLIT);
            if (rc) { // do not trust the graphs again; the failed pass is reported to the caller
                hipStreamSynchronize(w->stream);
                hipStreamSynchronize(msm_stream(w, 2));
                w->drop_graphs();
                w->no_graph = true;
            }
        } else {
            rc = enqueue_proof(w, z_src, false);
            w->eager_runs++;
        }
        return rc;
    }

    // host side of a pass: blinding terms while the GPU works, wait, fold the MSM results, assemble and encode
    int finish_pass(Pass &p, int rc) {
        ProveWs *w = p.w;
        if (!w) return rc ? rc : MG_ERR_HIP;
        const u32 k = p.k;
        const uint64_t *r = p.r, *s = p.s;
        // ---- host work that does not depend on the MSMs runs while the GPU is busy: the blinding terms
        // r*delta_g1, s*delta_g1, (r s)*delta_g1, s*delta_g2 are fixed-base (64 table additions each)
        struct Blind {
            u64 rc4[4], sc4[4], rs4[4];
            HostPoint t_rd, t_sd, t_rsd, t_sd2;
        };
        std::vector<Blind> bl(k);
        if (!rc) {
            for (u32 q = 0; q < k; ++q) {
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
            }
        }
        std::vector<HostPoint> res((size_t)5 * k); // res[i * k + q]: MSM i of proof q
        auto collect = [&](hipStream_t s, bool part_a) { // wait for one part and fold its MSMs
            hipError_t e = hipStreamSynchronize(s);
            if (e != hipSuccess && !rc) {
                set_last_hip_error(e, "prove: hipStreamSynchronize", __FILE__, __LINE__);
                rc = MG_ERR_HIP;
            }
            for (int i = 0; i < 5; ++i) {
                if (in_part_a(i) != part_a) continue;
                if (w->mw[i]->pending) {
                    int rc2 = w->me[i]->msm_finish(w->mw[i], &res[(size_t)i * k], true);
                    if (!rc) rc = rc2;
                } else {
                    hipStreamSynchronize(msm_stream(w, i));
                    if (!rc) rc = MG_ERR_STATE;
                }
            }
        };
        const int b1 = g1_->point_bytes(true), b2 = g2_->point_bytes(true);
        // ---- part A is back: the G1 side of the assembly runs while the G2 MSM finishes
        collect(w->stream, true); // every G1 MSM stream has been joined into it
        if (!rc) {
            for (u32 q = 0; q < k; ++q) {
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
                uint8_t *out = p.out + (size_t)q * (2 * b1 + b2);
                g1_->hp_serialize(&g_a, out, true);
                g1_->hp_serialize(&g_c, out + b1 + b2, true);
            }
        }
        // ---- part B: the G2 element
        collect(msm_stream(w, 2), false);
        ws_release(w);
        p.w = nullptr;
        if (rc) return rc;
        for (u32 q = 0; q < k; ++q) {
            HostPoint g2_b = res[2 * (size_t)k + q];
            g2_->hp_add(&g2_b, &b20_beta_);
            g2_->hp_add(&g2_b, &bl[q].t_sd2);
            g2_->hp_serialize(&g2_b, p.out + (size_t)q * (2 * b1 + b2) + b1, true);
        }
        return MG_OK;
    }
};

} // namespace

int prover_create(int curve, const mg_pk_view *pk, Prover **out) {
    ProverImpl *p = new ProverImpl();
    int rc = p->init(curve, pk);
    if (rc) {
        delete p;
        return rc;
    }
    *out = p;
    return MG_OK;
}

} // namespace mg
