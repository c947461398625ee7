// Groth16 prover, part 2: ProverKey -- the device-resident proving key (bucket / wide / full tables), the circuit (CSR matrices, domain),
// table planning and mg_ctx_set_r1cs (staged, all shards or none).
// Included by prover.cpp only (one translation unit: the anonymous namespace is intended).
#pragma once

namespace mg {
namespace {

class ProverKey : public Prover {
  public:
    // what the deployment decided for THIS context (mg_ctx_opts.tuning, else the process-wide values when it was created): tuning.h
    Tuning tn_ = tuning();
    bool has_exchange_ = false; // an in-library RCCL exchange serves this (sharded) context (ProverImpl::exchange_init)
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
    ProverKey *shard_owner_ = nullptr; // in-process peers: the shard-0 object that owns this one
    std::vector<ProverKey *> peers_;  // shard 0 only: shards 1 .. G-1 (owned); a pass runs on all of them, shard 0 assembles
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

    ~ProverKey() override { // (ProverImpl's destructor has run: the exchange is gone, the peer shards are deleted)
        HeavyOp no_capture_meanwhile;
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
        for (const ProverKey *q : peers_) {
            u64 t[2];
            q->table_bytes(t);
            out2[0] += t[0], out2[1] += t[1];
        }
    }

    // window bits for precomputed tables, by MSM length (HBM is plentiful: trade table size for fewer
    // buckets to fold and no doubling chain -- tuned on MI355X)
    int pre_c_for(u64 n) const {
        if (tn_.window_bits_narrow > 0) return tn_.window_bits_narrow; // tuning override: ONE width for every bucket table of the key
        // proof-sized queries: few buckets keep the latency-bound bucket reduce short (B = 128: two tiles); the extra windows only
        // add perfectly parallel mixed additions
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
        if (small) { // batched passes are throughput-bound: wider windows = fewer mixed additions
            // (11 everywhere until round 6: from 2^15 scalars on -- PrivateTransfer -- the wider windows pay since the front levels of
            // the bucket reduce run inside the graphs; the smaller shapes keep 11: profiles/r06_batched_windows_front_levels.txt)
            int cw = zn >= (1u << 15) ? 12 : 11;
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
        std::vector<ProverKey *> all{this};
        all.insert(all.end(), peers_.begin(), peers_.end());
        // The exclusive locks are taken BEFORE staging as well: staging uploads with synchronous copies, builds window tables
        // on the default stream and ends in hipDeviceSynchronize, and the HIP runtime fails the graph capture of a proof
        // slot on another thread when that happens meanwhile (seen on MI355X: mg_groth16_prove returning a HIP error while a
        // circuit was being staged). With every shard locked no pass is in flight, none starts, nothing is capturing.
        std::vector<std::unique_lock<std::shared_mutex>> locks;
        for (ProverKey *q : all) locks.emplace_back(q->shape_mu_);
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
            // c = 8. Wider windows halve them, but lengthen its bucket reduce; for single proofs the reduce chain matters more.
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

};

} // namespace
} // namespace mg
