// Groth16 prover, part 4: ProverSlots -- acquiring / releasing proof slots, the witness map and the five MSMs of a pass enqueued on a
// slot, capture of the slot's hipGraphs.
// Included by prover.cpp only (one translation unit: the anonymous namespace is intended).
#pragma once

namespace mg {
namespace {

class ProverSlots : public ProverAssembly {
  public:
    // How a SINGLE proof's slot replays (ProveWs::linear3). 0: the forked graph. 1: three linear graphs -- witness map + h | a|b_g1|l
    // | G2 -- on three high-priority streams: the shortest chain for a LONE proof (company 0: no other pass of this context in
    // flight). 2 / 3: the same beside other passes, with ONE chain on a normal-priority stream -- the combined MSM beside a batched
    // pass (company 2), the G2 MSM beside single proofs only (company 1: two host threads). The streams come from
    // stream_set_acquire (runtime.cpp): three DIFFERENT hardware queues per slot, and the two slots that two host threads keep in
    // flight share none -- before, which chains of the two proofs met on one queue was decided by the order in which the process
    // had created its streams (profiles/r05_hw_queues.txt). A normal-priority chain costs a lone proof 18 %: flavour 1 keeps all
    // three high; which chain yields beside others was measured per company (same file, item 7).
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
            w->mw[i]->in_graph_slot = true; // (a slot's launches are captured: the engine's side stream must never join them)
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
        // (passes of one proof only: on full tables passes of 2-8 proofs are slower than on the narrow bucket tables)
        return z3_bs_full_ && k == 1 && full_max_k >= 1 && peers_.empty() && !has_exchange_;
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
            // (the forked form; single proofs of z3 slots take the three linear graphs above since the memset-node defect was found:
            // profiles/r05_linear_graph_defect.txt)
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

};

} // namespace
} // namespace mg
