// Groth16 prover, part 3: ProverAssembly -- blinding terms and the host assembly of (A, B, C) from the MSM results, arkworks bytes.
// Included by prover.cpp only (one translation unit: the anonymous namespace is intended).
#pragma once

namespace mg {
namespace {

class ProverAssembly : public ProverKey {
  public:
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

};

} // namespace
} // namespace mg
