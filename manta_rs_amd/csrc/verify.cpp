// Groth16 verification on one MI355X.
//
// Replaces `Groth16::verify` (manta-crypto/src/arkworks/groth16.rs:603-609 -> ark-groth16 0.3
// verify_with_processed_vk):
//     prepared_inputs = gamma_abc_g1[0] + sum_j input_j gamma_abc_g1[j+1]
//     final_exponentiation( ML(A, B) . ML(prepared_inputs, -gamma_g2) . ML(C, -delta_g2) ) == e(alpha_g1, beta_g2)
// and the `VerifyingContext` codec (groth16.rs:337-539): the wire format carries the prepared key -- vk, e(alpha, beta),
// and the G2Prepared line coefficients of -gamma_g2 and -delta_g2 -- which this file both parses and PRODUCES: a context
// built from the five key components recomputes all of it on the GPU, and `encode` then yields the reference's file bytes.
//
// Batch verification (the throughput companion of mg_groth16_prove_batch; ledger use manta-pay/src/simulation/ledger/
// mod.rs:626-651): for caller-drawn 128-bit r_i,
//     prod_i ML(r_i A_i, B_i) . ML(sum_i r_i PI_i, -gamma) . ML(sum_i r_i C_i, -delta) . ML(-(sum_i r_i) alpha, beta)  -> 1
// i.e. k + 3 Miller loops (one wavefront each, pairing_coop.h), two small MSMs and ONE final exponentiation for k proofs.
#include "verify.h"
#include "tuning.h"
#include "host_ec.h"
#include "params_gen.h"
#include "prover.h"
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

namespace mg {

PairingEngine *get_pairing_engine(int curve) {
    static std::mutex mu;
    static PairingEngine *tab[MAX_DEVICES][2] = {};
    if (curve < 0 || curve > 1) return nullptr;
    const int dev = current_device();
    std::lock_guard<std::mutex> g(mu);
    if (!tab[dev][curve]) tab[dev][curve] = curve == 0 ? make_pairing_engine_bn254() : make_pairing_engine_bls381();
    return tab[dev][curve];
}

namespace {

using host::HFp;
using host::HFp2;
using host::HPoint;

template <class C> HFp<C> hpow(const HFp<C> &a, const u64 *e, int nl) {
    HFp<C> acc = HFp<C>::one();
    for (int i = 64 * nl - 1; i >= 0; --i) {
        acc = HFp<C>::sqr(acc);
        if ((e[i >> 6] >> (i & 63)) & 1) acc = HFp<C>::mul(acc, a);
    }
    return acc;
}
template <class C> void p_words(u64 *p) {
    for (int i = 0; i < C::N64; ++i) p[i] = (u64)C::P[2 * i] | ((u64)C::P[2 * i + 1] << 32);
}
// square root in Fq, q = 3 (mod 4) for both curves: a^((q+1)/4), checked
template <class C> bool hsqrt(const HFp<C> &a, HFp<C> &r) {
    static_assert((C::P[0] & 3) == 3, "q = 3 mod 4");
    u64 e[C::N64];
    p_words<C>(e);
    u64 cy = 1; // (q + 1) >> 2
    for (int i = 0; i < C::N64 && cy; ++i) {
        e[i] += cy;
        cy = e[i] == 0;
    }
    for (int i = 0; i < C::N64; ++i) e[i] = (e[i] >> 2) | (i + 1 < C::N64 ? e[i + 1] << 62 : 0);
    r = hpow<C>(a, e, C::N64);
    return HFp<C>::sqr(r) == a;
}
// square root in Fq2 = Fq[u]/(u^2 + 1) (complex method)
template <class C> bool hsqrt2(const HFp2<C> &a, HFp2<C> &r) {
    typedef HFp<C> B;
    if (a.c1.is_zero()) {
        B s;
        if (hsqrt<C>(a.c0, s)) {
            r = HFp2<C>{s, B::zero()};
            return true;
        }
        if (hsqrt<C>(B::neg(a.c0), s)) {
            r = HFp2<C>{B::zero(), s};
            return true;
        }
        return false;
    }
    B n = B::add(B::sqr(a.c0), B::sqr(a.c1)), s;
    if (!hsqrt<C>(n, s)) return false;
    B two_inv = B::inv(B::dbl(B::one()));
    B t = B::mul(B::add(a.c0, s), two_inv), x0;
    if (!hsqrt<C>(t, x0)) {
        t = B::mul(B::sub(a.c0, s), two_inv);
        if (!hsqrt<C>(t, x0)) return false;
    }
    B x1 = B::mul(a.c1, B::inv(B::dbl(x0)));
    r = HFp2<C>{x0, x1};
    return HFp2<C>::sqr(r).c0 == a.c0 && HFp2<C>::sqr(r).c1 == a.c1;
}
template <class C> bool read_fq(const uint8_t *in, unsigned char mask_top, HFp<C> &out) { // canonical LE bytes -> Montgomery
    typedef HFp<C> HF;
    HF c = HF::zero();
    for (int i = 0; i < HF::BYTES; ++i) {
        uint8_t v = in[i];
        if (i == HF::BYTES - 1) v &= (uint8_t)~mask_top;
        c.v[i >> 3] |= (u64)v << ((i & 7) * 8);
    }
    if (HF::geq_p(c.v)) return false;
    out = HF::to_mont(c);
    return true;
}

template <class Curve, class K> class VerifierT : public Verifier {
  public:
    GraphClient counted_; // (engine.h: stand-alone MSMs keep to ordinary streams while a context lives)
    typedef typename Curve::Fq C;
    typedef typename Curve::Fr CR;
    typedef HFp<C> HF;
    typedef HFp2<C> HF2;
    typedef HFp<CR> HR;
    static constexpr int N64 = C::N64, G1L = 2 * N64, G2L = 4 * N64; // u64 limbs per affine point
    static constexpr int FB = HF::BYTES;
    int curve_ = 0, dev_ = 0;
    GroupEngine *g1_ = nullptr, *g2_ = nullptr;
    PairingEngine *pe_ = nullptr;
    u64 P_ = 0;
    std::vector<u64> alpha_, beta_, gamma_, delta_, abc_;           // affine Montgomery limbs
    std::vector<u32> alpha_beta_, gneg_host_, dneg_host_;           // e(alpha, beta); line coefficients (Montgomery words)
    u32 *d_gneg_ = nullptr, *d_dneg_ = nullptr, *d_beta_ = nullptr; // prepared -gamma, -delta, beta on the device
    BaseSet *abc_bs_ = nullptr;                                     // gamma_abc_g1[0..P) as MSM bases
    std::mutex mu_;
    // G1's endomorphism phi(x, y) = (beta x, y) = lambda (x, y) (beta^3 = 1 in Fq, lambda^3 = 1 in Fr): the batch check draws
    // its coefficients as k1 + lambda k2 with 64-bit k1, k2 -- as many distinct values as 128-bit integers (the lattice of
    // pairs with k1 + lambda k2 = 0 mod r has no vector shorter than ~2^127), and k_i A_i is then one chain of 64 doublings
    // instead of 128. The pair (lambda, beta) is checked against alpha_g1 when the context is made.
    bool glv_ok_ = false;
    HR glv_lambda_;                  // Montgomery
    std::vector<u32> glv_beta_std_;  // arkworks-format words
    static void glv_constants(u64 lambda[4], u64 *beta) {
        if constexpr (N64 == 4) { // BN254
            const u64 l[4] = {0x8b17ea66b99c90ddull, 0x5bfc41088d8daaa7ull, 0xb3c4d79d41a91758ull, 0};
            const u64 b[4] = {0x5763473177fffffeull, 0xd4f263f1acdb5c4full, 0x59e26bcea0d48bacull, 0};
            std::memcpy(lambda, l, 32), std::memcpy(beta, b, 32);
        } else { // BLS12-381
            static_assert(N64 == 4 || N64 == 6, "BN254 or BLS12-381");
            const u64 l[4] = {0x00000000ffffffffull, 0xac45a4010001a402ull, 0, 0};
            const u64 b[6] = {0x8bfd00000000aaacull, 0x409427eb4f49fffdull, 0x897d29650fb85f9bull,
                              0xaa0d857d89759ad4ull, 0xec02408663d4de85ull, 0x1a0111ea397fe699ull};
            std::memcpy(lambda, l, 32), std::memcpy(beta, b, 48);
        }
    }
    void glv_init() {
        glv_ok_ = false;
        if (alpha_.size() != (size_t)G1L || is_zero_limbs(alpha_.data(), G1L)) return;
        u64 lam[4], bet[N64];
        glv_constants(lam, bet);
        HF b;
        std::memcpy(b.v, bet, sizeof(bet));
        b = HF::to_mont(b);
        HostPoint p;
        g1_->hp_from_affine(&p, (const u32 *)alpha_.data());
        g1_->hp_mul(&p, lam);
        std::vector<u64> lp(G1L), want(alpha_);
        g1_->hp_to_affine(&p, (u32 *)lp.data());
        HF ax;
        std::memcpy(ax.v, alpha_.data(), N64 * 8);
        ax = HF::mul(ax, b);
        std::memcpy(want.data(), ax.v, N64 * 8);
        if (lp != want) return; // (never with the constants above; the batch check then multiplies by plain 128-bit coefficients)
        HR l;
        std::memcpy(l.v, lam, 32);
        glv_lambda_ = HR::to_mont(l);
        glv_beta_std_.resize(2 * N64);
        b.store_words(glv_beta_std_.data());
        glv_ok_ = true;
    }

    ~VerifierT() override {
        int prev = 0;
        hipGetDevice(&prev);
        hipSetDevice(dev_);
        if (d_gneg_) hipFree(d_gneg_);
        if (d_dneg_) hipFree(d_dneg_);
        if (d_beta_) hipFree(d_beta_);
        if (abc_bs_) g1_->bases_destroy(abc_bs_);
        hipSetDevice(prev);
    }
    u64 n_inputs() const override { return P_; }

    static HF b1() {
        HF b;
        b.load_words(Curve::G1_B);
        return b;
    }
    static HF2 b2() {
        HF2 b;
        b.c0.load_words(Curve::G2_B0);
        b.c1.load_words(Curve::G2_B1);
        return b;
    }
    static void r_words(u64 *r) { p_words<CR>(r); }
    // ---- point decoding (ark-serialize 0.3 compressed short-Weierstrass: flags in the two top bits of the last byte,
    // bit 6 = infinity, bit 7 = "y is the lexicographically larger root"); checked like `CanonicalDeserialize::deserialize`
    static bool g1_decompress(const uint8_t *in, u64 *out) {
        const uint8_t flags = in[FB - 1];
        HF x, y;
        // ark-serialize 0.3 reads the field element first (rejecting x >= p) and the flags with it: `SWFlags::from_u8` has no
        // value for "both bits set", and an infinity encoding still has to carry a canonical x (its value is ignored)
        if ((flags & 0xC0) == 0xC0) return false;
        if (!read_fq<C>(in, 0xC0, x)) return false;
        if (flags & 0x40) {
            std::memset(out, 0, G1L * 8);
            return true;
        }
        if (!hsqrt<C>(HF::add(HF::mul(HF::sqr(x), x), b1()), y)) return false;
        if (y.is_high() != ((flags & 0x80) != 0)) y = HF::neg(y);
        std::memcpy(out, x.v, N64 * 8);
        std::memcpy(out + N64, y.v, N64 * 8);
        u64 r[4];
        r_words(r);
        HPoint<HF> p{x, y, HF::one(), HF::one()};
        return HPoint<HF>::mul(p, r, 4).is_inf(); // subgroup check (trivial cofactor on BN254, not on BLS12-381)
    }
    static bool g2_decompress(const uint8_t *in, u64 *out) {
        const uint8_t flags = in[2 * FB - 1];
        HF2 x, y;
        if ((flags & 0xC0) == 0xC0) return false; // no such SWFlags value
        if (!read_fq<C>(in, 0, x.c0) || !read_fq<C>(in + FB, 0xC0, x.c1)) return false; // canonical even when infinity is flagged
        if (flags & 0x40) {
            std::memset(out, 0, G2L * 8);
            return true;
        }
        if (!hsqrt2<C>(HF2::add(HF2::mul(HF2::sqr(x), x), b2()), y)) return false;
        if (y.is_high() != ((flags & 0x80) != 0)) y = HF2::neg(y);
        std::memcpy(out, x.c0.v, N64 * 8);
        std::memcpy(out + N64, x.c1.v, N64 * 8);
        std::memcpy(out + 2 * N64, y.c0.v, N64 * 8);
        std::memcpy(out + 3 * N64, y.c1.v, N64 * 8);
        u64 r[4];
        r_words(r);
        HPoint<HF2> p{x, y, HF2::one(), HF2::one()};
        return HPoint<HF2>::mul(p, r, 4).is_inf();
    }
    static void neg_g2(const u64 *q, u64 *out) { // (x, -y); infinity stays infinity
        std::memcpy(out, q, G2L * 8);
        HF2 y;
        std::memcpy(y.c0.v, q + 2 * N64, N64 * 8);
        std::memcpy(y.c1.v, q + 3 * N64, N64 * 8);
        y = HF2::neg(y);
        std::memcpy(out + 2 * N64, y.c0.v, N64 * 8);
        std::memcpy(out + 3 * N64, y.c1.v, N64 * 8);
    }
    static bool is_zero_limbs(const u64 *p, int n) {
        u64 x = 0;
        for (int i = 0; i < n; ++i) x |= p[i];
        return x == 0;
    }

    int common_init(int curve) {
        curve_ = curve;
        dev_ = current_device();
        g1_ = get_engine(curve, 1);
        g2_ = get_engine(curve, 2);
        pe_ = get_pairing_engine(curve);
        if (!g1_ || !g2_ || !pe_) return MG_ERR_ARG;
        // gamma_abc_g1 with FULL tables (every multiple of every 6-bit window: 2.4 MB for 27 inputs): the prepared-inputs MSM is
        // then digits + one accumulate + one merge kernel and its point is back on the host 140 us after the call instead of 205
        // -- the third Miller loop, which waits for it, no longer ends after the other two (tools/abc_sweep_r4.sh;
        // MANTA_VERIFY_ABC_C = 0: plain bases as in round 3, c > 0: window tables, c < 0: full tables of |c|-bit windows)
        static const int abc_c = [] {
            return ab_knob("MANTA_VERIFY_ABC_C", -6);
        }();
        int rc = g1_->bases_create((const u32 *)abc_.data(), P_, false, abc_c, &abc_bs_);
        if (rc) return rc;
        glv_init();
        return pe_->prepare((const u32 *)beta_.data(), 1, &d_beta_);
    }

    // from the five key components: everything `ArkGroth16::process_vk` derives is computed on the GPU
    int init_from_points(int curve, const u64 *alpha, const u64 *beta, const u64 *gamma, const u64 *delta, const u64 *abc, u64 P) {
        if (!alpha || !beta || !gamma || !delta || !abc || P < 1 || P > (1u << 24)) return MG_ERR_ARG;
        P_ = P;
        alpha_.assign(alpha, alpha + G1L), beta_.assign(beta, beta + G2L), gamma_.assign(gamma, gamma + G2L);
        delta_.assign(delta, delta + G2L), abc_.assign(abc, abc + P * G1L);
        int rc = common_init(curve);
        if (rc) return rc;
        std::vector<u64> ng(2 * G2L);
        neg_g2(gamma, ng.data());
        neg_g2(delta, ng.data() + G2L);
        u32 *d2 = nullptr;
        if ((rc = pe_->prepare((const u32 *)ng.data(), 2, &d2))) return rc;
        const size_t cw = (size_t)pe_->n_coeffs() * pe_->coeff_words();
        gneg_host_.resize(cw), dneg_host_.resize(cw);
        hipError_t e = memcpy_sync(gneg_host_.data(), d2, cw * 4, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = memcpy_sync(dneg_host_.data(), d2 + cw, cw * 4, hipMemcpyDeviceToHost);
        d_gneg_ = d2; // the two blocks stay in this one allocation
        d_dneg_ = nullptr;
        if (e != hipSuccess) {
            set_last_hip_error(e, "verifier init", __FILE__, __LINE__);
            return MG_ERR_HIP;
        }
        // e(alpha_g1, beta_g2) with arkworks' final exponentiation
        alpha_beta_.resize(pe_->f12_words());
        const u32 *cp[1] = {d_beta_};
        unsigned char skip[1] = {(unsigned char)is_zero_limbs(beta, G2L)};
        return pe_->pairing_product((const u32 *)alpha_.data(), cp, nullptr, skip, 1, true, alpha_beta_.data());
    }
    const u32 *d_gamma_neg() const { return d_gneg_; }
    const u32 *d_delta_neg() const { return d_dneg_ ? d_dneg_ : d_gneg_ + (size_t)pe_->n_coeffs() * pe_->coeff_words(); }

    // from the wire format: the prepared parts are taken from the bytes, as `VerifyingContext::deserialize` does
    int init_from_bytes(int curve, const uint8_t *in, size_t len) {
        const size_t g1b = FB, g2b = 2 * FB;
        size_t off = 0;
        auto need = [&](size_t n) { return len - off >= n; };
        if (!need(g1b + 3 * g2b + 8)) return MG_ERR_ARG;
        alpha_.resize(G1L), beta_.resize(G2L), gamma_.resize(G2L), delta_.resize(G2L);
        if (!g1_decompress(in + off, alpha_.data())) return MG_ERR_ARG;
        off += g1b;
        u64 *g2s[3] = {beta_.data(), gamma_.data(), delta_.data()};
        for (int k = 0; k < 3; ++k) {
            if (!g2_decompress(in + off, g2s[k])) return MG_ERR_ARG;
            off += g2b;
        }
        u64 P = 0;
        for (int i = 0; i < 8; ++i) P |= (u64)in[off + i] << (8 * i);
        off += 8;
        if (P < 1 || P > (len - off) / g1b) return MG_ERR_ARG;
        P_ = P;
        abc_.resize(P * G1L);
        for (u64 j = 0; j < P; ++j) {
            if (!g1_decompress(in + off, abc_.data() + j * G1L)) return MG_ERR_ARG;
            off += g1b;
        }
        int rc = common_init(curve);
        if (rc) return rc;
        const size_t nco = (size_t)pe_->n_coeffs(), cw = nco * pe_->coeff_words();
        if (!need(12 * (size_t)FB)) return MG_ERR_ARG;
        alpha_beta_.resize(pe_->f12_words());
        for (int k = 0; k < 12; ++k) {
            HF v;
            if (!read_fq<C>(in + off, 0, v)) return MG_ERR_ARG;
            v.store_words(alpha_beta_.data() + (size_t)k * 2 * N64);
            off += FB;
        }
        std::vector<u32> *dst[2] = {&gneg_host_, &dneg_host_};
        for (int b = 0; b < 2; ++b) {
            if (!need(8)) return MG_ERR_ARG;
            u64 cnt = 0;
            for (int i = 0; i < 8; ++i) cnt |= (u64)in[off + i] << (8 * i);
            off += 8;
            if (cnt != nco || !need(cnt * 6 * FB + 1)) return MG_ERR_ARG;
            dst[b]->resize(cw);
            for (size_t k = 0; k < cnt * 6; ++k) {
                HF v;
                if (!read_fq<C>(in + off, 0, v)) return MG_ERR_ARG;
                v.store_words(dst[b]->data() + k * 2 * N64);
                off += FB;
            }
            if (in[off++] != 0) return MG_ERR_ARG; // G2Prepared.infinity: gamma, delta are never infinity in a usable key
        }
        if (off != len) return MG_ERR_ARG;
        MG_HIP(hipMalloc((void **)&d_gneg_, 2 * cw * 4));
        MG_HIP(memcpy_sync(d_gneg_, gneg_host_.data(), cw * 4, hipMemcpyHostToDevice));
        MG_HIP(memcpy_sync(d_gneg_ + cw, dneg_host_.data(), cw * 4, hipMemcpyHostToDevice));
        return MG_OK;
    }

    // sum_j scalars_j * gamma_abc_g1[j] on the GPU (scalars Montgomery Fr, n <= P)
    int abc_msm(const u64 *scalars_mont, size_t n, HostPoint *out) {
        MsmWorkspace *ws = g1_->ws_acquire();
        if (!ws) return MG_ERR_HIP;
        // the scalars ride in the workspace's own grow-only buffer, on its stream: no hipMalloc / hipFree per verification
        // (hipFree synchronises the whole device -- every proof in flight on this GPU would wait for it)
        int rc = ws->scratch.reserve(n * 32);
        if (!rc && hipMemcpyAsync(ws->scratch.p, scalars_mont, n * 32, hipMemcpyHostToDevice, ws->stream) != hipSuccess) rc = MG_ERR_HIP;
        if (!rc) rc = g1_->msm_launch(abc_bs_, ws->scratch.as<u32>(), n, SCALARS_MONT, 0, ws);
        if (!rc) rc = g1_->msm_finish(ws, out);
        else hipStreamSynchronize(ws->stream), ws->pending = 0;
        g1_->ws_release(ws);
        return rc;
    }

    int verify(const u64 *inputs, const u64 *proof, int *ok) override {
        if ((!inputs && P_ > 1) || !proof || !ok) return MG_ERR_ARG;
        DeviceScope on_device(dev_);
        *ok = 0;
        const u64 *A = proof, *B = proof + G1L, *Cc = proof + G1L + G2L;
        // e(A, B) and e(C, -delta) do not depend on the public inputs: their Miller loops start first. B is the one G2 point that
        // is new with every proof: its line coefficients are computed next to its Miller loop
        const u32 *cp[3] = {nullptr, d_delta_neg(), d_gamma_neg()};
        std::vector<u64> ps(2 * G1L);
        std::memcpy(ps.data(), A, G1L * 8);
        std::memcpy(ps.data() + G1L, Cc, G1L * 8);
        std::vector<u64> qs(3 * G2L, 0);
        std::memcpy(qs.data(), B, G2L * 8);
        unsigned char skip[3] = {(unsigned char)is_zero_limbs(B, G2L), 0, 0};
        void *pp = nullptr;
        int rc = pe_->pairing_product_begin((const u32 *)ps.data(), cp, (const u32 *)qs.data(), skip, 3, 2, &pp);
        if (rc) return rc;
        // prepared_inputs = abc[0] + sum_j input_j abc[j+1]: a P-term MSM with the scalar 1 in front, next to those loops
        std::vector<u64> sc(P_ * 4);
        HR one = HR::one();
        std::memcpy(sc.data(), one.v, 32);
        if (P_ > 1) std::memcpy(sc.data() + 4, inputs, (P_ - 1) * 32);
        HostPoint pi;
        rc = abc_msm(sc.data(), P_, &pi);
        if (rc) {
            pe_->pairing_product_abandon(pp);
            return rc;
        }
        std::vector<u64> late(G1L);
        g1_->hp_to_affine(&pi, (u32 *)late.data());
        std::vector<u32> out(pe_->f12_words());
        rc = pe_->pairing_product_end(pp, (const u32 *)late.data(), true, out.data()); // + e(prepared_inputs, -gamma)
        if (rc) return rc;
        *ok = std::memcmp(out.data(), alpha_beta_.data(), out.size() * 4) == 0;
        return MG_OK;
    }

    int verify_batch(u64 k, const u64 *inputs, const u64 *proofs, const u64 *rand128, int *ok) override {
        if (k == 0 || k > (1u << 20) || (!inputs && P_ > 1) || !proofs || !rand128 || !ok) return MG_ERR_ARG;
        DeviceScope on_device(dev_);
        *ok = 0;
        const size_t PL = 2 * G1L + G2L; // limbs per proof
        // r_i as Montgomery Fr; s = sum r_i; comb_j = sum_i r_i x_ij
        std::vector<HR> r(k);
        HR s = HR::zero();
        std::vector<HR> comb(P_, HR::zero());
        std::vector<u64> r_can(k * 4, 0), r_full(k * 4, 0); // (k1, k2, 0, 0) per proof for the multiplication; the coefficient itself
        for (u64 i = 0; i < k; ++i) {
            HR c = HR::zero();
            c.v[0] = rand128[2 * i], c.v[1] = rand128[2 * i + 1];
            if ((c.v[0] | c.v[1]) == 0) return MG_ERR_ARG; // a zero coefficient would drop proof i from the check
            r_can[4 * i] = c.v[0], r_can[4 * i + 1] = c.v[1];
            if (glv_ok_) { // the coefficient of proof i is k1 + lambda k2 (k1, k2 = the two halves of its 128 random bits)
                HR k1 = HR::zero(), k2 = HR::zero();
                k1.v[0] = c.v[0], k2.v[0] = c.v[1];
                r[i] = HR::add(HR::to_mont(k1), HR::mul(glv_lambda_, HR::to_mont(k2)));
                const HR full = HR::from_mont(r[i]);
                std::memcpy(&r_full[4 * i], full.v, 32);
            } else {
                r[i] = HR::to_mont(c);
                std::memcpy(&r_full[4 * i], c.v, 32);
            }
            s = HR::add(s, r[i]);
        }
        std::vector<u64> cs(k * G1L), as(k * G1L), bs(k * G2L);
        for (u64 i = 0; i < k; ++i) {
            std::memcpy(as.data() + i * G1L, proofs + i * PL, G1L * 8);
            std::memcpy(bs.data() + i * G2L, proofs + i * PL + G1L, G2L * 8);
            std::memcpy(cs.data() + i * G1L, proofs + i * PL + G1L + G2L, G1L * 8);
        }
        // r_i A_i, element-wise on the GPU -- k independent 128-bit multiplications, one per lane: a millisecond of pure latency
        // on a handful of wavefronts. It is launched first, on a workspace stream of its own, and the two MSMs below (which fill
        // the rest of the machine for 0.4 ms) run next to it; the results come back as XYZZ points and the host turns them into
        // affine ones with ONE inversion (Montgomery's trick, ~0.1 ms), where the device spent 0.56 ms on Fermat inversions at
        // one-lane latency. (The base set of the C points is made before the launch: its allocation and synchronous upload would
        // otherwise wait for that kernel.)
        BaseSet *cb = nullptr;
        int rc = g1_->bases_create((const u32 *)cs.data(), k, false, 0, &cb);
        if (rc) return rc;
        MsmWorkspace *wse = g1_->ws_acquire();
        int rc_a = wse ? g1_->ec_mul_xyzz_begin((const u32 *)as.data(), (const u32 *)r_can.data(), k, wse, glv_ok_ ? glv_beta_std_.data() : nullptr)
                       : MG_ERR_HIP;
        if (rc_a) {
            if (wse) g1_->ws_release(wse);
            g1_->bases_destroy(cb);
            return rc_a;
        }
        for (u64 i = 0; i < k; ++i) // (k (P - 1) host products: next to the kernel just launched)
            for (u64 j = 1; j < P_; ++j) {
                HR x;
                std::memcpy(x.v, inputs + (i * (P_ - 1) + (j - 1)) * 4, 32);
                comb[j] = HR::add(comb[j], HR::mul(r[i], x));
            }
        comb[0] = s;
        HostPoint pi;
        rc = abc_msm((const u64 *)comb.data(), P_, &pi);
        // sum_i r_i C_i: a k-term MSM over the proofs' C points
        HostPoint csum;
        if (!rc) {
            MsmWorkspace *ws = g1_->ws_acquire();
            rc = ws ? ws->scratch.reserve(k * 32) : MG_ERR_HIP;
            if (!rc && hipMemcpyAsync(ws->scratch.p, r_full.data(), k * 32, hipMemcpyHostToDevice, ws->stream) != hipSuccess) rc = MG_ERR_HIP;
            if (!rc) rc = g1_->msm_launch(cb, ws->scratch.as<u32>(), k, SCALARS_CANONICAL, 0, ws);
            if (!rc) rc = g1_->msm_finish(ws, &csum);
            else if (ws) hipStreamSynchronize(ws->stream), ws->pending = 0;
            if (ws) g1_->ws_release(ws);
        }
        // -(sum r_i) alpha on the host (still next to that kernel)
        HostPoint al;
        g1_->hp_from_affine(&al, (const u32 *)alpha_.data());
        HR sc = HR::from_mont(s);
        g1_->hp_mul(&al, sc.v);
        g1_->hp_neg(&al);
        // pairs: (r_i A_i, B_i) ..., (PI, -gamma), (C, -delta), (-s alpha, beta); the three fixed ones are made affine (an inversion
        // each) while the multiplications still run. The k products r_i A_i never leave the device: the Miller kernel waits for
        // their kernel and reads them as (X, Y, ZZ, ZZZ) -- no download, inversion, upload between the two
        const size_t n = k + 3;
        std::vector<u64> ps(n * G1L, 0);
        if (!rc) {
            g1_->hp_to_affine(&pi, (u32 *)(ps.data() + k * G1L));
            g1_->hp_to_affine(&csum, (u32 *)(ps.data() + (k + 1) * G1L));
            g1_->hp_to_affine(&al, (u32 *)(ps.data() + (k + 2) * G1L));
        }
        std::vector<u32> out(pe_->f12_words());
        if (!rc) {
            std::vector<const u32 *> cp(n, nullptr); // the k proof points B_i are prepared on the fly
            std::vector<unsigned char> skip(n, 0);
            std::vector<u64> qs(n * G2L, 0);
            std::memcpy(qs.data(), bs.data(), k * G2L * 8);
            for (u64 i = 0; i < k; ++i) skip[i] = (unsigned char)is_zero_limbs(bs.data() + i * G2L, G2L);
            cp[k] = d_gamma_neg(), cp[k + 1] = d_delta_neg(), cp[k + 2] = d_beta_;
            skip[k + 2] = (unsigned char)is_zero_limbs(beta_.data(), G2L);
            rc = pe_->pairing_product_xyzz((const u32 *)ps.data(), g1_->ec_mul_xyzz_device(wse, k, glv_ok_), k, (void *)wse->stream, cp.data(),
                                           (const u32 *)qs.data(), skip.data(), n, true, out.data());
        }
        hipStreamSynchronize(wse->stream); // (also on the error paths: nothing of this call stays in flight)
        g1_->ws_release(wse);
        g1_->bases_destroy(cb);
        if (rc) return rc;
        std::vector<u32> one(out.size(), 0u);
        HF::one().store_words(one.data());
        *ok = std::memcmp(out.data(), one.data(), out.size() * 4) == 0;
        return MG_OK;
    }

    size_t encoded_size() const override {
        return FB + 3 * 2 * FB + 8 + P_ * FB + 12 * FB + 2 * (8 + (size_t)pe_->n_coeffs() * 6 * FB + 1);
    }
    static void write_fq_words(const u32 *w, uint8_t *out) {
        HF v;
        v.load_words(w);
        v.write_canonical(out);
    }
    int encode(uint8_t *out) const override {
        if (!out) return MG_ERR_ARG;
        HostPoint hp;
        size_t off = 0;
        g1_->hp_from_affine(&hp, (const u32 *)alpha_.data());
        g1_->hp_serialize(&hp, out + off, true);
        off += FB;
        const std::vector<u64> *g2s[3] = {&beta_, &gamma_, &delta_};
        for (int k = 0; k < 3; ++k) {
            g2_->hp_from_affine(&hp, (const u32 *)g2s[k]->data());
            g2_->hp_serialize(&hp, out + off, true);
            off += 2 * FB;
        }
        for (int i = 0; i < 8; ++i) out[off + i] = (uint8_t)(P_ >> (8 * i));
        off += 8;
        for (u64 j = 0; j < P_; ++j) {
            g1_->hp_from_affine(&hp, (const u32 *)(abc_.data() + j * G1L));
            g1_->hp_serialize(&hp, out + off, true);
            off += FB;
        }
        for (int k = 0; k < 12; ++k, off += FB) write_fq_words(alpha_beta_.data() + (size_t)k * 2 * N64, out + off);
        const std::vector<u32> *co[2] = {&gneg_host_, &dneg_host_};
        const u64 nco = (u64)pe_->n_coeffs();
        for (int b = 0; b < 2; ++b) {
            for (int i = 0; i < 8; ++i) out[off + i] = (uint8_t)(nco >> (8 * i));
            off += 8;
            for (size_t k = 0; k < nco * 6; ++k, off += FB) write_fq_words(co[b]->data() + k * 2 * N64, out + off);
            out[off++] = 0;
        }
        return off == encoded_size() ? MG_OK : MG_ERR_STATE;
    }
    int alpha_beta_bytes(uint8_t *out) const override {
        if (!out) return MG_ERR_ARG;
        for (int k = 0; k < 12; ++k) write_fq_words(alpha_beta_.data() + (size_t)k * 2 * N64, out + (size_t)k * FB);
        return MG_OK;
    }

    static int decode_proof(const uint8_t *in, u64 *out) {
        if (!g1_decompress(in, out)) return MG_ERR_ARG;
        if (!g2_decompress(in + FB, out + G1L)) return MG_ERR_ARG;
        if (!g1_decompress(in + 3 * FB, out + G1L + G2L)) return MG_ERR_ARG;
        return MG_OK;
    }
};

} // namespace

int verifier_create(int curve, const u64 *alpha, const u64 *beta, const u64 *gamma, const u64 *delta, const u64 *abc, u64 P,
                    Verifier **out) {
    if (!out) return MG_ERR_ARG;
    int rc;
    if (curve == 0) {
        auto *v = new VerifierT<Bn254, Bn254Pairing>();
        if ((rc = v->init_from_points(curve, alpha, beta, gamma, delta, abc, P))) {
            delete v;
            return rc;
        }
        *out = v;
        return MG_OK;
    }
    if (curve == 1) {
        auto *v = new VerifierT<Bls381, Bls381Pairing>();
        if ((rc = v->init_from_points(curve, alpha, beta, gamma, delta, abc, P))) {
            delete v;
            return rc;
        }
        *out = v;
        return MG_OK;
    }
    return MG_ERR_ARG;
}
int verifier_create_from_bytes(int curve, const uint8_t *bytes, size_t len, Verifier **out) {
    if (!out || !bytes) return MG_ERR_ARG;
    int rc;
    if (curve == 0) {
        auto *v = new VerifierT<Bn254, Bn254Pairing>();
        if ((rc = v->init_from_bytes(curve, bytes, len))) {
            delete v;
            return rc;
        }
        *out = v;
        return MG_OK;
    }
    if (curve == 1) {
        auto *v = new VerifierT<Bls381, Bls381Pairing>();
        if ((rc = v->init_from_bytes(curve, bytes, len))) {
            delete v;
            return rc;
        }
        *out = v;
        return MG_OK;
    }
    return MG_ERR_ARG;
}
int proof_decode(int curve, const uint8_t *bytes, u64 *points_out) {
    if (!bytes || !points_out) return MG_ERR_ARG;
    if (curve == 0) return VerifierT<Bn254, Bn254Pairing>::decode_proof(bytes, points_out);
    if (curve == 1) return VerifierT<Bls381, Bls381Pairing>::decode_proof(bytes, points_out);
    return MG_ERR_ARG;
}

} // namespace mg
