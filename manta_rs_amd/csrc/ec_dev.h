// Device-side short-Weierstrass (a = 0) group arithmetic for gfx950, generic over the coordinate
// field F: canonical Fp<Fq>/Fp2<Fq> (32-bit saturated limbs) or the lazily-reduced FpR<Fq>/Fp2R<Fq>
// (fpr_dev.h) the MSM kernels use.
//
// Replaces (on the GPU) ark-ec ^0.3.0 `short_weierstrass_jacobian` add/double/mixed-add used inside
// VariableBaseMSM (reference call site
// manta-crypto/src/arkworks/groth16.rs:597). Group elements are unique, so the GPU is free to use a
// different coordinate system: bucket accumulators use XYZZ ("extended Jacobian": x = X/ZZ,
// y = Y/ZZZ, ZZ^3 = ZZZ^2), whose mixed add is 8M+2S vs 7M+4S and needs no Z inversion trickery.
// Exceptional cases (infinity, P+P, P+(-P)) are handled exactly: real proving keys contain infinity
// entries and repeated bases (SURVEY.md section 7 "Hard parts").
//
// Memory formats: affine = x || y (Montgomery limbs), infinity = all zero (b != 0 so (0,0) is never
// on the curve); XYZZ = X || Y || ZZ || ZZZ, infinity <=> ZZ == 0.
//
// Lazy reduction is made safe at COMPILE TIME: every intermediate is a Bv<F, B> ("value < B*p"); the
// operators below propagate B, static_assert that each product's operand bounds fit the field's
// Montgomery headroom (F::LIM), that every subtraction adds a large enough multiple of p and that every
// zero test knows how many multiples of p to compare with; b_fit<T>() inserts a cheap reduction only
// where a bound would otherwise exceed T. Stored coordinates obey the invariants X < F::BX p,
// Y < F::BY p, ZZ, ZZZ < F::BM p, affine x, y < F::BM p. For canonical fields all of it folds to nothing.
#pragma once
#include "fp_dev.h"

namespace mg {

// Over Fp2 (G2) the group operations are real functions (like Fp::mul): an XYZZ point over
// Fp2/BLS12-381 is ~100 VGPRs, and with everything inlined hipcc (ROCm 7.2) runs out of registers and --
// observed on gfx950, tools/ectest2.hip -- miscompiles the generic add inside the scan kernels. As
// calls, operands travel through scratch (<2 % of the ~40 field multiplications an add costs). Over Fp
// (G1) they stay inline: measured 25 % faster on the BLS12-381 accumulate kernel.
#define MG_EC_CALL __device__ __noinline__

// ---- compile-time bound tracking -----------------------------------------------------------------
template <class F, int B> struct Bv {
    F v;
};
template <int B, class F> MG_DEV Bv<F, B> bv(const F &f) { return Bv<F, B>{f}; }
template <class F, int A, int B> MG_DEV Bv<F, F::BM> operator*(const Bv<F, A> &a, const Bv<F, B> &b) {
    static_assert((long)A * B * F::MULK <= F::LIM, "product of operand bounds exceeds the Montgomery headroom");
    return Bv<F, F::BM>{F::template mulb<A, B>(a.v, b.v)};
}
template <class F, int A> MG_DEV Bv<F, F::BM> b_sqr(const Bv<F, A> &a) {
    static_assert((long)A * A * F::MULK <= F::LIM, "square of operand bound exceeds the Montgomery headroom");
    return Bv<F, F::BM>{F::template sqrb<A>(a.v)};
}
template <class F, int A, int B> MG_DEV Bv<F, A + B> operator+(const Bv<F, A> &a, const Bv<F, B> &b) {
    return Bv<F, A + B>{F::add(a.v, b.v)};
}
template <class F, int A> MG_DEV Bv<F, 2 * A> b_dbl(const Bv<F, A> &a) { return Bv<F, 2 * A>{F::dbl(a.v)}; }
template <class F, int A, int B> MG_DEV Bv<F, A + B> operator-(const Bv<F, A> &a, const Bv<F, B> &b) {
    static_assert(B <= F::MAXM, "no multiple of p large enough for this subtrahend");
    return Bv<F, A + B>{F::template sub<B>(a.v, b.v)}; // a + B*p - b
}
// a - b - 2c
template <class F, int A, int B, int C>
MG_DEV Bv<F, A + B + 2 * C> b_sub2(const Bv<F, A> &a, const Bv<F, B> &b, const Bv<F, C> &c) {
    static_assert(B + 2 * C <= F::MAXM, "no multiple of p large enough for this subtrahend");
    return Bv<F, A + B + 2 * C>{F::template sub2<B + 2 * C>(a.v, b.v, c.v)};
}
template <class F, int A> MG_DEV Bv<F, A> b_neg(const Bv<F, A> &a) {
    static_assert(A <= F::MAXM, "no multiple of p large enough");
    return Bv<F, A>{F::template neg<A>(a.v)};
}
template <class F, int A> MG_DEV bool b_is_zero(const Bv<F, A> &a) {
    static_assert(A - 1 <= F::MAXM, "multiple table too short for this zero test");
    return a.v.template is_zero_mod<A>();
}
// make the value fit bound T, reducing only if it has to
template <int T, class F, int A> MG_DEV Bv<F, T> b_fit(const Bv<F, A> &a) {
    if constexpr (A <= T) {
        return Bv<F, T>{a.v};
    } else {
        static_assert(F::BRED <= T, "reduction cannot reach the requested bound");
        return Bv<F, T>{F::template reduce<A>(a.v)};
    }
}

template <class F> struct Affine {
    F x, y;
    MG_DEV bool is_inf() const { return x.is_zero_exact() && y.is_zero_exact(); }
    // memory: x || y, each F::AFF_N words (the field's affine-coordinate format: packed 32-bit words for reduced-radix BN254)
    static MG_DEV Affine load(const u32 *p) { return Affine{F::load_aff(p), F::load_aff(p + F::AFF_N)}; }
    MG_DEV void store(u32 *p) const {
        x.store_aff(p);
        y.store_aff(p + F::AFF_N);
    }
    static constexpr int WORDS = 2 * F::AFF_N;
};

template <class F> struct XYZZ {
    F x, y, zz, zzz;
    static constexpr int WORDS = 4 * F::N;
    static constexpr int BX = F::BX, BY = F::BY, BM = F::BM;
    MG_DEV bool is_inf() const { return zz.is_zero_exact(); }
    static MG_DEV XYZZ inf() { return XYZZ{F::zero(), F::zero(), F::zero(), F::zero()}; }
    static MG_DEV XYZZ from_affine(const Affine<F> &a) {
        if (a.is_inf()) return inf();
        return XYZZ{a.x, a.y, F::one(), F::one()};
    }
    static MG_DEV XYZZ load(const u32 *p) {
        return XYZZ{F::load(p), F::load(p + F::N), F::load(p + 2 * F::N), F::load(p + 3 * F::N)};
    }
    MG_DEV void store(u32 *p) const {
        x.store(p);
        y.store(p + F::N);
        zz.store(p + 2 * F::N);
        zzz.store(p + 3 * F::N);
    }
    // arkworks-format store (canonical 32-bit Montgomery limbs) for the few points staged to the host
    MG_DEV void store_std(u32 *p) const {
        typedef typename F::Std S;
        x.to_std().store(p);
        y.to_std().store(p + S::N);
        zz.to_std().store(p + 2 * S::N);
        zzz.to_std().store(p + 3 * S::N);
    }
    static MG_DEV XYZZ shfl(const XYZZ &a, int src) {
        return XYZZ{F::shfl(a.x, src), F::shfl(a.y, src), F::shfl(a.zz, src), F::shfl(a.zzz, src)};
    }
    static MG_DEV XYZZ select(bool c, const XYZZ &a, const XYZZ &b) {
        return XYZZ{F::select(c, a.x, b.x), F::select(c, a.y, b.y), F::select(c, a.zz, b.zz), F::select(c, a.zzz, b.zzz)};
    }

    // shared body of dbl / dbl_affine -- dbl-2008-s-1 / mdbl-2008-s-1: X3, Y3 and the factors V, W
    template <int AX, int AY>
    static MG_DEV void dbl_core(const Bv<F, AX> &px, const Bv<F, AY> &py, F &X3o, F &Y3o, F &Vo, F &Wo) {
        auto U = b_fit<8>(b_dbl(py));
        auto V = b_sqr(U);
        auto W = U * V;
        auto S = px * V;
        auto X2 = b_sqr(px);
        auto M = b_fit<6>(b_dbl(X2) + X2);                         // 3 x^2
        auto X3 = b_fit<BX>(b_sub2(b_sqr(M), bv<0>(F::zero()), S)); // M^2 - 2S
        auto Y3 = b_fit<BY>(M * (S - X3) - W * py);
        X3o = X3.v;
        Y3o = Y3.v;
        Vo = V.v;
        Wo = W.v;
    }
    static MG_EC_CALL XYZZ dbl_affine_call(const Affine<F> &p) { return dbl_affine_body(p); }
    static MG_DEV XYZZ dbl_affine(const Affine<F> &p) {
        if constexpr (F::EXT) return dbl_affine_call(p);
        else return dbl_affine_body(p);
    }
    static MG_DEV XYZZ dbl_affine_body(const Affine<F> &p) {
        XYZZ r;
        dbl_core(bv<BM>(p.x), bv<BM>(p.y), r.x, r.y, r.zz, r.zzz);
        return r;
    }
    static MG_EC_CALL XYZZ dbl_call(const XYZZ &p) { return dbl_body(p); }
    static MG_DEV XYZZ dbl(const XYZZ &p) {
        if constexpr (F::EXT) return dbl_call(p);
        else return dbl_body(p);
    }
    static MG_DEV XYZZ dbl_body(const XYZZ &p) {
        if (p.is_inf()) return p;
        XYZZ r;
        F V, W;
        dbl_core(bv<BX>(p.x), bv<BY>(p.y), r.x, r.y, V, W);
        r.zz = (bv<BM>(V) * bv<BM>(p.zz)).v;
        r.zzz = (bv<BM>(W) * bv<BM>(p.zzz)).v;
        return r;
    }
    // acc += (neg ? -q : q), q affine -- madd-2008-s with exact exceptional cases
    MG_EC_CALL void madd_call(const Affine<F> &q, bool neg) { madd_body(q, neg); }
    MG_DEV void madd(const Affine<F> &q, bool neg) {
        if constexpr (F::EXT) madd_call(q, neg);
        else madd_body(q, neg);
    }
    // the mixed addition for THROUGHPUT-bound callers (>= 2 wavefronts per SIMD: the bucket-accumulate kernel): the
    // single-chain coding of the products (fpr_dev.h mad_chain_*: every multiply-add of a column in ONE dependent chain,
    // several links per asm block). Measured inside the 2^20 BLS12-381 accumulate launch, same box, three alternations:
    // 2.56 ms against 2.63 ms for the compiler's two-chain coding (-3 %; MSM 363 against 355 Mscalar/s). A first version with
    // one asm statement per multiply-add LOST 7 %: the compiler pads every asm statement with an s_nop (~3 400 per loop
    // body). BN254's 9-limb kernel gains the same 3 % (1.55 against 1.60 ms per batched launch). MG_ACC_TWO_CHAINS restores
    // the C coding for re-measurement.
    MG_DEV void madd_throughput(const Affine<F> &q, bool neg) {
        if constexpr (!F::EXT && F::LAZY) {
#ifdef MG_ACC_TWO_CHAINS
            madd_lazy<false>(q, neg);
#else
            madd_lazy<true>(q, neg);
#endif
            return;
        }
        madd(q, neg);
    }
    // madd for the reduced-radix base field (the bucket-accumulate inner loop: ~16 of these per scalar), restructured
    // around two savings the lazy representation allows:
    //   (1) Y3 = R (Q - X3) - Y1 PPP as ONE fused product R*T + Y1*N with N = 3p - PPP: one Montgomery reduction less
    //       (3 K^2 + K instead of 4 K^2 + 2 K multiply-adds for the pair);
    //   (2) the differences P = U2 - X1, R = S2 - Y1, T = Q - X3, N and the negated q.y are taken carry-free (`subl`:
    //       limbs < 3 * 2^LB, no normalisation pass) -- they only ever feed multiplications -- and the P = 0 test moves
    //       to PP = P^2, a normalised product output (two candidates 0, p instead of eleven multiples of p).
    // Bounds (values as multiples of p; inputs X1 < 8, Y1 < 4, ZZ1, ZZZ1, q.x, q.y < 2; F::LIM >= 169):
    //   U2, S2 < 2;  P = U2 + 9p - X1 < 11;  R = S2 + 5p - Y1 < 7;  PP = P^2: 121 <= LIM;  PPP = P PP: 22;  Q = X1 PP: 16;
    //   X3 = R^2 - PPP - 2Q: R^2 49 <= LIM, value < 2 + 6 = 8 (normalised: it is stored);  T = Q + 9p - X3 < 11;
    //   N = 3p - PPP <= 3;  Y3 = R T + Y1 N: 7*11 + 4*3 = 89 <= LIM, value < 2;  ZZ3, ZZZ3 < 2.
    template <bool CH = false> MG_DEV void madd_lazy(const Affine<F> &q_in, bool negate) {
        static_assert(!F::EXT && F::LAZY, "reduced-radix base field only");
        static_assert(F::BX == 8 && F::BY == 4 && F::BM == 2 && F::LIM >= 121, "bound analysis above");
        // LL: the column accumulators have room for lazy limbs (BLS12-381's 14 x 28 bits); otherwise (BN254's 9 x 29
        // bits) the differences are normalised as before and only the fused product and the cheap zero test remain
        constexpr bool LL = F::LAZY_LIMBS;
        if (q_in.is_inf()) return;
        F qy = q_in.y;
        if (negate) {
            if constexpr (LL) qy = F::template negl<3>(q_in.y); // 3p - y with lazy limbs: feeds the product S2 only
            else qy = F::template neg<2>(q_in.y);
        }
        if (is_inf()) {
            x = q_in.x;
            y = negate ? F::template neg<2>(q_in.y) : q_in.y; // stored coordinates are normalised
            zz = F::one();
            zzz = F::one();
            return;
        }
        const F U2 = F::template mul_t<CH>(q_in.x, zz);
        const F S2 = F::template mul_t<CH>(qy, zzz);
        F P, R;
        if constexpr (LL) {
            P = F::template subl<9>(U2, x); // < 11p, lazy limbs
            R = F::template subl<5>(S2, y); // < 7p
        } else {
            P = F::template sub<8>(U2, x); // < 10p
            R = F::template sub<4>(S2, y); // < 6p
        }
        const F PP = F::template sqr_t<CH>(P);
        if (PP.template is_zero_mod<2>()) { // P = 0 (mod p): same x -- doubling or cancellation (rare; exact)
            if ((LL ? F::normalize_u(R) : R).template is_zero_mod<7>()) {
                Affine<F> q{q_in.x, negate ? F::template neg<2>(q_in.y) : q_in.y};
                *this = dbl_affine(q);
            } else {
                *this = inf();
            }
            return;
        }
        const F PPP = F::template mul_t<CH>(P, PP);
        const F Q = F::template mul_t<CH>(x, PP);
        const F X3 = F::template sub2<6>(F::template sqr_t<CH>(R), PPP, Q); // R^2 + 6p - PPP - 2Q < 8p, normalised
        F T, N;
        if constexpr (LL) {
            T = F::template subl<9>(Q, X3); // < 11p
            N = F::template negl<3>(PPP);   // 3p - PPP
        } else {
            T = F::template sub<8>(Q, X3); // < 10p
            N = F::template neg<2>(PPP);   // 2p - PPP
        }
        const F Y3 = F::template mul_add_t<CH>(R, T, y, N);
        zz = F::template mul_t<CH>(zz, PP);
        zzz = F::template mul_t<CH>(zzz, PPP);
        x = X3;
        y = Y3;
    }
    MG_DEV void madd_body(const Affine<F> &q_in, bool negate) {
        if constexpr (!F::EXT && F::LAZY) {
            madd_lazy<false>(q_in, negate);
            return;
        }
        if (q_in.is_inf()) return;
        Affine<F> q = q_in;
        if (negate) q.y = b_neg(bv<BM>(q.y)).v;
        if (is_inf()) {
            x = q.x;
            y = q.y;
            zz = F::one();
            zzz = F::one();
            return;
        }
        const auto X1 = bv<BX>(x);
        const auto Y1 = bv<BY>(y);
        const auto ZZ1 = bv<BM>(zz);
        const auto ZZZ1 = bv<BM>(zzz);
        auto U2 = bv<BM>(q.x) * ZZ1;
        auto S2 = bv<BM>(q.y) * ZZZ1;
        auto P = U2 - X1;
        auto R = S2 - Y1;
        if (b_is_zero(P)) {
            if (b_is_zero(R))
                *this = dbl_affine(q);
            else
                *this = inf();
            return;
        }
        auto PP = b_sqr(P);
        auto PPP = P * PP;
        auto Q = X1 * PP;
        auto X3 = b_fit<BX>(b_sub2(b_sqr(R), PPP, Q)); // R^2 - PPP - 2Q
        auto Y3 = b_fit<BY>(R * (Q - X3) - Y1 * PPP);
        x = X3.v;
        y = Y3.v;
        zz = (ZZ1 * PP).v;
        zzz = (ZZZ1 * PPP).v;
    }
    // acc += o -- add-2008-s with exact exceptional cases
    MG_EC_CALL void add_call(const XYZZ &o) { add_body(o); }
    MG_DEV void add(const XYZZ &o) {
        if constexpr (F::EXT) add_call(o);
        else add_body(o);
    }
    MG_DEV void add_body(const XYZZ &o) {
        if (o.is_inf()) return;
        if (is_inf()) {
            *this = o;
            return;
        }
        auto U1 = bv<BX>(x) * bv<BM>(o.zz);
        auto U2 = bv<BX>(o.x) * bv<BM>(zz);
        auto S1 = bv<BY>(y) * bv<BM>(o.zzz);
        auto S2 = bv<BY>(o.y) * bv<BM>(zzz);
        auto P = U2 - U1;
        auto R = S2 - S1;
        if (b_is_zero(P)) {
            if (b_is_zero(R))
                *this = dbl(*this);
            else
                *this = inf();
            return;
        }
        auto PP = b_sqr(P);
        auto PPP = P * PP;
        auto Q = U1 * PP;
        auto X3 = b_fit<BX>(b_sub2(b_sqr(R), PPP, Q));
        auto Y3 = b_fit<BY>(R * (Q - X3) - S1 * PPP);
        x = X3.v;
        y = Y3.v;
        zz = ((bv<BM>(zz) * bv<BM>(o.zz)) * PP).v;
        zzz = ((bv<BM>(zzz) * bv<BM>(o.zzz)) * PPP).v;
    }
};

// ---- cooperative addition: one group addition spread over the FOUR wavefronts of a 256-thread workgroup --------
// The latency-bound tails of an MSM (bucket reduce, the last merge levels) are chains of dependent additions
// executed by a single wavefront -- ~30 us per step over Fp2. The 14 field products of an addition form four
// dependency levels of at most four independent products (add-2008-s):
//     U1 U2 S1 S2  |  PP RR Z12 Z123  |  PPP Q ZZ3  |  T V ZZZ3
// so four wavefronts that hold IDENTICAL copies of (acc, o) in their registers each compute one product per level
// and swap the results through LDS (one barrier per level). All the linear work and the exceptional cases
// (infinity on either side, P = Q -> doubling, P = -Q) are evaluated redundantly and identically by every wave, so
// the copies stay bit-identical. `lds` = 2 x 4 exchange slots of F::XWORDS words (double-buffered).
template <class F, bool ALLOW_DOUBLE = true> struct CoopAdd {
    static constexpr int SLOT = F::XWORDS;
    // two exchange areas used alternately (one barrier per level) while they fit 48 KB; the largest field (Fp2 over
    // BLS12-381: 8 KB per slot) uses one area and a second barrier per level instead (ALLOW_DOUBLE = false: always one area --
    // callers whose LDS footprint decides their occupancy)
    static constexpr bool DOUBLE = ALLOW_DOUBLE && 2 * 4 * SLOT * 4 <= 49152;
    static constexpr int LDS_WORDS = (DOUBLE ? 2 : 1) * 4 * SLOT;
    // publish this wave's product, fetch all four (16-byte LDS accesses)
    static MG_DEV void swap(u32 *buf, int wave, int lane, const F &mine, F (&all)[4]) {
        mine.store_x(buf + wave * SLOT, lane);
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) all[q] = F::load_x(buf + q * SLOT, lane);
        if (!DOUBLE) __syncthreads();
    }
    static MG_DEV void add(XYZZ<F> &acc, const XYZZ<F> &o, u32 *lds, int wave, int lane) {
        constexpr int BX = F::BX, BY = F::BY, BM = F::BM;
        const bool o_inf = o.is_inf(), a_inf = acc.is_inf();
        F mine, r[4];
        // level 1
        if (wave == 0) mine = (bv<BX>(acc.x) * bv<BM>(o.zz)).v;        // U1
        else if (wave == 1) mine = (bv<BX>(o.x) * bv<BM>(acc.zz)).v;   // U2
        else if (wave == 2) mine = (bv<BY>(acc.y) * bv<BM>(o.zzz)).v;  // S1
        else mine = (bv<BY>(o.y) * bv<BM>(acc.zzz)).v;                 // S2
        swap(lds, wave, lane, mine, r);
        const auto U1 = bv<BM>(r[0]), U2 = bv<BM>(r[1]), S1 = bv<BM>(r[2]), S2 = bv<BM>(r[3]);
        const auto P = U2 - U1;
        const auto R = S2 - S1;
        const bool p0 = b_is_zero(P), r0 = b_is_zero(R);
        // level 2
        if (wave == 0) mine = b_sqr(P).v;                                // PP
        else if (wave == 1) mine = b_sqr(R).v;                           // RR
        else if (wave == 2) mine = (bv<BM>(acc.zz) * bv<BM>(o.zz)).v;    // Z12
        else mine = (bv<BM>(acc.zzz) * bv<BM>(o.zzz)).v;                 // Z123
        swap(lds + (DOUBLE ? 4 * SLOT : 0), wave, lane, mine, r);
        const auto PP = bv<BM>(r[0]), RR = bv<BM>(r[1]), Z12 = bv<BM>(r[2]), Z123 = bv<BM>(r[3]);
        // level 3
        if (wave == 0) mine = (P * PP).v;         // PPP
        else if (wave == 1) mine = (U1 * PP).v;   // Q
        else if (wave == 2) mine = (Z12 * PP).v;  // ZZ3
        else mine = F::zero();
        swap(lds, wave, lane, mine, r);
        const auto PPP = bv<BM>(r[0]), Q = bv<BM>(r[1]);
        const F ZZ3 = r[2];
        const auto X3 = b_fit<BX>(b_sub2(RR, PPP, Q)); // R^2 - PPP - 2Q
        // level 4
        if (wave == 0) mine = (R * (Q - X3)).v;   // T
        else if (wave == 1) mine = (S1 * PPP).v;  // V
        else if (wave == 2) mine = (Z123 * PPP).v; // ZZZ3
        else mine = F::zero();
        swap(lds + (DOUBLE ? 4 * SLOT : 0), wave, lane, mine, r);
        const auto Y3 = b_fit<BY>(bv<BM>(r[0]) - bv<BM>(r[1]));
        XYZZ<F> sum{X3.v, Y3.v, ZZ3, r[2]};
        // exceptional cases, decided identically in every wave
        if (o_inf) {
            sum = acc;
        } else if (a_inf) {
            sum = o;
        } else if (p0) {
            if (r0) sum = XYZZ<F>::dbl(acc);
            else sum = XYZZ<F>::inf();
        }
        acc = sum;
    }
};

} // namespace mg
