// Device-side short-Weierstrass (a = 0) group arithmetic for gfx950, generic over the coordinate
// field F (Fp<Fq> for G1, Fp2<Fq> for G2).
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
// The same formulas serve canonical fields (Fp, Fp2) and the lazily-reduced FpR (fpr_dev.h). For FpR
// the coordinates obey the invariants  X < 8p, Y < 4p, ZZ, ZZZ < 2p, affine x, y < 2p  (every product
// is < 2p); each sub<M>/sub2<M> adds M*p with M >= the subtrahend's bound, each is_zero_mod<B> is told
// the bound B of its operand. The largest product of operand bounds is 100 (P^2 in madd) <= 128.
#pragma once
#include "fp_dev.h"

namespace mg {

// Over Fp2 (G2) the group operations are real functions (like Fp::mul): an XYZZ point over
// Fp2/BLS12-381 is 96 VGPRs, and with everything inlined hipcc (ROCm 7.2) runs out of registers and --
// observed on gfx950, tools/ectest2.hip -- miscompiles the generic add inside the scan kernels. As
// calls, operands travel through scratch (<2 % of the ~40 field multiplications an add costs). Over Fp
// (G1) they stay inline: measured 25 % faster on the BLS12-381 accumulate kernel.
#define MG_EC_CALL __device__ __noinline__

template <class F> struct Affine {
    F x, y;
    MG_DEV bool is_inf() const { return x.is_zero_exact() & y.is_zero_exact(); }
    static MG_DEV Affine load(const u32 *p) { return Affine{F::load(p), F::load(p + F::N)}; }
    MG_DEV void store(u32 *p) const {
        x.store(p);
        y.store(p + F::N);
    }
    static constexpr int WORDS = 2 * F::N;
};

template <class F> struct XYZZ {
    F x, y, zz, zzz;
    static constexpr int WORDS = 4 * F::N;
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

    // 2*(affine) -- mdbl-2008-s-1
    static MG_EC_CALL XYZZ dbl_affine_call(const Affine<F> &p) { return dbl_affine_body(p); }
    static MG_DEV XYZZ dbl_affine(const Affine<F> &p) {
        if constexpr (F::EXT) return dbl_affine_call(p);
        else return dbl_affine_body(p);
    }
    static MG_DEV XYZZ dbl_affine_body(const Affine<F> &p) {
        F U = F::dbl(p.y);                                     // < 4p
        F V = F::sqr(U);
        F W = F::mul(U, V);
        F S = F::mul(p.x, V);
        F X2 = F::sqr(p.x);
        F M = F::add(F::dbl(X2), X2);                          // < 6p
        F X3 = F::template sub2<4>(F::sqr(M), F::zero(), S);   // M^2 - 2S        < 6p
        F Y3 = F::template sub<2>(F::mul(M, F::template sub<8>(S, X3)), F::mul(W, p.y)); // < 4p
        return XYZZ{X3, Y3, V, W};
    }
    // dbl-2008-s-1
    static MG_EC_CALL XYZZ dbl_call(const XYZZ &p) { return dbl_body(p); }
    static MG_DEV XYZZ dbl(const XYZZ &p) {
        if constexpr (F::EXT) return dbl_call(p);
        else return dbl_body(p);
    }
    static MG_DEV XYZZ dbl_body(const XYZZ &p) {
        if (p.is_inf()) return p;
        F U = F::dbl(p.y);                                     // < 8p
        F V = F::sqr(U);
        F W = F::mul(U, V);
        F S = F::mul(p.x, V);
        F X2 = F::sqr(p.x);
        F M = F::add(F::dbl(X2), X2);                          // < 6p
        F X3 = F::template sub2<4>(F::sqr(M), F::zero(), S);   // < 6p
        F Y3 = F::template sub<2>(F::mul(M, F::template sub<8>(S, X3)), F::mul(W, p.y)); // < 4p
        return XYZZ{X3, Y3, F::mul(V, p.zz), F::mul(W, p.zzz)};
    }
    // acc += (neg ? -q : q), q affine -- madd-2008-s with exact exceptional cases
    MG_EC_CALL void madd_call(const Affine<F> &q, bool neg) { madd_body(q, neg); }
    MG_DEV void madd(const Affine<F> &q, bool neg) {
        if constexpr (F::EXT) madd_call(q, neg);
        else madd_body(q, neg);
    }
    MG_DEV void madd_body(const Affine<F> &q_in, bool neg) {
        if (q_in.is_inf()) return;
        Affine<F> q = q_in;
        if (neg) q.y = F::template neg<2>(q.y);                // <= 2p
        if (is_inf()) {
            x = q.x;
            y = q.y;
            zz = F::one();
            zzz = F::one();
            return;
        }
        F U2 = F::mul(q.x, zz);
        F S2 = F::mul(q.y, zzz);
        F P = F::template sub<8>(U2, x);                       // < 10p
        F R = F::template sub<4>(S2, y);                       // < 6p
        if (P.template is_zero_mod<10>()) {
            if (R.template is_zero_mod<6>())
                *this = dbl_affine(q);
            else
                *this = inf();
            return;
        }
        F PP = F::sqr(P);
        F PPP = F::mul(P, PP);
        F Q = F::mul(x, PP);
        F X3 = F::template sub2<6>(F::sqr(R), PPP, Q);         // R^2 - PPP - 2Q   < 8p
        F Y3 = F::template sub<2>(F::mul(R, F::template sub<8>(Q, X3)), F::mul(y, PPP)); // < 4p
        x = X3;
        y = Y3;
        zz = F::mul(zz, PP);
        zzz = F::mul(zzz, PPP);
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
        F U1 = F::mul(x, o.zz);
        F U2 = F::mul(o.x, zz);
        F S1 = F::mul(y, o.zzz);
        F S2 = F::mul(o.y, zzz);
        F P = F::template sub<2>(U2, U1);                      // < 4p
        F R = F::template sub<2>(S2, S1);                      // < 4p
        if (P.template is_zero_mod<4>()) {
            if (R.template is_zero_mod<4>())
                *this = dbl(*this);
            else
                *this = inf();
            return;
        }
        F PP = F::sqr(P);
        F PPP = F::mul(P, PP);
        F Q = F::mul(U1, PP);
        F X3 = F::template sub2<6>(F::sqr(R), PPP, Q);         // < 8p
        F Y3 = F::template sub<2>(F::mul(R, F::template sub<8>(Q, X3)), F::mul(S1, PPP)); // < 4p
        x = X3;
        y = Y3;
        zz = F::mul(F::mul(zz, o.zz), PP);
        zzz = F::mul(F::mul(zzz, o.zzz), PPP);
    }
};

} // namespace mg
