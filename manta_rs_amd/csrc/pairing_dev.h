// Optimal-ate pairing on gfx950 for the Groth16 verifier: the arkworks 0.3 construction --
// `PairingEngine::miller_loop` over prepared G2 line coefficients + `final_exponentiation` (ark-ec 0.3
// models/bn/{mod,g2}.rs and models/bls12/{mod,g2}.rs), reached from `Groth16::verify`
// (manta-crypto/src/arkworks/groth16.rs:603-609 -> ark-groth16 verify_proof_with_prepared_inputs). The reference keeps the
// RESULTS of this code in its committed verifying keys (manta-parameters/data/pay/verifying/*.dat: e(alpha, beta) and the
// 91 line-coefficient triples of -gamma_g2 and -delta_g2), which is what the tests compare the GPU's output with,
// byte for byte.
//
// This header holds the field types, sizes and constants; the pairing itself runs one WAVEFRONT per pairing
// (pairing_coop.h). (The first version ran a whole pairing on one lane, the tower as out-of-line functions over private
// memory: 45.7 ms per verification.)
// Arithmetic is the canonical saturated Montgomery Fp<> of fp_dev.h: values are always fully reduced, so equality tests
// and serialisation need no normalisation.
//
//   Fq2 = Fq[u]/(u^2 + 1),  Fq6 = Fq2[v]/(v^3 - xi),  Fq12 = Fq6[w]/(w^2 - v),  xi = U0 + u  (9 + u BN254, 1 + u BLS12-381)
//   memory order of an Fq12 = arkworks' serialisation order: c0.c0, c0.c1, c0.c2, c1.c0, c1.c1, c1.c2, each c0 then c1.
#pragma once
#include "fp_dev.h"
#include "params_gen.h"

namespace mg {

template <class K> struct Pairing {
    typedef typename K::Fq C;
    typedef Fp<C> F;
    typedef Fp2<C> F2;
    static constexpr int N = C::N;
    static constexpr int F2W = 2 * N, F12W = 12 * N, COEFFW = 3 * F2W;
    // line coefficients per prepared G2 point: one per doubling, one per non-zero loop digit, two Frobenius additions (BN)
    static constexpr int n_coeffs() {
        int n = 0;
        for (int i = K::LOOP_LEN - 2; i >= 0; --i) n += 1 + (K::LOOP[i] != 0);
        return n + (K::BN ? 2 : 0);
    }
    static constexpr int NCOEFF = n_coeffs();
    struct Coeff {
        F2 a, b, c;
    };

    static MG_DEV F fconst(const u32 *w) {
        F r;
#pragma unroll
        for (int i = 0; i < N; ++i) r.v[i] = w[i];
        return r;
    }
    template <class T> static MG_DEV F2 f2const(const T &w) { return F2{fconst(w[0]), fconst(w[1])}; }

    // ---- Fq2
    static MG_DEV F2 add(const F2 &a, const F2 &b) { return F2::add(a, b); }
    static MG_DEV F2 sub(const F2 &a, const F2 &b) { return F2::sub(a, b); }
    static MG_DEV F2 neg(const F2 &a) { return F2::neg(a); }
    static MG_DEV F2 dbl(const F2 &a) { return F2::dbl(a); }
    static MG_DEV F2 conj(const F2 &a) { return F2{a.c0, F::neg(a.c1)}; }
    static MG_DEV F small_mul(const F &a) { // U0 * a for U0 in {1, 9}
        if constexpr (K::U0 == 1) return a;
        const F a2 = F::dbl(a), a4 = F::dbl(a2), a8 = F::dbl(a4);
        static_assert(K::U0 == 1 || K::U0 == 9, "xi = U0 + u with U0 in {1, 9}");
        return F::add(a8, a);
    }
    static MG_DEV void store_coeff(const Coeff &c, u32 *p) {
        c.a.store(p), c.b.store(p + F2W), c.c.store(p + 2 * F2W);
    }
};

} // namespace mg
