/* ORACLE -- TEST INFRASTRUCTURE ONLY. Never linked into, imported by, or called from the product
 * (manta_rs_amd/). Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
 *
 * CPU restatement of the reference's Groth16 prove hot path:
 *   manta-crypto/src/arkworks/groth16.rs:589-600  Groth16::prove -> ArkGroth16::prove
 * whose arithmetic lives in un-vendored third-party crates (no Cargo.lock in the reference):
 *   ark-groth16 ^0.3.0 (prover.rs create_random_proof/create_proof, r1cs_to_qap.rs witness_map)
 *   ark-ec      ^0.3.0 (msm/variable_base.rs VariableBaseMSM::multi_scalar_mul)
 *   ark-poly    ^0.3.0 (domain/radix2 Radix2EvaluationDomain fft/ifft/coset_*)
 *   ark-ff      ^0.3.0, ark-serialize ^0.3.0, ark-bn254 ^0.3.0, ark-bls12-381 ^0.3.0
 * (pins: manta-crypto/Cargo.toml:76-87). Their published algorithms are restated from SURVEY.md
 * App. A/B; the in-repo statement of the key/QAP conventions is
 * manta-trusted-setup/src/groth16/mpc.rs:251-312,353-431.
 *
 * PINNING (oracle/README.md): pinned against the reference's own committed fixtures -- the three
 * BN254 verifying-key files (decompression of every point, and e(alpha_g1,beta_g2) equals the stored
 * Fq12) -- and against the Groth16 verification equation for every proof. No bit-level golden
 * proof/MSM/NTT vectors exist in the reference; a proof is nevertheless a unique
 * function of (pk, z, r, s), so "verifies + unique" fixes the bytes.
 */
#include "consts_gen.h"
#include "mo_pairing.h"
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define API __attribute__((visibility("default")))

/* ------------------------------------------------------------------ contexts */
static fp_t FR[2], FQ[2];
static curve_t G1c[2], G2c[2];
static pairing_t PE[2];
static u64 FR_GEN[2][MO_MAXL], FR_ROOT[2][MO_MAXL];
static int FR_TWO_ADICITY[2];
static int g_init = 0;

static void fp_setup(fp_t *F, int n, int bits, const u64 *p, const u64 *one, const u64 *r2, const u64 *r3, u64 inv,
                     const u64 *pm2, const u64 *pm1h, const u64 *sq) {
    memset(F, 0, sizeof(*F));
    F->n = n;
    F->bits = bits;
    F->inv = inv;
    memcpy(F->p, p, 8 * n);
    memcpy(F->one, one, 8 * n);
    memcpy(F->r2, r2, 8 * n);
    memcpy(F->r3, r3, 8 * n);
    memcpy(F->pm2, pm2, 8 * n);
    memcpy(F->pm1h, pm1h, 8 * n);
    if (sq) memcpy(F->sqrt_exp, sq, 8 * n);
}
#define FP_SETUP(F, NAME, SQ)                                                                                     \
    do {                                                                                                          \
        static const u64 p_[] = NAME##_P, one_[] = NAME##_R, r2_[] = NAME##_R2, r3_[] = NAME##_R3,               \
                         pm2_[] = NAME##_PM2, pm1h_[] = NAME##_PM1_HALF;                                         \
        fp_setup(F, NAME##_LIMBS, NAME##_BITS, p_, one_, r2_, r3_, NAME##_INV64, pm2_, pm1h_, SQ);               \
    } while (0)

static const u64 bn_ate[] = BN254_ATE_LOOP, bls_ate[] = BLS381_ATE_LOOP;
static const u64 bn_fe[] = BN254_FINAL_EXP, bn_fe_ark[] = BN254_FINAL_EXP_ARK, bls_fe[] = BLS381_FINAL_EXP;

static void cat2(u64 *dst, const u64 *a, const u64 *b, int n) {
    memcpy(dst, a, 8 * n);
    memcpy(dst + n, b, 8 * n);
}

API void mo_init(void) {
    if (g_init) return;
    static const u64 bn_sq[] = BN254_FQ_SQRT_EXP, bls_sq[] = BLS381_FQ_SQRT_EXP;
    FP_SETUP(&FR[0], BN254_FR, NULL);
    FP_SETUP(&FQ[0], BN254_FQ, bn_sq);
    FP_SETUP(&FR[1], BLS381_FR, NULL);
    FP_SETUP(&FQ[1], BLS381_FQ, bls_sq);
    {
        static const u64 g0[] = BN254_FR_GEN_MONT, w0[] = BN254_FR_ROOT_MONT, g1[] = BLS381_FR_GEN_MONT,
                         w1[] = BLS381_FR_ROOT_MONT;
        memcpy(FR_GEN[0], g0, 32);
        memcpy(FR_ROOT[0], w0, 32);
        memcpy(FR_GEN[1], g1, 32);
        memcpy(FR_ROOT[1], w1, 32);
        FR_TWO_ADICITY[0] = BN254_FR_TWO_ADICITY;
        FR_TWO_ADICITY[1] = BLS381_FR_TWO_ADICITY;
    }
    /* BN254 */
    {
        static const u64 b[] = BN254_G1_B_MONT, gx[] = BN254_G1_GX_MONT, gy[] = BN254_G1_GY_MONT;
        static const u64 b0[] = BN254_G2_B_C0_MONT, b1[] = BN254_G2_B_C1_MONT, x0[] = BN254_G2_GX_C0_MONT,
                         x1[] = BN254_G2_GX_C1_MONT, y0[] = BN254_G2_GY_C0_MONT, y1[] = BN254_G2_GY_C1_MONT;
        curve_t *c = &G1c[0];
        memset(c, 0, sizeof(*c));
        c->K.fp = &FQ[0];
        c->K.deg = 1;
        c->Fr = &FR[0];
        memcpy(c->b, b, 32);
        cat2(c->gen, gx, gy, 4);
        c = &G2c[0];
        memset(c, 0, sizeof(*c));
        c->K.fp = &FQ[0];
        c->K.deg = 2;
        c->Fr = &FR[0];
        cat2(c->b, b0, b1, 4);
        cat2(c->gen, x0, x1, 4);
        cat2(c->gen + 8, y0, y1, 4);
    }
    /* BLS12-381 */
    {
        static const u64 b[] = BLS381_G1_B_MONT, gx[] = BLS381_G1_GX_MONT, gy[] = BLS381_G1_GY_MONT;
        static const u64 b0[] = BLS381_G2_B_C0_MONT, b1[] = BLS381_G2_B_C1_MONT, x0[] = BLS381_G2_GX_C0_MONT,
                         x1[] = BLS381_G2_GX_C1_MONT, y0[] = BLS381_G2_GY_C0_MONT, y1[] = BLS381_G2_GY_C1_MONT;
        curve_t *c = &G1c[1];
        memset(c, 0, sizeof(*c));
        c->K.fp = &FQ[1];
        c->K.deg = 1;
        c->Fr = &FR[1];
        memcpy(c->b, b, 48);
        cat2(c->gen, gx, gy, 6);
        c = &G2c[1];
        memset(c, 0, sizeof(*c));
        c->K.fp = &FQ[1];
        c->K.deg = 2;
        c->Fr = &FR[1];
        cat2(c->b, b0, b1, 6);
        cat2(c->gen, x0, x1, 6);
        cat2(c->gen + 12, y0, y1, 6);
    }
    /* pairing engines */
    for (int id = 0; id < 2; ++id) {
        pairing_t *E = &PE[id];
        memset(E, 0, sizeof(*E));
        E->F = &FQ[id];
        E->G1 = &G1c[id];
        E->G2 = &G2c[id];
        E->is_bn = (id == 0);
        E->twist_m = (id == 1);
        E->u0 = id == 0 ? 9 : 1;
        fp_set_u64(E->F, E->cU0, (u64)E->u0);
        fp_set_u64(E->F, E->cA, id == 0 ? 18 : 2);
        fp_set_u64(E->F, E->cB, id == 0 ? 82 : 2);
        if (id == 0) {
            static const u64 fx0[] = BN254_TWIST_FROB_X_C0_MONT, fx1[] = BN254_TWIST_FROB_X_C1_MONT,
                             fy0[] = BN254_TWIST_FROB_Y_C0_MONT, fy1[] = BN254_TWIST_FROB_Y_C1_MONT;
            E->loop = bn_ate;
            E->loop_limbs = 2;
            E->final_exp = bn_fe;
            E->final_exp_limbs = BN254_FINAL_EXP_LIMBS;
            E->final_exp_ark = bn_fe_ark;
            E->final_exp_ark_limbs = BN254_FINAL_EXP_ARK_LIMBS;
            cat2(E->frob_x, fx0, fx1, 4);
            cat2(E->frob_y, fy0, fy1, 4);
        } else {
            E->loop = bls_ate;
            E->loop_limbs = 1;
            E->final_exp = bls_fe;
            E->final_exp_limbs = BLS381_FINAL_EXP_LIMBS;
        }
    }
    g_init = 1;
}

static const curve_t *get_curve(int curve, int group) { return group == 2 ? &G2c[curve] : &G1c[curve]; }

/* ------------------------------------------------------------------ field API (tests) */
/* field ids: 0 = BN254 Fr, 1 = BN254 Fq, 2 = BLS12-381 Fr, 3 = BLS12-381 Fq */
static const fp_t *get_field(int id) { return (id & 1) ? &FQ[id >> 1] : &FR[id >> 1]; }
API int mo_field_limbs(int id) { return get_field(id)->n; }
/* op: 0 add, 1 sub, 2 mul, 3 inv(a), 4 from_canonical(a), 5 to_canonical(a), 6 neg(a), 7 sqr(a) */
API void mo_field_op(int id, int op, const u64 *a, const u64 *b, u64 *out, size_t count) {
    const fp_t *F = get_field(id);
    const int n = F->n;
    for (size_t i = 0; i < count; ++i) {
        const u64 *x = a + i * n, *y = b ? b + i * n : NULL;
        u64 *o = out + i * n;
        switch (op) {
        case 0: fp_add(F, o, x, y); break;
        case 1: fp_sub(F, o, x, y); break;
        case 2: fp_mul(F, o, x, y); break;
        case 3: fp_inv(F, o, x); break;
        case 4: fp_from_canonical(F, o, x); break;
        case 5: fp_to_canonical(F, o, x); break;
        case 6: fp_neg(F, o, x); break;
        case 7: fp_sqr(F, o, x); break;
        }
    }
}

/* ------------------------------------------------------------------ group API (tests) */
API int mo_point_limbs(int curve, int group) { return 2 * c_el(get_curve(curve, group)); }
API void mo_generator(int curve, int group, u64 *out_aff) {
    const curve_t *C = get_curve(curve, group);
    memcpy(out_aff, C->gen, 8 * 2 * (size_t)c_el(C));
}
API int mo_on_curve(int curve, int group, const u64 *aff) { return aff_on_curve(get_curve(curve, group), aff); }
API void mo_g_add(int curve, int group, const u64 *a, const u64 *b, u64 *out) {
    const curve_t *C = get_curve(curve, group);
    u64 J[3 * MO_MAXE];
    jac_from_affine(C, J, a);
    jac_add_mixed(C, J, J, b);
    jac_to_affine(C, out, J);
}
/* out = [k]P, k canonical 4x u64 */
API void mo_g_mul(int curve, int group, const u64 *p, const u64 *k, u64 *out) {
    const curve_t *C = get_curve(curve, group);
    u64 J[3 * MO_MAXE];
    jac_mul(C, J, p, k, 4);
    jac_to_affine(C, out, J);
}

/* fixed-base window table: T[w][d-1] = d * 2^(8w) * B for w<32, d in 1..255 (Jacobian) */
typedef struct {
    const curve_t *C;
    u64 *tab; /* 32*255 jacobian points */
} fbt_t;
static void fbt_build(fbt_t *T, const curve_t *C, const u64 *base_aff) {
    const int E = c_el(C), PJ = 3 * E;
    T->C = C;
    T->tab = (u64 *)malloc(8 * (size_t)PJ * 32 * 255);
    u64 cur[3 * MO_MAXE];
    jac_from_affine(C, cur, base_aff);
    for (int w = 0; w < 32; ++w) {
        u64 *row = T->tab + (size_t)w * 255 * PJ;
        jac_copy(C, row, cur);
        for (int d = 2; d <= 255; ++d) jac_add(C, row + (size_t)(d - 1) * PJ, row + (size_t)(d - 2) * PJ, cur);
        jac_add(C, cur, row + (size_t)254 * PJ, cur); /* 256 * cur */
    }
}
static void fbt_mul(const fbt_t *T, u64 *outJ, const u64 *k /*canonical 4 limbs*/) {
    const curve_t *C = T->C;
    const int PJ = 3 * c_el(C);
    jac_set_inf(C, outJ);
    for (int w = 0; w < 32; ++w) {
        unsigned d = (unsigned)((k[w >> 3] >> ((w & 7) * 8)) & 0xff);
        if (d) jac_add(C, outJ, outJ, T->tab + ((size_t)w * 255 + (d - 1)) * PJ);
    }
}
static void fbt_free(fbt_t *T) { free(T->tab); }

/* batch [k_i]B for one base B (fixed-base): scalars canonical. out affine */
API void mo_fixed_base_mul(int curve, int group, const u64 *base_aff, const u64 *scalars, size_t n, u64 *out_aff) {
    const curve_t *C = get_curve(curve, group);
    const int E = c_el(C);
    fbt_t T;
    fbt_build(&T, C, base_aff);
    for (size_t i = 0; i < n; ++i) {
        u64 J[3 * MO_MAXE];
        fbt_mul(&T, J, scalars + 4 * i);
        jac_to_affine(C, out_aff + (size_t)i * 2 * E, J);
    }
    fbt_free(&T);
}

/* ------------------------------------------------------------------ serialisation (ark-serialize 0.3; App. A.3)
 * Reference call sites: manta-crypto/src/arkworks/groth16.rs:186-195 (proof_as_bytes),
 * :268-303 (ProvingContext codec = *_unchecked / uncompressed). */
static int fp_nbytes(const fp_t *F) { return (F->bits + 7) / 8; }
static void fp_write(const fp_t *F, unsigned char *out, const u64 *a_mont) {
    u64 c[MO_MAXL];
    fp_to_canonical(F, c, a_mont);
    int nb = fp_nbytes(F);
    for (int i = 0; i < nb; ++i) out[i] = (unsigned char)(c[i >> 3] >> ((i & 7) * 8));
}
static int fp_read(const fp_t *F, u64 *a_mont, const unsigned char *in, unsigned char mask_top) {
    u64 c[MO_MAXL] = {0};
    int nb = fp_nbytes(F);
    for (int i = 0; i < nb; ++i) {
        unsigned char b = in[i];
        if (i == nb - 1) b &= (unsigned char)~mask_top;
        c[i >> 3] |= (u64)b << ((i & 7) * 8);
    }
    if (limbs_geq(c, F->p, F->n)) return 0;
    fp_from_canonical(F, a_mont, c);
    return 1;
}
API int mo_point_bytes(int curve, int group, int compressed) {
    const curve_t *C = get_curve(curve, group);
    int nb = fp_nbytes(C->K.fp) * C->K.deg;
    return compressed ? nb : 2 * nb;
}
API void mo_point_serialize(int curve, int group, int compressed, const u64 *aff, unsigned char *out) {
    const curve_t *C = get_curve(curve, group);
    const fp_t *F = C->K.fp;
    const int n = F->n, E = c_el(C), nb = fp_nbytes(F), deg = C->K.deg;
    const int xb = nb * deg;
    if (aff_is_inf(C, aff)) {
        memset(out, 0, compressed ? xb : 2 * xb);
        out[(compressed ? xb : 2 * xb) - 1] |= 0x40;
        return;
    }
    for (int d = 0; d < deg; ++d) fp_write(F, out + d * nb, aff + d * n);
    if (compressed) {
        if (f_is_high(&C->K, aff + E)) out[xb - 1] |= 0x80;
    } else {
        for (int d = 0; d < deg; ++d) fp_write(F, out + xb + d * nb, aff + E + d * n);
    }
}
/* returns 1 on success (point decodes; for compressed form also that x^3+b is a square) */
API int mo_point_deserialize(int curve, int group, int compressed, const unsigned char *in, u64 *aff) {
    const curve_t *C = get_curve(curve, group);
    const fp_t *F = C->K.fp;
    const int n = F->n, E = c_el(C), nb = fp_nbytes(F), deg = C->K.deg;
    const int xb = nb * deg;
    const int total = compressed ? xb : 2 * xb;
    unsigned char flags = in[total - 1];
    if (flags & 0x40) {
        memset(aff, 0, 8 * 2 * (size_t)E);
        return 1;
    }
    if (compressed) {
        for (int d = 0; d < deg; ++d)
            if (!fp_read(F, aff + d * n, in + d * nb, d == deg - 1 ? 0xC0 : 0)) return 0;
        u64 rhs[MO_MAXE], y[MO_MAXE];
        f_sqr(&C->K, rhs, aff);
        f_mul(&C->K, rhs, rhs, aff);
        f_add(&C->K, rhs, rhs, C->b);
        if (!f_sqrt(&C->K, y, rhs)) return 0;
        int want_high = (flags & 0x80) != 0;
        if (f_is_high(&C->K, y) != want_high) f_neg(&C->K, y, y);
        f_copy(&C->K, aff + E, y);
    } else {
        for (int d = 0; d < deg; ++d)
            if (!fp_read(F, aff + d * n, in + d * nb, 0)) return 0;
        for (int d = 0; d < deg; ++d)
            if (!fp_read(F, aff + E + d * n, in + xb + d * nb, d == deg - 1 ? 0xC0 : 0)) return 0;
    }
    return 1;
}

/* ------------------------------------------------------------------ MSM
 * ark-ec 0.3.0 msm/variable_base.rs VariableBaseMSM::multi_scalar_mul, restated from SURVEY.md App. B.2.
 * Called 4x over G1 and 1x over G2 per proof by ark-groth16's create_proof (reached from
 * manta-crypto/src/arkworks/groth16.rs:597). Single-threaded like the reference. */
static int g_threads = 1; /* mo_set_threads: 1 = what the reference ships; > 1 = arkworks-`parallel`-style decomposition */
/* threads for a parallel region with `units` independent pieces of work: never more threads than pieces (a barrier over
 * idle threads is pure cost), never more than the configured count */
static int nt(size_t units) {
    size_t t = (size_t)g_threads;
    if (units < t) t = units;
    return t < 1 ? 1 : (int)t;
}
static unsigned ark_log2(size_t x) { /* ark_std::log2: ceil(log2 x) */
    if (x <= 1) return 0;
    unsigned l = 0;
    size_t v = x - 1;
    while (v) {
        ++l;
        v >>= 1;
    }
    return l;
}
static int scalar_is_one(const u64 *s) { return s[0] == 1 && s[1] == 0 && s[2] == 0 && s[3] == 0; }
static unsigned scalar_window(const u64 *s, unsigned start, unsigned c) { /* (s >> start) mod 2^c */
    unsigned limb = start >> 6, off = start & 63;
    u64 v = limb < 4 ? s[limb] >> off : 0;
    if (off && limb + 1 < 4) v |= s[limb + 1] << (64 - off);
    return (unsigned)(v & ((1ull << c) - 1));
}
static void msm_arkworks(const curve_t *C, const u64 *bases, const u64 *scalars, size_t n, u64 *outJ) {
    const int E = c_el(C), PA = 2 * E, PJ = 3 * E;
    const unsigned c = n < 32 ? 3 : (ark_log2(n) * 69 / 100) + 2;
    const unsigned num_bits = (unsigned)C->Fr->bits;
    const size_t nb = ((size_t)1 << c) - 1;
    unsigned nwin = (num_bits + c - 1) / c;
    u64 *wsums = (u64 *)malloc(8 * (size_t)PJ * nwin);
    /* g_threads > 1: one task per window, exactly the decomposition of arkworks' `parallel` feature
     * (`cfg_into_iter!(window_starts)` in variable_base.rs); g_threads == 1: the loop the reference ships (F3) */
#pragma omp parallel for schedule(dynamic, 1) num_threads(nt(nwin)) if (g_threads > 1)
    for (unsigned wi = 0; wi < nwin; ++wi) {
        const unsigned w_start = wi * c;
        u64 *buckets = (u64 *)malloc(8 * (size_t)PJ * nb);
        u64 res[3 * MO_MAXE];
        jac_set_inf(C, res);
        for (size_t b = 0; b < nb; ++b) jac_set_inf(C, buckets + b * PJ);
        for (size_t i = 0; i < n; ++i) {
            const u64 *s = scalars + 4 * i;
            if (limbs_is_zero(s, 4)) continue;
            if (scalar_is_one(s)) {
                if (w_start == 0) jac_add_mixed(C, res, res, bases + i * PA);
            } else {
                unsigned d = scalar_window(s, w_start, c);
                if (d) jac_add_mixed(C, buckets + (size_t)(d - 1) * PJ, buckets + (size_t)(d - 1) * PJ, bases + i * PA);
            }
        }
        u64 running[3 * MO_MAXE];
        jac_set_inf(C, running);
        for (size_t b = nb; b-- > 0;) {
            jac_add(C, running, running, buckets + b * PJ);
            jac_add(C, res, res, running);
        }
        jac_copy(C, wsums + (size_t)wi * PJ, res);
        free(buckets);
    }
    u64 total[3 * MO_MAXE];
    jac_set_inf(C, total);
    for (unsigned w = nwin; w-- > 1;) {
        jac_add(C, total, total, wsums + (size_t)w * PJ);
        for (unsigned k = 0; k < c; ++k) jac_double(C, total, total);
    }
    jac_add(C, outJ, total, wsums);
    free(wsums);
}
static void msm_naive(const curve_t *C, const u64 *bases, const u64 *scalars, size_t n, u64 *outJ) {
    const int PA = 2 * c_el(C);
    u64 acc[3 * MO_MAXE], t[3 * MO_MAXE];
    jac_set_inf(C, acc);
    for (size_t i = 0; i < n; ++i) {
        jac_mul(C, t, bases + i * PA, scalars + 4 * i, 4);
        jac_add(C, acc, acc, t);
    }
    jac_copy(C, outJ, acc);
}
/* algo 0 = naive double-and-add (truth), 1 = arkworks Pippenger (timed CPU baseline) */
API void mo_msm(int curve, int group, const u64 *bases_aff, const u64 *scalars_canonical, size_t n, int algo,
                u64 *out_aff) {
    const curve_t *C = get_curve(curve, group);
    u64 J[3 * MO_MAXE];
    if (algo == 0)
        msm_naive(C, bases_aff, scalars_canonical, n, J);
    else
        msm_arkworks(C, bases_aff, scalars_canonical, n, J);
    jac_to_affine(C, out_aff, J);
}
/* sum of affine points */
API void mo_g_sum(int curve, int group, const u64 *pts, size_t n, u64 *out_aff) {
    const curve_t *C = get_curve(curve, group);
    const int PA = 2 * c_el(C);
    u64 J[3 * MO_MAXE];
    jac_set_inf(C, J);
    for (size_t i = 0; i < n; ++i) jac_add_mixed(C, J, J, pts + i * PA);
    jac_to_affine(C, out_aff, J);
}

/* ------------------------------------------------------------------ NTT
 * ark-poly 0.3.0 Radix2EvaluationDomain: natural order in and out;
 * omega_D = omega_{2^s}^(2^(s-log D)); ifft scales by D^-1; coset shift g = multiplicative generator. */
static void fr_domain_root(int curve, unsigned log_n, u64 *w) {
    const fp_t *F = &FR[curve];
    fp_copy(F, w, FR_ROOT[curve]);
    for (int i = FR_TWO_ADICITY[curve]; i > (int)log_n; --i) fp_sqr(F, w, w);
}
static void ntt_core(const fp_t *F, u64 *a, unsigned log_n, const u64 *root) {
    const size_t n = (size_t)1 << log_n;
    for (size_t i = 0; i < n; ++i) { /* bit reversal */
        size_t j = 0;
        for (unsigned b = 0; b < log_n; ++b) j |= ((i >> b) & 1) << (log_n - 1 - b);
        if (i < j) {
            u64 t[4];
            memcpy(t, a + 4 * i, 32);
            memcpy(a + 4 * i, a + 4 * j, 32);
            memcpy(a + 4 * j, t, 32);
        }
    }
    for (unsigned s = 1; s <= log_n; ++s) {
        const size_t m = (size_t)1 << s, half = m >> 1;
        u64 wm[4];
        fp_copy(F, wm, root);
        for (unsigned k = s; k < log_n; ++k) fp_sqr(F, wm, wm); /* root^(n/m) */
        if (g_threads > 1 && n >= 4096) { /* chunk-parallel butterflies (ark-poly's parallel FFT splits likewise) */
            const size_t CH = 2048, total = n / 2, nch = (total + CH - 1) / CH;
#pragma omp parallel for schedule(static) num_threads(nt(nch))
            for (size_t ch = 0; ch < nch; ++ch) {
                size_t t0 = ch * CH, t1 = t0 + CH < total ? t0 + CH : total;
                while (t0 < t1) {
                    const size_t blk = t0 / half, j0 = t0 % half;
                    size_t j1 = half;
                    if (blk * half + j1 > t1) j1 = t1 - blk * half;
                    u64 w[4], e[4] = {(u64)j0, 0, 0, 0};
                    fp_pow(F, w, wm, e, 4);
                    const size_t k = blk * m;
                    for (size_t j = j0; j < j1; ++j) {
                        u64 t[4], u[4];
                        fp_mul(F, t, w, a + 4 * (k + j + half));
                        fp_copy(F, u, a + 4 * (k + j));
                        fp_add(F, a + 4 * (k + j), u, t);
                        fp_sub(F, a + 4 * (k + j + half), u, t);
                        fp_mul(F, w, w, wm);
                    }
                    t0 = blk * half + j1;
                }
            }
            continue;
        }
        for (size_t k = 0; k < n; k += m) {
            u64 w[4];
            fp_set_one(F, w);
            for (size_t j = 0; j < half; ++j) {
                u64 t[4], u[4];
                fp_mul(F, t, w, a + 4 * (k + j + half));
                fp_copy(F, u, a + 4 * (k + j));
                fp_add(F, a + 4 * (k + j), u, t);
                fp_sub(F, a + 4 * (k + j + half), u, t);
                fp_mul(F, w, w, wm);
            }
        }
    }
}
static void distribute_powers(const fp_t *F, u64 *a, size_t n, const u64 *g) {
    u64 pw[4];
    fp_set_one(F, pw);
    for (size_t i = 0; i < n; ++i) {
        fp_mul(F, a + 4 * i, a + 4 * i, pw);
        fp_mul(F, pw, pw, g);
    }
}
/* in-place; data Montgomery Fr. inverse: 0 fft / 1 ifft; coset: 0/1 (coset_fft / coset_ifft) */
API int mo_ntt(int curve, u64 *data, unsigned log_n, int inverse, int coset) {
    const fp_t *F = &FR[curve];
    if ((int)log_n > FR_TWO_ADICITY[curve]) return 1;
    const size_t n = (size_t)1 << log_n;
    u64 w[4];
    fr_domain_root(curve, log_n, w);
    if (!inverse) {
        if (coset) distribute_powers(F, data, n, FR_GEN[curve]);
        ntt_core(F, data, log_n, w);
    } else {
        u64 wi[4], ninv[4], nn[4];
        fp_inv(F, wi, w);
        ntt_core(F, data, log_n, wi);
        fp_set_u64(F, nn, (u64)n);
        fp_inv(F, ninv, nn);
        for (size_t i = 0; i < n; ++i) fp_mul(F, data + 4 * i, data + 4 * i, ninv);
        if (coset) {
            u64 gi[4];
            fp_inv(F, gi, FR_GEN[curve]);
            distribute_powers(F, data, n, gi);
        }
    }
    return 0;
}

/* Radix2EvaluationDomain::{fft, ifft} over GROUP elements -- what manta-trusted-setup/src/groth16/mpc.rs:378-381 applies
 * to the powers of tau (`domain.ifft(&batch_into_projective(..))`). pts: 2^log_n affine points, natural order in and out. */
API int mo_group_ntt(int curve, int group, u64 *pts, unsigned log_n, int inverse) {
    const fp_t *F = &FR[curve];
    const curve_t *C = get_curve(curve, group);
    if ((int)log_n > FR_TWO_ADICITY[curve]) return 1;
    const int PA = 2 * c_el(C), PJ = 3 * c_el(C);
    const size_t n = (size_t)1 << log_n;
    u64 w[4], wi[4];
    fr_domain_root(curve, log_n, w);
    if (inverse) {
        fp_inv(F, wi, w);
        fp_copy(F, w, wi);
    }
    u64 *J = (u64 *)malloc(8 * (size_t)PJ * n);
    for (size_t i = 0; i < n; ++i) { /* bit reversal on the way in */
        size_t j = 0;
        for (unsigned b = 0; b < log_n; ++b) j |= ((i >> b) & 1) << (log_n - 1 - b);
        jac_set_inf(C, J + i * PJ);
        jac_add_mixed(C, J + i * PJ, J + i * PJ, pts + j * PA);
    }
    for (unsigned s = 1; s <= log_n; ++s) {
        const size_t m = (size_t)1 << s, half = m >> 1;
        u64 wm[4];
        fp_copy(F, wm, w);
        for (unsigned k = s; k < log_n; ++k) fp_sqr(F, wm, wm);
        for (size_t k = 0; k < n; k += m) {
            u64 tw[4];
            fp_set_one(F, tw);
            for (size_t j = 0; j < half; ++j) {
                u64 twc[4], t[3 * MO_MAXE], u[3 * MO_MAXE], nt[3 * MO_MAXE];
                fp_to_canonical(F, twc, tw);
                jac_mul_jac(C, t, J + (k + j + half) * PJ, twc, 4);
                jac_copy(C, u, J + (k + j) * PJ);
                jac_add(C, J + (k + j) * PJ, u, t);
                jac_neg(C, nt, t);
                jac_add(C, J + (k + j + half) * PJ, u, nt);
                fp_mul(F, tw, tw, wm);
            }
        }
    }
    u64 ninv[4], nn[4], nc[4];
    fp_set_u64(F, nn, (u64)n);
    fp_inv(F, ninv, nn);
    fp_to_canonical(F, nc, ninv);
    for (size_t i = 0; i < n; ++i) {
        if (inverse) {
            u64 t[3 * MO_MAXE];
            jac_mul_jac(C, t, J + i * PJ, nc, 4);
            jac_to_affine(C, pts + i * PA, t);
        } else {
            jac_to_affine(C, pts + i * PA, J + i * PJ);
        }
    }
    free(J);
    return 0;
}

/* ------------------------------------------------------------------ R1CS / QAP witness map
 * ark-groth16 0.3.0 r1cs_to_qap.rs R1CStoQAP::witness_map; conventions
 * mirrored in-repo at manta-trusted-setup/src/groth16/mpc.rs:299-312,367-368. */
typedef struct {
    const uint32_t *row_ptr; /* m+1 */
    const uint32_t *col;     /* nnz */
    const u64 *val;          /* nnz x 4, Montgomery Fr */
} mo_csr;

static void csr_row_dot(const fp_t *F, const mo_csr *M, size_t row, const u64 *z, u64 *out) {
    u64 acc[4] = {0, 0, 0, 0}, t[4];
    for (uint32_t k = M->row_ptr[row]; k < M->row_ptr[row + 1]; ++k) {
        const u64 *coeff = M->val + 4 * (size_t)k;
        if (fp_eq(F, coeff, F->one))
            fp_add(F, acc, acc, z + 4 * (size_t)M->col[k]);
        else {
            fp_mul(F, t, z + 4 * (size_t)M->col[k], coeff);
            fp_add(F, acc, acc, t);
        }
    }
    memcpy(out, acc, 32);
}
static unsigned domain_log(size_t m, size_t P) {
    size_t need = m + P;
    unsigned l = 0;
    while (((size_t)1 << l) < need) ++l;
    return l;
}
/* h_out: D x 4 limbs (coefficients of h, h[D-1] == 0). returns log2(D), or -1 on error */
API int mo_witness_map(int curve, const mo_csr *A, const mo_csr *B, const mo_csr *Cm, size_t m, size_t P,
                       const u64 *z, u64 *h_out) {
    const fp_t *F = &FR[curve];
    const unsigned lg = domain_log(m, P);
    if ((int)lg > FR_TWO_ADICITY[curve]) return -1;
    const size_t D = (size_t)1 << lg;
    u64 *a = (u64 *)calloc(D, 32), *b = (u64 *)calloc(D, 32), *c = (u64 *)calloc(D, 32);
#pragma omp parallel for schedule(static) num_threads(nt(m / 1024)) if (g_threads > 1)
    for (size_t i = 0; i < m; ++i) {
        csr_row_dot(F, A, i, z, a + 4 * i);
        csr_row_dot(F, B, i, z, b + 4 * i);
        csr_row_dot(F, Cm, i, z, c + 4 * i);
    }
    for (size_t j = 0; j < P; ++j) memcpy(a + 4 * (m + j), z + 4 * j, 32);
    mo_ntt(curve, a, lg, 1, 0);
    mo_ntt(curve, b, lg, 1, 0);
    mo_ntt(curve, a, lg, 0, 1);
    mo_ntt(curve, b, lg, 0, 1);
#pragma omp parallel for schedule(static) num_threads(nt(D / 2048)) if (g_threads > 1)
    for (size_t i = 0; i < D; ++i) fp_mul(F, a + 4 * i, a + 4 * i, b + 4 * i);
    mo_ntt(curve, c, lg, 1, 0);
    mo_ntt(curve, c, lg, 0, 1);
    /* (g^D - 1)^-1 */
    u64 gd[4], zi[4];
    fp_copy(F, gd, FR_GEN[curve]);
    for (unsigned k = 0; k < lg; ++k) fp_sqr(F, gd, gd);
    fp_sub(F, gd, gd, F->one);
    fp_inv(F, zi, gd);
#pragma omp parallel for schedule(static) num_threads(nt(D / 2048)) if (g_threads > 1)
    for (size_t i = 0; i < D; ++i) {
        fp_sub(F, a + 4 * i, a + 4 * i, c + 4 * i);
        fp_mul(F, a + 4 * i, a + 4 * i, zi);
    }
    mo_ntt(curve, a, lg, 1, 1);
    memcpy(h_out, a, D * 32);
    free(a);
    free(b);
    free(c);
    return (int)lg;
}

/* ------------------------------------------------------------------ Groth16 keys
 * Field list of ark_groth16::ProvingKey visible at manta-crypto/src/arkworks/groth16.rs:253-264 and
 * manta-trusted-setup/src/groth16/mpc.rs:415-430. All points affine, Montgomery, infinity = zeros. */
typedef struct {
    u64 n_vars, n_inputs, domain, h_len; /* V, P, D, len(h_query) */
    const u64 *alpha_g1, *beta_g1, *delta_g1;
    const u64 *beta_g2, *gamma_g2, *delta_g2;
    const u64 *gamma_abc_g1; /* P */
    const u64 *a_query;      /* V   G1 */
    const u64 *b_g1_query;   /* V   G1 */
    const u64 *b_g2_query;   /* V   G2 */
    const u64 *h_query;      /* h_len G1 */
    const u64 *l_query;      /* V-P G1 */
} mo_pk;

/* Toy trusted setup from explicit toxic waste (tau, alpha, beta, gamma, delta: Montgomery Fr),
 * following ark-groth16 0.3.0 generator.rs generate_parameters / mpc.rs:251-431 conventions with the
 * curve's standard generators. Output buffers are caller-allocated (sizes per mo_pk). */
API int mo_groth16_setup(int curve, const mo_csr *A, const mo_csr *B, const mo_csr *Cm, size_t m, size_t P, size_t V,
                         const u64 *toxic /* 5 x 4 */, u64 *alpha_g1, u64 *beta_g1, u64 *delta_g1, u64 *beta_g2,
                         u64 *gamma_g2, u64 *delta_g2, u64 *gamma_abc_g1, u64 *a_query, u64 *b_g1_query,
                         u64 *b_g2_query, u64 *h_query /* D-1 */, u64 *l_query) {
    const fp_t *F = &FR[curve];
    const curve_t *C1 = &G1c[curve], *C2 = &G2c[curve];
    const int E1 = c_el(C1), E2 = c_el(C2);
    const unsigned lg = domain_log(m, P);
    if ((int)lg > FR_TWO_ADICITY[curve]) return -1;
    const size_t D = (size_t)1 << lg;
    const u64 *tau = toxic, *alpha = toxic + 4, *beta = toxic + 8, *gamma = toxic + 12, *delta = toxic + 16;
    /* Lagrange coefficients L_i(tau) = Z(tau)/D * w^i / (tau - w^i) */
    u64 *L = (u64 *)malloc(D * 32);
    u64 w[4], zt[4], t[4], dn[4], pw[4];
    fr_domain_root(curve, lg, w);
    fp_copy(F, zt, tau);
    for (unsigned k = 0; k < lg; ++k) fp_sqr(F, zt, zt);
    fp_sub(F, zt, zt, F->one); /* Z(tau) = tau^D - 1 */
    fp_set_u64(F, dn, (u64)D);
    fp_inv(F, dn, dn);
    fp_mul(F, dn, dn, zt); /* Z/D */
    fp_set_one(F, pw);
    for (size_t i = 0; i < D; ++i) {
        fp_sub(F, t, tau, pw);
        fp_inv(F, t, t);
        fp_mul(F, t, t, pw);
        fp_mul(F, L + 4 * i, t, dn);
        fp_mul(F, pw, pw, w);
    }
    u64 *a = (u64 *)calloc(V, 32), *b = (u64 *)calloc(V, 32), *c = (u64 *)calloc(V, 32);
    for (size_t i = 0; i < m; ++i) {
        for (uint32_t k = A->row_ptr[i]; k < A->row_ptr[i + 1]; ++k) {
            fp_mul(F, t, A->val + 4 * (size_t)k, L + 4 * i);
            fp_add(F, a + 4 * (size_t)A->col[k], a + 4 * (size_t)A->col[k], t);
        }
        for (uint32_t k = B->row_ptr[i]; k < B->row_ptr[i + 1]; ++k) {
            fp_mul(F, t, B->val + 4 * (size_t)k, L + 4 * i);
            fp_add(F, b + 4 * (size_t)B->col[k], b + 4 * (size_t)B->col[k], t);
        }
        for (uint32_t k = Cm->row_ptr[i]; k < Cm->row_ptr[i + 1]; ++k) {
            fp_mul(F, t, Cm->val + 4 * (size_t)k, L + 4 * i);
            fp_add(F, c + 4 * (size_t)Cm->col[k], c + 4 * (size_t)Cm->col[k], t);
        }
    }
    for (size_t j = 0; j < P; ++j) fp_add(F, a + 4 * j, a + 4 * j, L + 4 * (m + j)); /* input rows */
    fbt_t T1, T2;
    fbt_build(&T1, C1, C1->gen);
    fbt_build(&T2, C2, C2->gen);
    u64 J[3 * MO_MAXE], k[4], ginv[4], dinv[4];
    fp_inv(F, ginv, gamma);
    fp_inv(F, dinv, delta);
#define FB1(dst, sc_mont)                                                                                         \
    do {                                                                                                          \
        fp_to_canonical(F, k, sc_mont);                                                                           \
        fbt_mul(&T1, J, k);                                                                                       \
        jac_to_affine(C1, dst, J);                                                                                \
    } while (0)
#define FB2(dst, sc_mont)                                                                                         \
    do {                                                                                                          \
        fp_to_canonical(F, k, sc_mont);                                                                           \
        fbt_mul(&T2, J, k);                                                                                       \
        jac_to_affine(C2, dst, J);                                                                                \
    } while (0)
    FB1(alpha_g1, alpha);
    FB1(beta_g1, beta);
    FB1(delta_g1, delta);
    FB2(beta_g2, beta);
    FB2(gamma_g2, gamma);
    FB2(delta_g2, delta);
    for (size_t j = 0; j < V; ++j) {
        FB1(a_query + j * 2 * E1, a + 4 * j);
        FB1(b_g1_query + j * 2 * E1, b + 4 * j);
        FB2(b_g2_query + j * 2 * E2, b + 4 * j);
        u64 e[4], u[4];
        fp_mul(F, e, beta, a + 4 * j);
        fp_mul(F, u, alpha, b + 4 * j);
        fp_add(F, e, e, u);
        fp_add(F, e, e, c + 4 * j);
        if (j < P) {
            fp_mul(F, e, e, ginv);
            FB1(gamma_abc_g1 + j * 2 * E1, e);
        } else {
            fp_mul(F, e, e, dinv);
            FB1(l_query + (j - P) * 2 * E1, e);
        }
    }
    /* h_query[i] = tau^i * Z(tau)/delta, i < D-1 */
    u64 hz[4];
    fp_mul(F, hz, zt, dinv);
    for (size_t i = 0; i + 1 < D; ++i) {
        FB1(h_query + i * 2 * E1, hz);
        fp_mul(F, hz, hz, tau);
    }
    fbt_free(&T1);
    fbt_free(&T2);
    free(L);
    free(a);
    free(b);
    free(c);
    return (int)lg;
}

/* ark-groth16 0.3.0 prover.rs create_proof, reached from
 * manta-crypto/src/arkworks/groth16.rs:597. z = instance || witness (Montgomery), r, s Montgomery.
 * proof_out: a || b || c compressed canonical bytes (groth16.rs:186-195). msm_algo as mo_msm. */
API int mo_groth16_prove(int curve, const mo_pk *pk, const mo_csr *A, const mo_csr *B, const mo_csr *Cm, size_t m,
                         const u64 *z, const u64 *r, const u64 *s, int msm_algo, unsigned char *proof_out,
                         u64 *h_out_opt) {
    const fp_t *F = &FR[curve];
    const curve_t *C1 = &G1c[curve], *C2 = &G2c[curve];
    const int PA1 = 2 * c_el(C1), PA2 = 2 * c_el(C2);
    const size_t V = pk->n_vars, P = pk->n_inputs;
    const unsigned lg = domain_log(m, P);
    const size_t D = (size_t)1 << lg;
    u64 *h = (u64 *)malloc(D * 32);
    if (mo_witness_map(curve, A, B, Cm, m, P, z, h) < 0) {
        free(h);
        return -1;
    }
    if (h_out_opt) memcpy(h_out_opt, h, D * 32);
    /* into_repr */
    u64 *zc = (u64 *)malloc(V * 32), *hc = (u64 *)malloc(D * 32);
#pragma omp parallel for schedule(static) num_threads(nt(V / 2048)) if (g_threads > 1)
    for (size_t i = 0; i < V; ++i) fp_to_canonical(F, zc + 4 * i, z + 4 * i);
#pragma omp parallel for schedule(static) num_threads(nt(D / 2048)) if (g_threads > 1)
    for (size_t i = 0; i < D; ++i) fp_to_canonical(F, hc + 4 * i, h + 4 * i);
    void (*msm)(const curve_t *, const u64 *, const u64 *, size_t, u64 *) = msm_algo ? msm_arkworks : msm_naive;
    u64 h_acc[3 * MO_MAXE], l_acc[3 * MO_MAXE], g_a[3 * MO_MAXE], g1_b[3 * MO_MAXE], g2_b[3 * MO_MAXE],
        g_c[3 * MO_MAXE], t[3 * MO_MAXE];
    size_t hl = pk->h_len < D ? pk->h_len : D;
    msm(C1, pk->h_query, hc, hl, h_acc);
    msm(C1, pk->l_query, zc + 4 * P, V - P, l_acc);
    u64 rc[4], sc[4], rs[4], rsc[4];
    fp_to_canonical(F, rc, r);
    fp_to_canonical(F, sc, s);
    fp_mul(F, rs, r, s);
    fp_to_canonical(F, rsc, rs);
    /* g_a = r*delta_g1 + a_query[0] + MSM(a_query[1..], z[1..]) + alpha_g1 */
    msm(C1, pk->a_query + PA1, zc + 4, V - 1, g_a);
    jac_add_mixed(C1, g_a, g_a, pk->a_query);
    jac_mul(C1, t, pk->delta_g1, rc, 4);
    jac_add(C1, g_a, g_a, t);
    jac_add_mixed(C1, g_a, g_a, pk->alpha_g1);
    /* g1_b (only if r != 0) */
    jac_set_inf(C1, g1_b);
    if (!fp_is_zero(F, r)) {
        msm(C1, pk->b_g1_query + PA1, zc + 4, V - 1, g1_b);
        jac_add_mixed(C1, g1_b, g1_b, pk->b_g1_query);
        jac_mul(C1, t, pk->delta_g1, sc, 4);
        jac_add(C1, g1_b, g1_b, t);
        jac_add_mixed(C1, g1_b, g1_b, pk->beta_g1);
    }
    /* g2_b */
    msm(C2, pk->b_g2_query + PA2, zc + 4, V - 1, g2_b);
    jac_add_mixed(C2, g2_b, g2_b, pk->b_g2_query);
    jac_mul(C2, t, pk->delta_g2, sc, 4);
    jac_add(C2, g2_b, g2_b, t);
    jac_add_mixed(C2, g2_b, g2_b, pk->beta_g2);
    /* g_c = s*g_a + r*g1_b - rs*delta_g1 + l_acc + h_acc */
    jac_mul_jac(C1, g_c, g_a, sc, 4);
    jac_mul_jac(C1, t, g1_b, rc, 4);
    jac_add(C1, g_c, g_c, t);
    jac_mul(C1, t, pk->delta_g1, rsc, 4);
    jac_neg(C1, t, t);
    jac_add(C1, g_c, g_c, t);
    jac_add(C1, g_c, g_c, l_acc);
    jac_add(C1, g_c, g_c, h_acc);
    u64 aa[2 * MO_MAXE], ba[2 * MO_MAXE], ca[2 * MO_MAXE];
    jac_to_affine(C1, aa, g_a);
    jac_to_affine(C2, ba, g2_b);
    jac_to_affine(C1, ca, g_c);
    int b1 = mo_point_bytes(curve, 1, 1), b2 = mo_point_bytes(curve, 2, 1);
    mo_point_serialize(curve, 1, 1, aa, proof_out);
    mo_point_serialize(curve, 2, 1, ba, proof_out + b1);
    mo_point_serialize(curve, 1, 1, ca, proof_out + b1 + b2);
    free(h);
    free(zc);
    free(hc);
    return 0;
}

/* ------------------------------------------------------------------ pairing / verification */
static void pairing_full(const pairing_t *E, fq12_t *out, const u64 *P, const u64 *Q, int ark_exp) {
    fq12_t f;
    fq12_one(E, &f);
    pairing_miller(E, &f, P, Q);
    if (ark_exp && E->final_exp_ark)
        fq12_pow(E, out, &f, E->final_exp_ark, E->final_exp_ark_limbs);
    else
        fq12_pow(E, out, &f, E->final_exp, E->final_exp_limbs);
}
/* out_bytes: 12 Fq elements, arkworks tower order, canonical LE */
API void mo_pairing_bytes(int curve, const u64 *P_g1, const u64 *Q_g2, int ark_exp, unsigned char *out_bytes) {
    const pairing_t *E = &PE[curve];
    fq12_t f;
    pairing_full(E, &f, P_g1, Q_g2, ark_exp);
    u64 tw[12][MO_MAXL];
    fq12_to_tower(E, tw, &f);
    int nb = fp_nbytes(E->F);
    for (int i = 0; i < 12; ++i) fp_write(E->F, out_bytes + i * nb, tw[i]);
}
/* textbook pairing f^((q^12-1)/r) raised to a further small power `mult` (arkworks' final exponentiations compute a
 * fixed multiple of the textbook exponent: BN Fuentes-Castaneda, BLS12 Hayashida et al.) */
API void mo_pairing_bytes_pow(int curve, const u64 *P_g1, const u64 *Q_g2, u64 mult, unsigned char *out_bytes) {
    const pairing_t *E = &PE[curve];
    fq12_t f, g;
    pairing_full(E, &f, P_g1, Q_g2, 0);
    u64 e[1] = {mult};
    fq12_pow(E, &g, &f, e, 1);
    u64 tw[12][MO_MAXL];
    fq12_to_tower(E, tw, &g);
    int nb = fp_nbytes(E->F);
    for (int i = 0; i < 12; ++i) fp_write(E->F, out_bytes + i * nb, tw[i]);
}
/* prod e(P_i, Q_i) == 1 ? (points with an infinity member are skipped) */
API int mo_pairing_product_is_one(int curve, const u64 *Ps, const u64 *Qs, size_t n) {
    const pairing_t *E = &PE[curve];
    const int PA1 = 2 * c_el(E->G1), PA2 = 2 * c_el(E->G2);
    fq12_t f, g;
    fq12_one(E, &f);
    for (size_t i = 0; i < n; ++i) {
        if (aff_is_inf(E->G1, Ps + i * PA1) || aff_is_inf(E->G2, Qs + i * PA2)) continue;
        pairing_miller(E, &f, Ps + i * PA1, Qs + i * PA2);
    }
    fq12_pow(E, &g, &f, E->final_exp, E->final_exp_limbs);
    return fq12_is_one(E, &g);
}
/* Groth16 verification equation: e(A,B) = e(alpha,beta) e(sum x_i gamma_abc_i, gamma) e(C,delta)
 * (ark-groth16 0.3.0 verifier.rs; reference call site manta-crypto/src/arkworks/groth16.rs:603-609).
 * inputs: P-1 public inputs (Montgomery Fr; the leading 1 is implicit). returns 1 valid / 0 invalid / -1 malformed */
API int mo_groth16_verify(int curve, const mo_pk *pk, const u64 *inputs, const unsigned char *proof) {
    const fp_t *F = &FR[curve];
    const curve_t *C1 = &G1c[curve], *C2 = &G2c[curve];
    const int PA1 = 2 * c_el(C1), PA2 = 2 * c_el(C2);
    int b1 = mo_point_bytes(curve, 1, 1), b2 = mo_point_bytes(curve, 2, 1);
    u64 A[2 * MO_MAXE], Bp[2 * MO_MAXE], Cp[2 * MO_MAXE];
    if (!mo_point_deserialize(curve, 1, 1, proof, A)) return -1;
    if (!mo_point_deserialize(curve, 2, 1, proof + b1, Bp)) return -1;
    if (!mo_point_deserialize(curve, 1, 1, proof + b1 + b2, Cp)) return -1;
    if (!aff_on_curve(C1, A) || !aff_on_curve(C2, Bp) || !aff_on_curve(C1, Cp)) return -1;
    u64 acc[3 * MO_MAXE], t[3 * MO_MAXE], k[4];
    jac_from_affine(C1, acc, pk->gamma_abc_g1);
    for (size_t j = 1; j < pk->n_inputs; ++j) {
        fp_to_canonical(F, k, inputs + 4 * (j - 1));
        jac_mul(C1, t, pk->gamma_abc_g1 + j * PA1, k, 4);
        jac_add(C1, acc, acc, t);
    }
    u64 Ps[4 * 2 * MO_MAXL], Qs[4 * 2 * MO_MAXE];
    aff_neg(C1, Ps, A);
    memcpy(Qs, Bp, 8 * PA2);
    memcpy(Ps + PA1, pk->alpha_g1, 8 * PA1);
    memcpy(Qs + PA2, pk->beta_g2, 8 * PA2);
    jac_to_affine(C1, Ps + 2 * PA1, acc);
    memcpy(Qs + 2 * PA2, pk->gamma_g2, 8 * PA2);
    memcpy(Ps + 3 * PA1, Cp, 8 * PA1);
    memcpy(Qs + 3 * PA2, pk->delta_g2, 8 * PA2);
    return mo_pairing_product_is_one(curve, Ps, Qs, 4);
}

/* ------------------------------------------------------------------ CPU baseline timers (bench.py cpu_baseline leg) */
static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}
/* threads for the timed baselines: 1 (default) = the single-threaded arkworks the reference ships;
 * n > 1 = the decomposition of arkworks' `parallel` feature (one task per MSM window, chunk-parallel FFT stages and
 * element loops) on n threads; 0 = all hardware threads. Returns the thread count in effect. */
API int mo_set_threads(int n) {
#ifdef _OPENMP
    if (n <= 0) n = omp_get_num_procs();
    g_threads = n < 1 ? 1 : n;
#else
    (void)n;
    g_threads = 1;
#endif
    return g_threads;
}
API int mo_hardware_threads(void) {
#ifdef _OPENMP
    return omp_get_num_procs();
#else
    return 1;
#endif
}
API double mo_time_msm(int curve, int group, const u64 *bases_aff, const u64 *scalars_canonical, size_t n,
                       u64 *out_aff) {
    double t0 = now_s();
    mo_msm(curve, group, bases_aff, scalars_canonical, n, 1, out_aff);
    return now_s() - t0;
}
