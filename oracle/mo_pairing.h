/* ORACLE -- TEST INFRASTRUCTURE ONLY (see mo_field.h header).
 *
 * Textbook optimal-ate pairing for BN254 and BLS12-381, used (a) to pin the oracle against the
 * reference's committed verifying-key fixtures (tests/golden/ *.dat = manta-parameters/data/pay/
 * verifying/ *.dat, layout manta-crypto/src/arkworks/groth16.rs:337-361), which store
 * `alpha_g1_beta_g2 = e(alpha_g1, beta_g2)`, and (b) to run the Groth16 verification equation
 * (`Groth16::verify`, manta-crypto/src/arkworks/groth16.rs:603-609 -> ark-groth16 ^0.3.0
 * `verify_with_processed_vk`) on every proof the oracle or the GPU product emits.
 *
 * Representation: Fq12 = Fq[w]/(w^12 - A w^6 + B) with xi = w^6 = U0 + u:
 *   BN254     : U0 = 9, A = 18, B = 82, D-type twist, untwist (x,y) -> (x w^2, y w^3)
 *   BLS12-381 : U0 = 1, A = 2,  B = 2,  M-type twist, untwist (x,y) -> (x / w^2, y / w^3)
 * Dense schoolbook arithmetic, affine Miller loop, square-and-multiply final exponentiation:
 * slow but transparent (tens of ms per pairing), which is all a checker needs.
 */
#ifndef MO_PAIRING_H
#define MO_PAIRING_H
#include "mo_curve.h"

typedef struct {
    const fp_t *F;
    const curve_t *G1, *G2;
    int u0;                  /* xi = u0 + u */
    u64 cA[MO_MAXL], cB[MO_MAXL], cU0[MO_MAXL]; /* Montgomery small constants */
    int twist_m;             /* 0 = D-type (BN254), 1 = M-type (BLS12-381) */
    int is_bn;
    const u64 *loop;
    int loop_limbs;
    const u64 *final_exp;
    int final_exp_limbs;
    const u64 *final_exp_ark; /* BN254 only: arkworks' fixed multiple of the final exponent */
    int final_exp_ark_limbs;
    u64 frob_x[MO_MAXE], frob_y[MO_MAXE]; /* BN254 twist Frobenius constants */
} pairing_t;

typedef struct {
    u64 c[12][MO_MAXL];
} fq12_t;

static inline void fq12_one(const pairing_t *E, fq12_t *r) {
    memset(r, 0, sizeof(*r));
    fp_set_one(E->F, r->c[0]);
}
static inline void fq12_mul(const pairing_t *E, fq12_t *r, const fq12_t *a, const fq12_t *b) {
    const fp_t *F = E->F;
    u64 t[23][MO_MAXL];
    u64 m[MO_MAXL];
    memset(t, 0, sizeof(t));
    for (int i = 0; i < 12; ++i) {
        if (fp_is_zero(F, a->c[i])) continue;
        for (int j = 0; j < 12; ++j) {
            if (fp_is_zero(F, b->c[j])) continue;
            fp_mul(F, m, a->c[i], b->c[j]);
            fp_add(F, t[i + j], t[i + j], m);
        }
    }
    for (int k = 22; k >= 12; --k) { /* w^k = A w^(k-6) - B w^(k-12) */
        if (fp_is_zero(F, t[k])) continue;
        fp_mul(F, m, t[k], E->cA);
        fp_add(F, t[k - 6], t[k - 6], m);
        fp_mul(F, m, t[k], E->cB);
        fp_sub(F, t[k - 12], t[k - 12], m);
    }
    for (int i = 0; i < 12; ++i) fp_copy(F, r->c[i], t[i]);
}
static inline int fq12_is_one(const pairing_t *E, const fq12_t *a) {
    if (!fp_eq(E->F, a->c[0], E->F->one)) return 0;
    for (int i = 1; i < 12; ++i)
        if (!fp_is_zero(E->F, a->c[i])) return 0;
    return 1;
}
static inline void fq12_pow(const pairing_t *E, fq12_t *r, const fq12_t *a, const u64 *e, int nl) {
    fq12_t acc, base = *a;
    fq12_one(E, &acc);
    int top = limbs_top_bit(e, nl);
    for (int i = top; i >= 0; --i) {
        fq12_mul(E, &acc, &acc, &acc);
        if (limbs_bit(e, i)) fq12_mul(E, &acc, &acc, &base);
    }
    *r = acc;
}
/* add (c0 + c1 u) * w^k into f:  u = w^6 - U0 */
static inline void fq12_add_fq2_at(const pairing_t *E, fq12_t *f, const u64 *c, int k) {
    const fp_t *F = E->F;
    const int n = F->n;
    u64 t[MO_MAXL];
    fp_mul(F, t, c + n, E->cU0);
    fp_sub(F, t, c, t);
    fp_add(F, f->c[k], f->c[k], t);
    fp_add(F, f->c[k + 6], f->c[k + 6], c + n);
}
/* line through T (slope lam) on the twist, evaluated at P=(xP,yP) in G1, as an Fq12 element */
static inline void pairing_line(const pairing_t *E, fq12_t *l, const u64 *lam, const u64 *T, const u64 *P) {
    const fp_t *F = E->F;
    const fld_t *K2 = &E->G2->K;
    const int n = F->n;
    u64 a[MO_MAXE], b[MO_MAXE];
    memset(l, 0, sizeof(*l));
    f_mul_fp(K2, a, lam, P); /* lam * xP */
    f_neg(K2, a, a);
    f_mul(K2, b, lam, T); /* lam*xT - yT */
    f_sub(K2, b, b, T + 2 * n);
    if (!E->twist_m) { /* yP - lam xP w + (lam xT - yT) w^3 */
        fp_copy(F, l->c[0], P + n);
        fq12_add_fq2_at(E, l, a, 1);
        fq12_add_fq2_at(E, l, b, 3);
    } else { /* (yP w^3 - lam xP w^2 + (lam xT - yT)), i.e. the line times w^3 (in a proper subfield) */
        fp_copy(F, l->c[3], P + n);
        fq12_add_fq2_at(E, l, a, 2);
        fq12_add_fq2_at(E, l, b, 0);
    }
}
/* affine twist-point helpers (Fq2 coordinates) */
static inline void tw_double(const pairing_t *E, u64 *lam, u64 *T) {
    const fld_t *K = &E->G2->K;
    const int e = f_limbs(K);
    u64 num[MO_MAXE], den[MO_MAXE], x3[MO_MAXE], y3[MO_MAXE], t[MO_MAXE];
    f_sqr(K, num, T);
    f_dbl(K, t, num);
    f_add(K, num, num, t);
    f_dbl(K, den, T + e);
    f_inv(K, den, den);
    f_mul(K, lam, num, den);
    f_sqr(K, x3, lam);
    f_sub(K, x3, x3, T);
    f_sub(K, x3, x3, T);
    f_sub(K, t, T, x3);
    f_mul(K, y3, lam, t);
    f_sub(K, y3, y3, T + e);
    f_copy(K, T, x3);
    f_copy(K, T + e, y3);
}
static inline void tw_add(const pairing_t *E, u64 *lam, u64 *T, const u64 *Q) {
    const fld_t *K = &E->G2->K;
    const int e = f_limbs(K);
    u64 num[MO_MAXE], den[MO_MAXE], x3[MO_MAXE], y3[MO_MAXE], t[MO_MAXE];
    f_sub(K, num, Q + e, T + e);
    f_sub(K, den, Q, T);
    f_inv(K, den, den);
    f_mul(K, lam, num, den);
    f_sqr(K, x3, lam);
    f_sub(K, x3, x3, T);
    f_sub(K, x3, x3, Q);
    f_sub(K, t, T, x3);
    f_mul(K, y3, lam, t);
    f_sub(K, y3, y3, T + e);
    f_copy(K, T, x3);
    f_copy(K, T + e, y3);
}
/* f *= Miller(P, Q); P affine G1, Q affine G2 (neither infinity) */
static inline void pairing_miller(const pairing_t *E, fq12_t *f_io, const u64 *P, const u64 *Q) {
    const fld_t *K = &E->G2->K;
    const int e = f_limbs(K);
    u64 T[2 * MO_MAXE], Told[2 * MO_MAXE], lam[MO_MAXE];
    fq12_t f, l;
    fq12_one(E, &f);
    memcpy(T, Q, 8 * 2 * (size_t)e);
    int top = limbs_top_bit(E->loop, E->loop_limbs);
    for (int i = top - 1; i >= 0; --i) {
        fq12_mul(E, &f, &f, &f);
        memcpy(Told, T, sizeof(T));
        tw_double(E, lam, T);
        pairing_line(E, &l, lam, Told, P);
        fq12_mul(E, &f, &f, &l);
        if (limbs_bit(E->loop, i)) {
            memcpy(Told, T, sizeof(T));
            tw_add(E, lam, T, Q);
            pairing_line(E, &l, lam, Told, P);
            fq12_mul(E, &f, &f, &l);
        }
    }
    if (E->is_bn) {
        u64 Q1[2 * MO_MAXE], Q2[2 * MO_MAXE];
        f_conj(K, Q1, Q);
        f_mul(K, Q1, Q1, E->frob_x);
        f_conj(K, Q1 + e, Q + e);
        f_mul(K, Q1 + e, Q1 + e, E->frob_y);
        f_conj(K, Q2, Q1);
        f_mul(K, Q2, Q2, E->frob_x);
        f_conj(K, Q2 + e, Q1 + e);
        f_mul(K, Q2 + e, Q2 + e, E->frob_y);
        f_neg(K, Q2 + e, Q2 + e); /* -pi^2(Q) */
        memcpy(Told, T, sizeof(T));
        tw_add(E, lam, T, Q1);
        pairing_line(E, &l, lam, Told, P);
        fq12_mul(E, &f, &f, &l);
        memcpy(Told, T, sizeof(T));
        tw_add(E, lam, T, Q2);
        pairing_line(E, &l, lam, Told, P);
        fq12_mul(E, &f, &f, &l);
    }
    fq12_mul(E, f_io, f_io, &f);
}
/* arkworks tower serialisation order of an Fq12 element (c0{c0,c1,c2}, c1{...}; Fq2 = c0,c1) */
static inline void fq12_to_tower(const pairing_t *E, u64 out[12][MO_MAXL], const fq12_t *p) {
    const fp_t *F = E->F;
    int o = 0;
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 3; ++j) {
            int k = 2 * j + i;
            u64 t[MO_MAXL];
            fp_mul(F, t, p->c[k + 6], E->cU0);
            fp_add(F, out[o], p->c[k], t); /* a = p[k] + U0 p[k+6] */
            fp_copy(F, out[o + 1], p->c[k + 6]);
            o += 2;
        }
}
#endif
