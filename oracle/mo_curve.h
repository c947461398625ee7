/* ORACLE -- TEST INFRASTRUCTURE ONLY (see mo_field.h header).
 *
 * Short-Weierstrass a=0 curve arithmetic in Jacobian coordinates, generic over Fq (G1) and Fq2 (G2).
 * Restates ark-ec ^0.3.0 `models::short_weierstrass_jacobian` (un-vendored; pinned at
 * manta-crypto/Cargo.toml:78; reference call sites manta-crypto/src/arkworks/groth16.rs:33,
 * manta-benchmark/src/ecc.rs:30-128). Formulas: dbl-2009-l, madd-2007-bl, add-2007-bl with exact
 * handling of infinity, P+P and P+(-P).
 *
 * Memory formats shared with the product's C ABI (include/mantagpu.h):
 *   affine point  = x || y, Montgomery form, u64 limbs; infinity = all-zero (never on curve: b != 0)
 *   jacobian      = X || Y || Z; infinity <=> Z == 0
 */
#ifndef MO_CURVE_H
#define MO_CURVE_H
#include "mo_field.h"

typedef struct {
    fld_t K;        /* coordinate field */
    const fp_t *Fr; /* scalar field */
    u64 b[MO_MAXE]; /* curve coefficient (Montgomery) */
    u64 gen[2 * MO_MAXE];
} curve_t;

static inline int c_el(const curve_t *C) { return f_limbs(&C->K); } /* limbs per coordinate */
static inline int jac_is_inf(const curve_t *C, const u64 *P) { return f_is_zero(&C->K, P + 2 * c_el(C)); }
static inline void jac_set_inf(const curve_t *C, u64 *P) {
    const int E = c_el(C);
    memset(P, 0, 8 * 3 * (size_t)E);
    f_set_one(&C->K, P);
    f_set_one(&C->K, P + E);
}
static inline int aff_is_inf(const curve_t *C, const u64 *P) { return limbs_is_zero(P, 2 * c_el(C)); }
static inline void jac_from_affine(const curve_t *C, u64 *J, const u64 *A) {
    const int E = c_el(C);
    if (aff_is_inf(C, A)) {
        jac_set_inf(C, J);
        return;
    }
    memcpy(J, A, 8 * 2 * (size_t)E);
    f_set_one(&C->K, J + 2 * E);
}
static inline void jac_copy(const curve_t *C, u64 *R, const u64 *P) { memmove(R, P, 8 * 3 * (size_t)c_el(C)); }
static inline void jac_neg(const curve_t *C, u64 *R, const u64 *P) {
    const int E = c_el(C);
    jac_copy(C, R, P);
    f_neg(&C->K, R + E, P + E);
}
static inline void aff_neg(const curve_t *C, u64 *R, const u64 *P) {
    const int E = c_el(C);
    memmove(R, P, 8 * 2 * (size_t)E);
    f_neg(&C->K, R + E, P + E);
}

static inline void jac_double(const curve_t *C, u64 *R, const u64 *P) {
    const fld_t *K = &C->K;
    const int E = c_el(C);
    if (jac_is_inf(C, P)) {
        jac_copy(C, R, P);
        return;
    }
    const u64 *X1 = P, *Y1 = P + E, *Z1 = P + 2 * E;
    u64 A[MO_MAXE], B[MO_MAXE], Cc[MO_MAXE], D[MO_MAXE], Ee[MO_MAXE], F[MO_MAXE], t[MO_MAXE], X3[MO_MAXE],
        Y3[MO_MAXE], Z3[MO_MAXE];
    f_sqr(K, A, X1);
    f_sqr(K, B, Y1);
    f_sqr(K, Cc, B);
    f_add(K, t, X1, B);
    f_sqr(K, t, t);
    f_sub(K, t, t, A);
    f_sub(K, t, t, Cc);
    f_dbl(K, D, t);
    f_dbl(K, Ee, A);
    f_add(K, Ee, Ee, A);
    f_sqr(K, F, Ee);
    f_mul(K, Z3, Y1, Z1);
    f_dbl(K, Z3, Z3);
    f_dbl(K, t, D);
    f_sub(K, X3, F, t);
    f_sub(K, t, D, X3);
    f_mul(K, Y3, Ee, t);
    f_dbl(K, t, Cc);
    f_dbl(K, t, t);
    f_dbl(K, t, t);
    f_sub(K, Y3, Y3, t);
    f_copy(K, R, X3);
    f_copy(K, R + E, Y3);
    f_copy(K, R + 2 * E, Z3);
}

/* R = P (jacobian) + Q (affine) */
static inline void jac_add_mixed(const curve_t *C, u64 *R, const u64 *P, const u64 *Q) {
    const fld_t *K = &C->K;
    const int E = c_el(C);
    if (aff_is_inf(C, Q)) {
        jac_copy(C, R, P);
        return;
    }
    if (jac_is_inf(C, P)) {
        jac_from_affine(C, R, Q);
        return;
    }
    const u64 *X1 = P, *Y1 = P + E, *Z1 = P + 2 * E, *X2 = Q, *Y2 = Q + E;
    u64 Z1Z1[MO_MAXE], U2[MO_MAXE], S2[MO_MAXE], H[MO_MAXE], HH[MO_MAXE], I[MO_MAXE], J[MO_MAXE], r[MO_MAXE],
        V[MO_MAXE], t[MO_MAXE], X3[MO_MAXE], Y3[MO_MAXE], Z3[MO_MAXE];
    f_sqr(K, Z1Z1, Z1);
    f_mul(K, U2, X2, Z1Z1);
    f_mul(K, S2, Y2, Z1);
    f_mul(K, S2, S2, Z1Z1);
    f_sub(K, H, U2, X1);
    f_sub(K, r, S2, Y1);
    if (f_is_zero(K, H)) {
        if (f_is_zero(K, r)) {
            jac_double(C, R, P);
        } else {
            jac_set_inf(C, R);
        }
        return;
    }
    f_dbl(K, r, r);
    f_sqr(K, HH, H);
    f_dbl(K, I, HH);
    f_dbl(K, I, I);
    f_mul(K, J, H, I);
    f_mul(K, V, X1, I);
    f_sqr(K, X3, r);
    f_sub(K, X3, X3, J);
    f_sub(K, X3, X3, V);
    f_sub(K, X3, X3, V);
    f_sub(K, t, V, X3);
    f_mul(K, Y3, r, t);
    f_mul(K, t, Y1, J);
    f_dbl(K, t, t);
    f_sub(K, Y3, Y3, t);
    f_add(K, Z3, Z1, H);
    f_sqr(K, Z3, Z3);
    f_sub(K, Z3, Z3, Z1Z1);
    f_sub(K, Z3, Z3, HH);
    f_copy(K, R, X3);
    f_copy(K, R + E, Y3);
    f_copy(K, R + 2 * E, Z3);
}

static inline void jac_add(const curve_t *C, u64 *R, const u64 *P, const u64 *Q) {
    const fld_t *K = &C->K;
    const int E = c_el(C);
    if (jac_is_inf(C, Q)) {
        jac_copy(C, R, P);
        return;
    }
    if (jac_is_inf(C, P)) {
        jac_copy(C, R, Q);
        return;
    }
    const u64 *X1 = P, *Y1 = P + E, *Z1 = P + 2 * E, *X2 = Q, *Y2 = Q + E, *Z2 = Q + 2 * E;
    u64 Z1Z1[MO_MAXE], Z2Z2[MO_MAXE], U1[MO_MAXE], U2[MO_MAXE], S1[MO_MAXE], S2[MO_MAXE], H[MO_MAXE], I[MO_MAXE],
        J[MO_MAXE], r[MO_MAXE], V[MO_MAXE], t[MO_MAXE], X3[MO_MAXE], Y3[MO_MAXE], Z3[MO_MAXE];
    f_sqr(K, Z1Z1, Z1);
    f_sqr(K, Z2Z2, Z2);
    f_mul(K, U1, X1, Z2Z2);
    f_mul(K, U2, X2, Z1Z1);
    f_mul(K, S1, Y1, Z2);
    f_mul(K, S1, S1, Z2Z2);
    f_mul(K, S2, Y2, Z1);
    f_mul(K, S2, S2, Z1Z1);
    f_sub(K, H, U2, U1);
    f_sub(K, r, S2, S1);
    if (f_is_zero(K, H)) {
        if (f_is_zero(K, r)) {
            jac_double(C, R, P);
        } else {
            jac_set_inf(C, R);
        }
        return;
    }
    f_dbl(K, r, r);
    f_dbl(K, I, H);
    f_sqr(K, I, I);
    f_mul(K, J, H, I);
    f_mul(K, V, U1, I);
    f_sqr(K, X3, r);
    f_sub(K, X3, X3, J);
    f_sub(K, X3, X3, V);
    f_sub(K, X3, X3, V);
    f_sub(K, t, V, X3);
    f_mul(K, Y3, r, t);
    f_mul(K, t, S1, J);
    f_dbl(K, t, t);
    f_sub(K, Y3, Y3, t);
    f_add(K, Z3, Z1, Z2);
    f_sqr(K, Z3, Z3);
    f_sub(K, Z3, Z3, Z1Z1);
    f_sub(K, Z3, Z3, Z2Z2);
    f_mul(K, Z3, Z3, H);
    f_copy(K, R, X3);
    f_copy(K, R + E, Y3);
    f_copy(K, R + 2 * E, Z3);
}

static inline void jac_to_affine(const curve_t *C, u64 *A, const u64 *P) {
    const fld_t *K = &C->K;
    const int E = c_el(C);
    if (jac_is_inf(C, P)) {
        memset(A, 0, 8 * 2 * (size_t)E);
        return;
    }
    u64 zi[MO_MAXE], zi2[MO_MAXE], zi3[MO_MAXE];
    f_inv(K, zi, P + 2 * E);
    f_sqr(K, zi2, zi);
    f_mul(K, zi3, zi2, zi);
    f_mul(K, A, P, zi2);
    f_mul(K, A + E, P + E, zi3);
}

/* R = [k]P, k canonical little-endian limbs (nl of them). Plain double-and-add (MSB first). */
static inline void jac_mul(const curve_t *C, u64 *R, const u64 *Paff, const u64 *k, int nl) {
    u64 acc[3 * MO_MAXE];
    jac_set_inf(C, acc);
    int top = limbs_top_bit(k, nl);
    for (int i = top; i >= 0; --i) {
        jac_double(C, acc, acc);
        if (limbs_bit(k, i)) jac_add_mixed(C, acc, acc, Paff);
    }
    jac_copy(C, R, acc);
}
static inline void jac_mul_jac(const curve_t *C, u64 *R, const u64 *P, const u64 *k, int nl) {
    u64 acc[3 * MO_MAXE], base[3 * MO_MAXE];
    jac_copy(C, base, P);
    jac_set_inf(C, acc);
    int top = limbs_top_bit(k, nl);
    for (int i = top; i >= 0; --i) {
        jac_double(C, acc, acc);
        if (limbs_bit(k, i)) jac_add(C, acc, acc, base);
    }
    jac_copy(C, R, acc);
}
static inline int aff_on_curve(const curve_t *C, const u64 *A) {
    const fld_t *K = &C->K;
    const int E = c_el(C);
    if (aff_is_inf(C, A)) return 1;
    u64 l[MO_MAXE], r[MO_MAXE];
    f_sqr(K, l, A + E);
    f_sqr(K, r, A);
    f_mul(K, r, r, A);
    f_add(K, r, r, C->b);
    return f_eq(K, l, r);
}
#endif
