/* ORACLE -- TEST INFRASTRUCTURE ONLY. Never linked into, imported by, or called from the product
 * (manta_rs_amd/). Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
 *
 * CPU restatement of the prime-field arithmetic underneath the reference's Groth16 prover.
 * The reference's arithmetic lives in the un-vendored crate ark-ff ^0.3.0 (Fp256/Fp384 Montgomery
 * backend; pinned at manta-crypto/Cargo.toml:81, call sites manta-crypto/src/arkworks/ff.rs:26-72,
 * manta-crypto/src/arkworks/groth16.rs:33-34). Conventions restated:
 * little-endian u64 limbs, Montgomery form with R = 2^(64*limbs), values fully reduced to [0,p).
 *
 * Pinning: "parity pinned by fixtures" -- see oracle/README.md: the committed BN254 verifying-key
 * files (tests/golden/ *.dat, from manta-parameters/data/pay/verifying/) exercise Fq, sqrt, Fq2,
 * point (de)compression and the pairing end to end; Poseidon KATs pin Fr(BLS12-381).
 */
#ifndef MO_FIELD_H
#define MO_FIELD_H
#include <stdint.h>
#include <string.h>

typedef uint64_t u64;
typedef unsigned __int128 u128;

#define MO_MAXL 6 /* 384-bit */

typedef struct {
    int n;    /* 64-bit limbs */
    int bits; /* modulus bits */
    u64 p[MO_MAXL], one[MO_MAXL], r2[MO_MAXL], r3[MO_MAXL], pm2[MO_MAXL], pm1h[MO_MAXL], sqrt_exp[MO_MAXL];
    u64 inv; /* -p^-1 mod 2^64 */
} fp_t;

static inline int limbs_geq(const u64 *a, const u64 *b, int n) {
    for (int i = n - 1; i >= 0; --i) {
        if (a[i] > b[i]) return 1;
        if (a[i] < b[i]) return 0;
    }
    return 1;
}
static inline int limbs_is_zero(const u64 *a, int n) {
    u64 x = 0;
    for (int i = 0; i < n; ++i) x |= a[i];
    return x == 0;
}
static inline int limbs_eq(const u64 *a, const u64 *b, int n) { return memcmp(a, b, 8 * (size_t)n) == 0; }
static inline u64 limbs_sub(u64 *r, const u64 *a, const u64 *b, int n) {
    u64 borrow = 0;
    for (int i = 0; i < n; ++i) {
        u128 d = (u128)a[i] - b[i] - borrow;
        r[i] = (u64)d;
        borrow = (u64)(d >> 64) & 1;
    }
    return borrow;
}
static inline u64 limbs_add(u64 *r, const u64 *a, const u64 *b, int n) {
    u64 c = 0;
    for (int i = 0; i < n; ++i) {
        u128 s = (u128)a[i] + b[i] + c;
        r[i] = (u64)s;
        c = (u64)(s >> 64);
    }
    return c;
}
static inline int limbs_bit(const u64 *a, int i) { return (int)((a[i >> 6] >> (i & 63)) & 1); }
static inline int limbs_top_bit(const u64 *a, int n) {
    for (int i = 64 * n - 1; i >= 0; --i)
        if (limbs_bit(a, i)) return i;
    return -1;
}

/* ---- Montgomery multiplication (CIOS), unrolled per limb count by constant propagation ---- */
static inline __attribute__((always_inline)) void mo_mont_mul_k(const u64 *p, u64 inv, u64 *r, const u64 *a,
                                                                 const u64 *b, const int n) {
    u64 t[MO_MAXL + 2];
    for (int i = 0; i < n + 2; ++i) t[i] = 0;
    for (int i = 0; i < n; ++i) {
        u128 c = 0;
        for (int j = 0; j < n; ++j) {
            c += (u128)a[j] * b[i] + t[j];
            t[j] = (u64)c;
            c >>= 64;
        }
        c += t[n];
        t[n] = (u64)c;
        t[n + 1] = (u64)(c >> 64);
        u64 m = t[0] * inv;
        c = (u128)m * p[0] + t[0];
        c >>= 64;
        for (int j = 1; j < n; ++j) {
            c += (u128)m * p[j] + t[j];
            t[j - 1] = (u64)c;
            c >>= 64;
        }
        c += t[n];
        t[n - 1] = (u64)c;
        t[n] = t[n + 1] + (u64)(c >> 64);
    }
    if (t[n] || limbs_geq(t, p, n)) limbs_sub(t, t, p, n);
    for (int i = 0; i < n; ++i) r[i] = t[i];
}

static inline void fp_mul(const fp_t *F, u64 *r, const u64 *a, const u64 *b) {
    if (F->n == 4)
        mo_mont_mul_k(F->p, F->inv, r, a, b, 4);
    else
        mo_mont_mul_k(F->p, F->inv, r, a, b, 6);
}
static inline void fp_sqr(const fp_t *F, u64 *r, const u64 *a) { fp_mul(F, r, a, a); }
static inline void fp_add(const fp_t *F, u64 *r, const u64 *a, const u64 *b) {
    u64 t[MO_MAXL];
    limbs_add(t, a, b, F->n); /* 2p < 2^(64n): no carry out */
    if (limbs_geq(t, F->p, F->n)) limbs_sub(t, t, F->p, F->n);
    memcpy(r, t, 8 * (size_t)F->n);
}
static inline void fp_sub(const fp_t *F, u64 *r, const u64 *a, const u64 *b) {
    u64 t[MO_MAXL];
    if (limbs_sub(t, a, b, F->n)) limbs_add(t, t, F->p, F->n);
    memcpy(r, t, 8 * (size_t)F->n);
}
static inline void fp_neg(const fp_t *F, u64 *r, const u64 *a) {
    if (limbs_is_zero(a, F->n))
        memset(r, 0, 8 * (size_t)F->n);
    else
        limbs_sub(r, F->p, a, F->n);
}
static inline void fp_copy(const fp_t *F, u64 *r, const u64 *a) { memmove(r, a, 8 * (size_t)F->n); }
static inline void fp_zero(const fp_t *F, u64 *r) { memset(r, 0, 8 * (size_t)F->n); }
static inline void fp_set_one(const fp_t *F, u64 *r) { memcpy(r, F->one, 8 * (size_t)F->n); }
static inline int fp_is_zero(const fp_t *F, const u64 *a) { return limbs_is_zero(a, F->n); }
static inline int fp_eq(const fp_t *F, const u64 *a, const u64 *b) { return limbs_eq(a, b, F->n); }
/* canonical integer -> Montgomery (ark-ff `from_repr`): a * R2 * R^-1 */
static inline void fp_from_canonical(const fp_t *F, u64 *r, const u64 *a) { fp_mul(F, r, a, F->r2); }
/* Montgomery -> canonical integer (ark-ff `into_repr`): a * 1 * R^-1 */
static inline void fp_to_canonical(const fp_t *F, u64 *r, const u64 *a) {
    u64 o[MO_MAXL] = {1, 0, 0, 0, 0, 0};
    fp_mul(F, r, a, o);
}
static inline void fp_set_u64(const fp_t *F, u64 *r, u64 v) {
    u64 t[MO_MAXL] = {0};
    t[0] = v;
    fp_from_canonical(F, r, t);
}
/* r = a^e, e = nl little-endian limbs (plain integer) */
static inline void fp_pow(const fp_t *F, u64 *r, const u64 *a, const u64 *e, int nl) {
    u64 acc[MO_MAXL], base[MO_MAXL];
    fp_set_one(F, acc);
    fp_copy(F, base, a);
    int top = limbs_top_bit(e, nl);
    for (int i = top; i >= 0; --i) {
        fp_sqr(F, acc, acc);
        if (limbs_bit(e, i)) fp_mul(F, acc, acc, base);
    }
    fp_copy(F, r, acc);
}
static inline void fp_inv(const fp_t *F, u64 *r, const u64 *a) { fp_pow(F, r, a, F->pm2, F->n); }
/* sqrt for p = 3 mod 4; returns 1 iff a is a square */
static inline int fp_sqrt(const fp_t *F, u64 *r, const u64 *a) {
    u64 s[MO_MAXL], c[MO_MAXL];
    fp_pow(F, s, a, F->sqrt_exp, F->n);
    fp_sqr(F, c, s);
    if (!fp_eq(F, c, a)) return 0;
    fp_copy(F, r, s);
    return 1;
}
/* "lexicographically largest": canonical(a) > (p-1)/2 */
static inline int fp_is_high(const fp_t *F, const u64 *a) {
    u64 c[MO_MAXL];
    fp_to_canonical(F, c, a);
    return !limbs_geq(F->pm1h, c, F->n);
}

/* ------------------------------------------------------------------------------------------
 * fld_t: either Fp (deg 1) or Fp2 = Fp[u]/(u^2+1) (deg 2; both BN254 and BLS12-381 use the
 * non-residue -1). Elements are deg*n limbs, c0 then c1.
 * ------------------------------------------------------------------------------------------ */
#define MO_MAXE (2 * MO_MAXL)
typedef struct {
    const fp_t *fp;
    int deg;
} fld_t;
static inline int f_limbs(const fld_t *K) { return K->deg * K->fp->n; }
static inline void f_copy(const fld_t *K, u64 *r, const u64 *a) { memmove(r, a, 8 * (size_t)f_limbs(K)); }
static inline void f_zero(const fld_t *K, u64 *r) { memset(r, 0, 8 * (size_t)f_limbs(K)); }
static inline void f_set_one(const fld_t *K, u64 *r) {
    f_zero(K, r);
    fp_set_one(K->fp, r);
}
static inline int f_is_zero(const fld_t *K, const u64 *a) { return limbs_is_zero(a, f_limbs(K)); }
static inline int f_eq(const fld_t *K, const u64 *a, const u64 *b) { return limbs_eq(a, b, f_limbs(K)); }
static inline void f_add(const fld_t *K, u64 *r, const u64 *a, const u64 *b) {
    const int n = K->fp->n;
    for (int d = 0; d < K->deg; ++d) fp_add(K->fp, r + d * n, a + d * n, b + d * n);
}
static inline void f_sub(const fld_t *K, u64 *r, const u64 *a, const u64 *b) {
    const int n = K->fp->n;
    for (int d = 0; d < K->deg; ++d) fp_sub(K->fp, r + d * n, a + d * n, b + d * n);
}
static inline void f_neg(const fld_t *K, u64 *r, const u64 *a) {
    const int n = K->fp->n;
    for (int d = 0; d < K->deg; ++d) fp_neg(K->fp, r + d * n, a + d * n);
}
static inline void f_dbl(const fld_t *K, u64 *r, const u64 *a) { f_add(K, r, a, a); }
static inline void f_mul(const fld_t *K, u64 *r, const u64 *a, const u64 *b) {
    const fp_t *F = K->fp;
    if (K->deg == 1) {
        fp_mul(F, r, a, b);
        return;
    }
    const int n = F->n;
    u64 v0[MO_MAXL], v1[MO_MAXL], s[MO_MAXL], t[MO_MAXL];
    fp_mul(F, v0, a, b);
    fp_mul(F, v1, a + n, b + n);
    fp_add(F, s, a, a + n);
    fp_add(F, t, b, b + n);
    fp_mul(F, s, s, t); /* (a0+a1)(b0+b1) */
    fp_sub(F, s, s, v0);
    fp_sub(F, s, s, v1);
    fp_sub(F, r, v0, v1); /* c0 = a0b0 - a1b1 */
    fp_copy(F, r + n, s); /* c1 */
}
static inline void f_sqr(const fld_t *K, u64 *r, const u64 *a) { f_mul(K, r, a, a); }
static inline void f_mul_fp(const fld_t *K, u64 *r, const u64 *a, const u64 *s /* Fp scalar */) {
    const int n = K->fp->n;
    for (int d = 0; d < K->deg; ++d) fp_mul(K->fp, r + d * n, a + d * n, s);
}
static inline void f_inv(const fld_t *K, u64 *r, const u64 *a) {
    const fp_t *F = K->fp;
    if (K->deg == 1) {
        fp_inv(F, r, a);
        return;
    }
    const int n = F->n;
    u64 t0[MO_MAXL], t1[MO_MAXL];
    fp_sqr(F, t0, a);
    fp_sqr(F, t1, a + n);
    fp_add(F, t0, t0, t1); /* norm */
    fp_inv(F, t0, t0);
    fp_mul(F, t1, a + n, t0);
    fp_mul(F, r, a, t0);
    fp_neg(F, r + n, t1);
}
static inline void f_conj(const fld_t *K, u64 *r, const u64 *a) { /* Frobenius of Fp2 */
    const int n = K->fp->n;
    fp_copy(K->fp, r, a);
    if (K->deg == 2) fp_neg(K->fp, r + n, a + n);
}
/* sqrt; deg 2 uses the complex method. returns 1 iff square */
static inline int f_sqrt(const fld_t *K, u64 *r, const u64 *a) {
    const fp_t *F = K->fp;
    if (K->deg == 1) return fp_sqrt(F, r, a);
    const int n = F->n;
    if (fp_is_zero(F, a + n)) { /* a in Fp: sqrt(a0) or u*sqrt(-a0) */
        u64 s[MO_MAXL], na[MO_MAXL];
        if (fp_sqrt(F, s, a)) {
            fp_copy(F, r, s);
            fp_zero(F, r + n);
            return 1;
        }
        fp_neg(F, na, a);
        if (!fp_sqrt(F, s, na)) return 0;
        fp_zero(F, r);
        fp_copy(F, r + n, s);
        return 1;
    }
    u64 nrm[MO_MAXL], t[MO_MAXL], s[MO_MAXL], half[MO_MAXL], x0[MO_MAXL], x1[MO_MAXL], two[MO_MAXL];
    fp_sqr(F, nrm, a);
    fp_sqr(F, t, a + n);
    fp_add(F, nrm, nrm, t);
    if (!fp_sqrt(F, s, nrm)) return 0;
    fp_set_u64(F, two, 2);
    fp_inv(F, half, two);
    fp_add(F, t, a, s);
    fp_mul(F, t, t, half);
    if (!fp_sqrt(F, x0, t)) {
        fp_sub(F, t, a, s);
        fp_mul(F, t, t, half);
        if (!fp_sqrt(F, x0, t)) return 0;
    }
    fp_add(F, t, x0, x0);
    fp_inv(F, t, t);
    fp_mul(F, x1, a + n, t);
    u64 cand[MO_MAXE], chk[MO_MAXE];
    fp_copy(F, cand, x0);
    fp_copy(F, cand + n, x1);
    f_sqr(K, chk, cand);
    if (!f_eq(K, chk, a)) return 0;
    f_copy(K, r, cand);
    return 1;
}
/* arkworks sign rule for point compression: Fp: y > -y as integers;
 * Fp2: compare c1 first, then c0. */
static inline int f_is_high(const fld_t *K, const u64 *a) {
    const fp_t *F = K->fp;
    if (K->deg == 1) return fp_is_high(F, a);
    const int n = F->n;
    if (!fp_is_zero(F, a + n)) return fp_is_high(F, a + n);
    return fp_is_high(F, a);
}
#endif
