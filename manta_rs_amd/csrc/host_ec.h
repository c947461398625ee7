// Host-side (CPU) field and group arithmetic of the PRODUCT -- used only for the few serial,
// latency-bound steps a GPU lane is bad at (a 255-step double-and-add chain costs ~1.5 ms on one
// gfx950 lane and ~80 us on one x86 core): folding the <= few dozen partial points an MSM returns,
// the six scalar multiplications and three inversions of ark-groth16's final proof assembly
// (SURVEY.md row a-9), and arkworks canonical point serialisation (App. A.3;
// manta-crypto/src/arkworks/groth16.rs:186-195). Independent of oracle/ (which is test-only).
#pragma once
#include "params_gen.h"
#include <cstring>
#include <stdint.h>

namespace mg {
namespace host {

typedef uint64_t u64;
typedef unsigned __int128 u128;

template <class C> struct HFp {
    static constexpr int N = C::N64;
    u64 v[N];

    static constexpr u64 p64(int i) { return (u64)C::P[2 * i] | ((u64)C::P[2 * i + 1] << 32); }
    static HFp zero() {
        HFp r;
        for (int i = 0; i < N; ++i) r.v[i] = 0;
        return r;
    }
    static HFp one() {
        HFp r;
        for (int i = 0; i < N; ++i) r.v[i] = (u64)C::R[2 * i] | ((u64)C::R[2 * i + 1] << 32);
        return r;
    }
    static HFp r2() {
        HFp r;
        for (int i = 0; i < N; ++i) r.v[i] = (u64)C::R2[2 * i] | ((u64)C::R2[2 * i + 1] << 32);
        return r;
    }
    bool is_zero() const {
        u64 x = 0;
        for (int i = 0; i < N; ++i) x |= v[i];
        return x == 0;
    }
    bool operator==(const HFp &o) const { return std::memcmp(v, o.v, sizeof(v)) == 0; }
    static bool geq_p(const u64 *a) {
        for (int i = N - 1; i >= 0; --i) {
            if (a[i] > p64(i)) return true;
            if (a[i] < p64(i)) return false;
        }
        return true;
    }
    static void sub_p(u64 *a) {
        u64 bw = 0;
        for (int i = 0; i < N; ++i) {
            u128 d = (u128)a[i] - p64(i) - bw;
            a[i] = (u64)d;
            bw = (u64)(d >> 64) & 1;
        }
    }
    static HFp add(const HFp &a, const HFp &b) {
        HFp r;
        u64 c = 0;
        for (int i = 0; i < N; ++i) {
            u128 s = (u128)a.v[i] + b.v[i] + c;
            r.v[i] = (u64)s;
            c = (u64)(s >> 64);
        }
        if (c || geq_p(r.v)) sub_p(r.v);
        return r;
    }
    static HFp sub(const HFp &a, const HFp &b) {
        HFp r;
        u64 bw = 0;
        for (int i = 0; i < N; ++i) {
            u128 d = (u128)a.v[i] - b.v[i] - bw;
            r.v[i] = (u64)d;
            bw = (u64)(d >> 64) & 1;
        }
        if (bw) {
            u64 c = 0;
            for (int i = 0; i < N; ++i) {
                u128 s = (u128)r.v[i] + p64(i) + c;
                r.v[i] = (u64)s;
                c = (u64)(s >> 64);
            }
        }
        return r;
    }
    static HFp neg(const HFp &a) { return a.is_zero() ? a : sub(zero(), a); }
    static HFp dbl(const HFp &a) { return add(a, a); }
    static HFp mul(const HFp &a, const HFp &b) {
        u64 t[N + 2];
        for (int i = 0; i < N + 2; ++i) t[i] = 0;
        for (int i = 0; i < N; ++i) {
            u128 c = 0;
            for (int j = 0; j < N; ++j) {
                c += (u128)a.v[j] * b.v[i] + t[j];
                t[j] = (u64)c;
                c >>= 64;
            }
            c += t[N];
            t[N] = (u64)c;
            t[N + 1] = (u64)(c >> 64);
            const u64 m = t[0] * C::INV64;
            c = (u128)m * p64(0) + t[0];
            c >>= 64;
            for (int j = 1; j < N; ++j) {
                c += (u128)m * p64(j) + t[j];
                t[j - 1] = (u64)c;
                c >>= 64;
            }
            c += t[N];
            t[N - 1] = (u64)c;
            t[N] = t[N + 1] + (u64)(c >> 64);
        }
        if (t[N] || geq_p(t)) sub_p(t);
        HFp r;
        for (int i = 0; i < N; ++i) r.v[i] = t[i];
        return r;
    }
    static HFp sqr(const HFp &a) { return mul(a, a); }
    static HFp from_mont(const HFp &a) {
        HFp o = zero();
        o.v[0] = 1;
        return mul(a, o);
    }
    static HFp to_mont(const HFp &a) { return mul(a, r2()); }
    static HFp inv(const HFp &a) { // a^(p-2)
        HFp acc = one();
        for (int i = 64 * N - 1; i >= 0; --i) {
            acc = sqr(acc);
            u64 w = (u64)C::PM2[2 * (i >> 6)] | ((u64)C::PM2[2 * (i >> 6) + 1] << 32);
            if ((w >> (i & 63)) & 1) acc = mul(acc, a);
        }
        return acc;
    }
    // canonical(a) > (p-1)/2  (arkworks "lexicographically largest")
    bool is_high() const {
        HFp c = from_mont(*this);
        // compare c with (p-1)/2: c > (p-1)/2  <=>  2c > p-1  <=>  2c >= p+1 ; p odd so 2c != p
        u64 d[N + 1];
        u64 cy = 0;
        for (int i = 0; i < N; ++i) {
            d[i] = (c.v[i] << 1) | cy;
            cy = c.v[i] >> 63;
        }
        d[N] = cy;
        if (d[N]) return true;
        return geq_p(d);
    }
    static constexpr int BYTES = (C::BITS + 7) / 8;
    void write_canonical(unsigned char *out) const {
        HFp c = from_mont(*this);
        for (int i = 0; i < BYTES; ++i) out[i] = (unsigned char)(c.v[i >> 3] >> ((i & 7) * 8));
    }
    void load_words(const uint32_t *w) {
        for (int i = 0; i < N; ++i) v[i] = (u64)w[2 * i] | ((u64)w[2 * i + 1] << 32);
    }
    void store_words(uint32_t *w) const {
        for (int i = 0; i < N; ++i) {
            w[2 * i] = (uint32_t)v[i];
            w[2 * i + 1] = (uint32_t)(v[i] >> 32);
        }
    }
    static constexpr int WORDS = 2 * N;
};

template <class C> struct HFp2 {
    typedef HFp<C> B;
    B c0, c1;
    static HFp2 zero() { return HFp2{B::zero(), B::zero()}; }
    static HFp2 one() { return HFp2{B::one(), B::zero()}; }
    bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
    bool operator==(const HFp2 &o) const { return c0 == o.c0 && c1 == o.c1; }
    static HFp2 add(const HFp2 &a, const HFp2 &b) { return HFp2{B::add(a.c0, b.c0), B::add(a.c1, b.c1)}; }
    static HFp2 sub(const HFp2 &a, const HFp2 &b) { return HFp2{B::sub(a.c0, b.c0), B::sub(a.c1, b.c1)}; }
    static HFp2 neg(const HFp2 &a) { return HFp2{B::neg(a.c0), B::neg(a.c1)}; }
    static HFp2 dbl(const HFp2 &a) { return add(a, a); }
    static HFp2 mul(const HFp2 &a, const HFp2 &b) {
        B v0 = B::mul(a.c0, b.c0), v1 = B::mul(a.c1, b.c1);
        B s = B::mul(B::add(a.c0, a.c1), B::add(b.c0, b.c1));
        return HFp2{B::sub(v0, v1), B::sub(B::sub(s, v0), v1)};
    }
    static HFp2 sqr(const HFp2 &a) { return mul(a, a); }
    static HFp2 inv(const HFp2 &a) {
        B n = B::inv(B::add(B::sqr(a.c0), B::sqr(a.c1)));
        return HFp2{B::mul(a.c0, n), B::neg(B::mul(a.c1, n))};
    }
    bool is_high() const { return c1.is_zero() ? c0.is_high() : c1.is_high(); } // c1 first, then c0
    static constexpr int BYTES = 2 * B::BYTES;
    void write_canonical(unsigned char *out) const {
        c0.write_canonical(out);
        c1.write_canonical(out + B::BYTES);
    }
    void load_words(const uint32_t *w) {
        c0.load_words(w);
        c1.load_words(w + B::WORDS);
    }
    void store_words(uint32_t *w) const {
        c0.store_words(w);
        c1.store_words(w + B::WORDS);
    }
    static constexpr int WORDS = 2 * B::WORDS;
};

// XYZZ point on y^2 = x^3 + b (b irrelevant for the group law when a = 0)
template <class F> struct HPoint {
    F x, y, zz, zzz;
    static HPoint inf() { return HPoint{F::zero(), F::zero(), F::zero(), F::zero()}; }
    bool is_inf() const { return zz.is_zero(); }
    static HPoint from_affine_words(const uint32_t *w) {
        HPoint p;
        p.x.load_words(w);
        p.y.load_words(w + F::WORDS);
        if (p.x.is_zero() && p.y.is_zero()) return inf();
        p.zz = F::one();
        p.zzz = F::one();
        return p;
    }
    static HPoint from_xyzz_words(const uint32_t *w) {
        HPoint p;
        p.x.load_words(w);
        p.y.load_words(w + F::WORDS);
        p.zz.load_words(w + 2 * F::WORDS);
        p.zzz.load_words(w + 3 * F::WORDS);
        return p;
    }
    HPoint neg() const { return HPoint{x, F::neg(y), zz, zzz}; }
    static HPoint dbl(const HPoint &p) {
        if (p.is_inf()) return p;
        F U = F::dbl(p.y), V = F::sqr(U), W = F::mul(U, V), S = F::mul(p.x, V), X2 = F::sqr(p.x);
        F M = F::add(F::dbl(X2), X2);
        F X3 = F::sub(F::sqr(M), F::dbl(S));
        F Y3 = F::sub(F::mul(M, F::sub(S, X3)), F::mul(W, p.y));
        return HPoint{X3, Y3, F::mul(V, p.zz), F::mul(W, p.zzz)};
    }
    static HPoint add(const HPoint &a, const HPoint &b) {
        if (b.is_inf()) return a;
        if (a.is_inf()) return b;
        F U1 = F::mul(a.x, b.zz), U2 = F::mul(b.x, a.zz), S1 = F::mul(a.y, b.zzz), S2 = F::mul(b.y, a.zzz);
        F P = F::sub(U2, U1), R = F::sub(S2, S1);
        if (P.is_zero()) return R.is_zero() ? dbl(a) : inf();
        F PP = F::sqr(P), PPP = F::mul(P, PP), Q = F::mul(U1, PP);
        F X3 = F::sub(F::sub(F::sqr(R), PPP), F::dbl(Q));
        F Y3 = F::sub(F::mul(R, F::sub(Q, X3)), F::mul(S1, PPP));
        return HPoint{X3, Y3, F::mul(F::mul(a.zz, b.zz), PP), F::mul(F::mul(a.zzz, b.zzz), PPP)};
    }
    // [k]p, k = nl little-endian 64-bit limbs (plain integer)
    static HPoint mul(const HPoint &p, const u64 *k, int nl) {
        HPoint acc = inf();
        int top = -1;
        for (int i = 64 * nl - 1; i >= 0; --i)
            if ((k[i >> 6] >> (i & 63)) & 1) {
                top = i;
                break;
            }
        for (int i = top; i >= 0; --i) {
            acc = dbl(acc);
            if ((k[i >> 6] >> (i & 63)) & 1) acc = add(acc, p);
        }
        return acc;
    }
    // [k1]p + [k2]q with one shared doubling chain (Shamir's trick)
    static HPoint mul2(const HPoint &p, const u64 *k1, const HPoint &q, const u64 *k2, int nl) {
        const HPoint pq = add(p, q);
        HPoint acc = inf();
        bool started = false;
        for (int i = 64 * nl - 1; i >= 0; --i) {
            const int b1 = (int)((k1[i >> 6] >> (i & 63)) & 1), b2 = (int)((k2[i >> 6] >> (i & 63)) & 1);
            if (started) acc = dbl(acc);
            if (b1 | b2) {
                acc = add(acc, b1 && b2 ? pq : (b1 ? p : q));
                started = true;
            }
        }
        return acc;
    }
    static HPoint mul_pow2(HPoint p, unsigned k) { // 2^k * p
        for (unsigned i = 0; i < k; ++i) p = dbl(p);
        return p;
    }
    // affine x,y (Montgomery); infinity -> zeros. returns false if infinity
    bool to_affine(F &ax, F &ay) const {
        if (is_inf()) {
            ax = F::zero();
            ay = F::zero();
            return false;
        }
        F t = F::inv(F::mul(zz, zzz));
        F izz = F::mul(t, zzz), izzz = F::mul(t, zz);
        ax = F::mul(x, izz);
        ay = F::mul(y, izzz);
        return true;
    }
    void to_affine_words(uint32_t *w) const {
        F ax, ay;
        to_affine(ax, ay);
        ax.store_words(w);
        ay.store_words(w + F::WORDS);
    }
    // arkworks canonical compressed / uncompressed bytes
    void serialize(unsigned char *out, bool compressed) const {
        const int xb = F::BYTES;
        F ax, ay;
        if (!to_affine(ax, ay)) {
            std::memset(out, 0, compressed ? xb : 2 * xb);
            out[(compressed ? xb : 2 * xb) - 1] |= 0x40;
            return;
        }
        ax.write_canonical(out);
        if (compressed) {
            if (ay.is_high()) out[xb - 1] |= 0x80;
        } else {
            ay.write_canonical(out + xb);
        }
    }
};

// Fixed-base multiplication table: T[w][d-1] = d * 16^w * B for 64 nibble windows. [k]B = 64 additions,
// no doublings -- used for the blinding terms r*delta, s*delta, rs*delta of every proof.
template <class HP> struct FixedBaseTable {
    HP t[64][15];
    void build(const HP &base) {
        HP cur = base;
        for (int w = 0; w < 64; ++w) {
            t[w][0] = cur;
            for (int d = 2; d <= 15; ++d) t[w][d - 1] = HP::add(t[w][d - 2], cur);
            cur = HP::add(t[w][14], cur); // 16 * cur
        }
    }
    HP mul(const u64 *k4) const {
        HP acc = HP::inf();
        for (int w = 0; w < 64; ++w) {
            const unsigned d = (unsigned)((k4[w >> 4] >> ((w & 15) * 4)) & 15);
            if (d) acc = HP::add(acc, t[w][d - 1]);
        }
        return acc;
    }
};

} // namespace host
} // namespace mg
