// Device-side prime-field arithmetic for gfx950 (CDNA4), 32-bit limbs held in VGPRs.
//
// Replaces (on the GPU) ark-ff ^0.3.0 Fp256/Fp384 Montgomery arithmetic, which the reference reaches
// through manta-crypto/src/arkworks/groth16.rs:597. Same value semantics:
// little-endian limbs, Montgomery form with R = 2^(32*N) (= 2^256 / 2^384 as in arkworks), fully
// reduced results -- so device words are bit-identical to arkworks' in-memory u64 limbs.
//
// Multiplication is product-scanning (column-wise) Montgomery: every 32x32 product is one
// v_mad_u64_u32 into a 96-bit column accumulator plus one v_addc_co_u32 for the carry -- CDNA has no
// multiply-with-carry-in, and this form needs no operand moves (hipcc's u64 C lowering costs ~2.3
// v_mov + 1 v_lshl_add_u64 per product; measured in DESIGN.md). 2N^2+N multiplies per mont-mul.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mg {

typedef uint32_t u32;
typedef uint64_t u64;

#define MG_DEV __device__ __forceinline__
#define MG_HD __host__ __device__ __forceinline__
// A 12-limb Montgomery product is ~700 instructions (4.5 KB); a mixed add inlines ten of them. Fully
// inlined, one accumulate loop body is ~100 KB of straight-line code -- larger than the instruction
// cache -- and hipcc needs minutes and spills. The multiply is therefore a real function
// (s_swappc_b64; operands by value in VGPRs): call overhead is ~3 % of the product itself.
#ifndef MG_MUL_ATTR
#define MG_MUL_ATTR __device__ __noinline__
#endif
#ifndef MG_FP2_ATTR
#define MG_FP2_ATTR __device__ __forceinline__
#endif

struct Acc96 {
    u64 lo;
    u32 hi;
};
// c += a*b  (vector x vector)
MG_DEV void mac_vv(Acc96 &c, u32 a, u32 b) {
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"
        : "+v"(c.lo), "+v"(c.hi)
        : "v"(a), "v"(b)
        : "vcc");
}
// c += a*k  (k wave-uniform constant: lives in an SGPR, VOP3 on gfx9 takes no literal)
MG_DEV void mac_vs(Acc96 &c, u32 a, u32 k) {
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"
        : "+v"(c.lo), "+v"(c.hi)
        : "v"(a), "s"(k)
        : "vcc");
}
MG_DEV void acc_shr32(Acc96 &c) {
    c.lo = (c.lo >> 32) | ((u64)c.hi << 32);
    c.hi = 0;
}
// Q accumulators advanced in ONE asm block (see Fp::mul_many): the compiler's scheduler otherwise regroups separate blocks
// into one chain per accumulator, which is the dependent sequence the interleaving is meant to avoid
MG_DEV void mac3_vv(Acc96 &c0, Acc96 &c1, Acc96 &c2, u32 a0, u32 b0, u32 a1, u32 b1, u32 a2, u32 b2) {
    asm("v_mad_u64_u32 %0, vcc, %6, %7, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
        "v_mad_u64_u32 %2, vcc, %8, %9, %2\n\tv_addc_co_u32 %3, vcc, 0, %3, vcc\n\t"
        "v_mad_u64_u32 %4, vcc, %10, %11, %4\n\tv_addc_co_u32 %5, vcc, 0, %5, vcc"
        : "+v"(c0.lo), "+v"(c0.hi), "+v"(c1.lo), "+v"(c1.hi), "+v"(c2.lo), "+v"(c2.hi)
        : "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(a2), "v"(b2)
        : "vcc");
}
MG_DEV void mac3_vs(Acc96 &c0, Acc96 &c1, Acc96 &c2, u32 a0, u32 a1, u32 a2, u32 k) {
    asm("v_mad_u64_u32 %0, vcc, %6, %9, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
        "v_mad_u64_u32 %2, vcc, %7, %9, %2\n\tv_addc_co_u32 %3, vcc, 0, %3, vcc\n\t"
        "v_mad_u64_u32 %4, vcc, %8, %9, %4\n\tv_addc_co_u32 %5, vcc, 0, %5, vcc"
        : "+v"(c0.lo), "+v"(c0.hi), "+v"(c1.lo), "+v"(c1.hi), "+v"(c2.lo), "+v"(c2.hi)
        : "v"(a0), "v"(a1), "v"(a2), "s"(k)
        : "vcc");
}
MG_DEV void mac2_vv(Acc96 &c0, Acc96 &c1, u32 a0, u32 b0, u32 a1, u32 b1) {
    asm("v_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
        "v_mad_u64_u32 %2, vcc, %6, %7, %2\n\tv_addc_co_u32 %3, vcc, 0, %3, vcc"
        : "+v"(c0.lo), "+v"(c0.hi), "+v"(c1.lo), "+v"(c1.hi)
        : "v"(a0), "v"(b0), "v"(a1), "v"(b1)
        : "vcc");
}
MG_DEV void mac2_vs(Acc96 &c0, Acc96 &c1, u32 a0, u32 a1, u32 k) {
    asm("v_mad_u64_u32 %0, vcc, %4, %6, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
        "v_mad_u64_u32 %2, vcc, %5, %6, %2\n\tv_addc_co_u32 %3, vcc, 0, %3, vcc"
        : "+v"(c0.lo), "+v"(c0.hi), "+v"(c1.lo), "+v"(c1.hi)
        : "v"(a0), "v"(a1), "s"(k)
        : "vcc");
}

// C supplies: static constexpr int N; static constexpr u32 P[N], R[N] (one), R2[N], INV;
template <class C> struct Fp {
    static constexpr int N = C::N;
    static constexpr bool EXT = false;
    u32 v[N];

    static MG_DEV Fp zero() {
        Fp r;
#pragma unroll
        for (int i = 0; i < N; ++i) r.v[i] = 0;
        return r;
    }
    static MG_DEV Fp one() {
        Fp r;
#pragma unroll
        for (int i = 0; i < N; ++i) r.v[i] = C::R[i];
        return r;
    }
    MG_DEV bool is_zero() const {
        u32 x = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) x |= v[i];
        return x == 0;
    }
    MG_DEV bool operator==(const Fp &o) const {
        u32 x = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) x |= v[i] ^ o.v[i];
        return x == 0;
    }
    // r = a - P if a >= P (a < 2P)
    static MG_DEV Fp reduce_once(const Fp &a, u32 top = 0) {
        Fp s;
        u32 bw = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            u64 d = (u64)a.v[i] - C::P[i] - bw;
            s.v[i] = (u32)d;
            bw = (u32)(d >> 63);
        }
        const bool ge = (top != 0) | (bw == 0);
        Fp r;
#pragma unroll
        for (int i = 0; i < N; ++i) r.v[i] = ge ? s.v[i] : a.v[i];
        return r;
    }
    static MG_DEV Fp add(const Fp &a, const Fp &b) {
        Fp t;
        u32 c = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            u64 s = (u64)a.v[i] + b.v[i] + c;
            t.v[i] = (u32)s;
            c = (u32)(s >> 32);
        }
        return reduce_once(t, c); // 2P < 2^(32N): c is 0, kept for safety
    }
    static MG_DEV Fp sub(const Fp &a, const Fp &b) {
        Fp t;
        u32 bw = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            u64 d = (u64)a.v[i] - b.v[i] - bw;
            t.v[i] = (u32)d;
            bw = (u32)(d >> 63);
        }
        const u32 mask = 0u - bw; // add P back on borrow
        u32 c = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            u64 s = (u64)t.v[i] + (C::P[i] & mask) + c;
            t.v[i] = (u32)s;
            c = (u32)(s >> 32);
        }
        return t;
    }
    static MG_DEV Fp neg(const Fp &a) {
        Fp z = zero();
        return a.is_zero() ? z : sub(z, a);
    }
    static MG_DEV Fp dbl(const Fp &a) { return add(a, a); }
    // Interface shared with the lazily-reduced FpR (fpr_dev.h): the multiple-of-p / bound template
    // arguments are meaningless for canonical values and ignored here.
    static constexpr bool LAZY = false;
    static constexpr int BM = 1, LIM = 1 << 30, MULK = 1, MAXM = 1 << 30, BX = 1, BY = 1, BRED = 1;
    template <int A> static MG_DEV Fp reduce(const Fp &a) { return a; }
    template <int A, int B> static MG_DEV Fp mulb(const Fp &a, const Fp &b) { return mul(a, b); }
    template <int A> static MG_DEV Fp sqrb(const Fp &a) { return sqr(a); }
    template <int M> static MG_DEV Fp sub(const Fp &a, const Fp &b) { return sub(a, b); }
    template <int M> static MG_DEV Fp sub2(const Fp &a, const Fp &b, const Fp &c) { return sub(sub(a, b), dbl(c)); }
    template <int M> static MG_DEV Fp neg(const Fp &a) { return neg(a); }
    template <int B> MG_DEV bool is_zero_mod() const { return is_zero(); }
    MG_DEV bool is_zero_exact() const { return is_zero(); }
    typedef Fp Std;
    static MG_DEV Fp from_std(const Fp &s) { return s; }
    MG_DEV Fp to_std() const { return *this; }

    // Montgomery product a*b*R^-1 mod P, product scanning with interleaved reduction.
    static MG_MUL_ATTR Fp mul(const Fp a, const Fp b) {
        Acc96 c{0, 0};
        u32 m[N];
        Fp t;
#pragma unroll
        for (int k = 0; k < N; ++k) {
#pragma unroll
            for (int i = 0; i < k; ++i) {
                mac_vv(c, a.v[i], b.v[k - i]);
                mac_vs(c, m[i], C::P[k - i]);
            }
            mac_vv(c, a.v[k], b.v[0]);
            m[k] = (u32)c.lo * C::INV;
            mac_vs(c, m[k], C::P[0]);
            acc_shr32(c);
        }
#pragma unroll
        for (int k = N; k < 2 * N; ++k) {
#pragma unroll
            for (int i = k - N + 1; i < N; ++i) {
                mac_vv(c, a.v[i], b.v[k - i]);
                mac_vs(c, m[i], C::P[k - i]);
            }
            t.v[k - N] = (u32)c.lo;
            acc_shr32(c);
        }
        return reduce_once(t, (u32)c.lo);
    }
    static MG_DEV Fp sqr(const Fp &a) { return mul(a, a); }

    // Q independent Montgomery products in ONE instruction stream (the Karatsuba terms of an Fq2 product, the two halves
    // of an Fq2-by-Fq product). A single product is one chain of dependent multiply-adds; with a lone wavefront on the SIMD
    // (the wave-cooperative pairing, pairing_coop.h) each costs ~12 cycles, and Q interleaved chains issue back to back.
    template <int Q> static MG_DEV void mul_many(const Fp (&a)[Q], const Fp (&b)[Q], Fp (&r)[Q]) {
        static_assert(Q == 2 || Q == 3, "two or three interleaved products");
        Acc96 c[Q];
        u32 m[Q][N];
        Fp t[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) c[q] = Acc96{0, 0};
        auto vv = [&](int i, int j) {
            if constexpr (Q == 3) mac3_vv(c[0], c[1], c[2], a[0].v[i], b[0].v[j], a[1].v[i], b[1].v[j], a[2].v[i], b[2].v[j]);
            else mac2_vv(c[0], c[1], a[0].v[i], b[0].v[j], a[1].v[i], b[1].v[j]);
        };
        auto vs = [&](int i, u32 k) {
            if constexpr (Q == 3) mac3_vs(c[0], c[1], c[2], m[0][i], m[1][i], m[2][i], k);
            else mac2_vs(c[0], c[1], m[0][i], m[1][i], k);
        };
#pragma unroll
        for (int k = 0; k < N; ++k) {
#pragma unroll
            for (int i = 0; i < k; ++i) {
                vv(i, k - i);
                vs(i, C::P[k - i]);
            }
            vv(k, 0);
#pragma unroll
            for (int q = 0; q < Q; ++q) m[q][k] = (u32)c[q].lo * C::INV;
            vs(k, C::P[0]);
#pragma unroll
            for (int q = 0; q < Q; ++q) acc_shr32(c[q]);
        }
#pragma unroll
        for (int k = N; k < 2 * N; ++k) {
#pragma unroll
            for (int i = k - N + 1; i < N; ++i) {
                vv(i, k - i);
                vs(i, C::P[k - i]);
            }
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                t[q].v[k - N] = (u32)c[q].lo;
                acc_shr32(c[q]);
            }
        }
#pragma unroll
        for (int q = 0; q < Q; ++q) r[q] = reduce_once(t[q], (u32)c[q].lo);
    }

    // Montgomery -> canonical (ark-ff into_repr): a * 1 * R^-1
    static MG_DEV Fp from_mont(const Fp &a) {
        Fp o = zero();
        o.v[0] = 1;
        return mul(a, o);
    }
    static MG_DEV Fp to_mont(const Fp &a) {
        Fp r2;
#pragma unroll
        for (int i = 0; i < N; ++i) r2.v[i] = C::R2[i];
        return mul(a, r2);
    }
    // a^(P-2) (Fermat). Serial and slow: only for one-off conversions, never in a hot loop.
    static __device__ __noinline__ Fp inv(const Fp &a) {
        Fp acc = one();
        for (int i = 32 * N - 1; i >= 0; --i) {
            acc = sqr(acc);
            // exponent P-2, bit i
            u32 w = 0;
#pragma unroll
            for (int j = 0; j < N; ++j)
                if (j == (i >> 5)) w = C::PM2[j];
            if ((w >> (i & 31)) & 1) acc = mul(acc, a);
        }
        return acc;
    }
    static constexpr int AFF_N = C::N; // affine coordinates in memory: the plain limbs (fpr_dev.h packs the reduced-radix ones)
    static MG_DEV Fp load_aff(const u32 *p) { return load(p); }
    MG_DEV void store_aff(u32 *p) const { store(p); }
    static MG_DEV Fp load(const u32 *p) {
        Fp r;
#pragma unroll
        for (int i = 0; i < N; ++i) r.v[i] = p[i];
        return r;
    }
    MG_DEV void store(u32 *p) const {
#pragma unroll
        for (int i = 0; i < N; ++i) p[i] = v[i];
    }
    static MG_DEV Fp select(bool c, const Fp &a, const Fp &b) { // c ? a : b
        Fp r;
#pragma unroll
        for (int i = 0; i < N; ++i) r.v[i] = c ? a.v[i] : b.v[i];
        return r;
    }
    static MG_DEV Fp shfl(const Fp &a, int src_lane) {
        Fp r;
#pragma unroll
        for (int i = 0; i < N; ++i) r.v[i] = __shfl(a.v[i], src_lane, 64);
        return r;
    }
};

// Fp2 = Fp[u]/(u^2+1) (BN254 and BLS12-381 both use non-residue -1)
template <class C> struct Fp2 {
    typedef Fp<C> B;
    static constexpr int N = 2 * C::N;
    static constexpr bool EXT = true;
    B c0, c1;
    static MG_DEV Fp2 zero() { return Fp2{B::zero(), B::zero()}; }
    static MG_DEV Fp2 one() { return Fp2{B::one(), B::zero()}; }
    MG_DEV bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
    MG_DEV bool operator==(const Fp2 &o) const { return (c0 == o.c0) & (c1 == o.c1); }
    static MG_DEV Fp2 add(const Fp2 &a, const Fp2 &b) { return Fp2{B::add(a.c0, b.c0), B::add(a.c1, b.c1)}; }
    static MG_DEV Fp2 sub(const Fp2 &a, const Fp2 &b) { return Fp2{B::sub(a.c0, b.c0), B::sub(a.c1, b.c1)}; }
    static MG_DEV Fp2 neg(const Fp2 &a) { return Fp2{B::neg(a.c0), B::neg(a.c1)}; }
    static MG_DEV Fp2 dbl(const Fp2 &a) { return add(a, a); }
    static constexpr bool LAZY = false;
    static constexpr int BM = 1, LIM = 1 << 30, MULK = 1, MAXM = 1 << 30, BX = 1, BY = 1, BRED = 1;
    template <int A> static MG_DEV Fp2 reduce(const Fp2 &a) { return a; }
    template <int A, int B> static MG_DEV Fp2 mulb(const Fp2 &a, const Fp2 &b) { return mul(a, b); }
    template <int A> static MG_DEV Fp2 sqrb(const Fp2 &a) { return sqr(a); }
    template <int M> static MG_DEV Fp2 sub(const Fp2 &a, const Fp2 &b) { return sub(a, b); }
    template <int M> static MG_DEV Fp2 sub2(const Fp2 &a, const Fp2 &b, const Fp2 &c) { return sub(sub(a, b), dbl(c)); }
    template <int M> static MG_DEV Fp2 neg(const Fp2 &a) { return neg(a); }
    template <int B> MG_DEV bool is_zero_mod() const { return is_zero(); }
    MG_DEV bool is_zero_exact() const { return is_zero(); }
    typedef Fp2 Std;
    static MG_DEV Fp2 from_std(const Fp2 &s) { return s; }
    MG_DEV Fp2 to_std() const { return *this; }
    static MG_FP2_ATTR Fp2 mul(const Fp2 &a, const Fp2 &b) { // Karatsuba, 3 base mults
        B v0 = B::mul(a.c0, b.c0), v1 = B::mul(a.c1, b.c1);
        B s = B::mul(B::add(a.c0, a.c1), B::add(b.c0, b.c1));
        return Fp2{B::sub(v0, v1), B::sub(B::sub(s, v0), v1)};
    }
    static MG_FP2_ATTR Fp2 sqr(const Fp2 &a) { // (a0+a1)(a0-a1), 2 a0 a1
        B t = B::mul(B::add(a.c0, a.c1), B::sub(a.c0, a.c1));
        B u = B::mul(a.c0, a.c1);
        return Fp2{t, B::dbl(u)};
    }
    static constexpr int AFF_N = N;
    static MG_DEV Fp2 load_aff(const u32 *p) { return load(p); }
    MG_DEV void store_aff(u32 *p) const { store(p); }
    static MG_DEV Fp2 load(const u32 *p) { return Fp2{B::load(p), B::load(p + C::N)}; }
    MG_DEV void store(u32 *p) const {
        c0.store(p);
        c1.store(p + C::N);
    }
    static MG_DEV Fp2 select(bool c, const Fp2 &a, const Fp2 &b) {
        return Fp2{B::select(c, a.c0, b.c0), B::select(c, a.c1, b.c1)};
    }
    static MG_DEV Fp2 shfl(const Fp2 &a, int src_lane) { return Fp2{B::shfl(a.c0, src_lane), B::shfl(a.c1, src_lane)}; }
};

} // namespace mg
