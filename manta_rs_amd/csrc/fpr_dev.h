// Reduced-radix ("carry-free") prime-field arithmetic for gfx950 -- the internal representation of
// the G1 MSM kernels.
//
// Why: CDNA4 has no multiply-with-carry-in. With saturated 32-bit limbs every 32x32 product costs one
// v_mad_u64_u32 (4.2 cycles/wave) PLUS one v_addc_co_u32 (4.3 cycles, serialised through VCC) -- half of
// Fp::mul's cycles are carries (measured: tools/ubench3.hip, profiles/r01_ubench3_reduced_radix_mul.txt).
// With K limbs of LB < 32 bits a whole column of 2K products fits a 64-bit accumulator, so a product is
// exactly one v_mad_u64_u32 and a column costs one shift + mask: 1.55x faster at the accumulate kernel's
// occupancy, although it needs (K/N)^2 ~ 1.3x more multiplies.
//
//   BLS12-381 Fq: K = 14 limbs x 28 bits (R' = 2^392);  BN254 Fq: K = 9 x 29 bits (R' = 2^261).
//
// Values are kept LAZILY reduced: limbs are always normalised (< 2^LB), the integer value may be any
// representative below a small multiple of p (bounds are tracked per call site in ec_dev.h). Because
// R' >= 2^7 p, an "almost Montgomery" product of inputs a < Ba*p, b < Bb*p is < 2p whenever
// Ba*Bb <= 128 (BN254) / 2048 (BLS12-381), with no final subtraction. Subtractions add a multiple M*p
// chosen from the subtrahend's bound. Exact tests (is_zero_mod) compare against the multiples of p.
//
// Memory format: K u32 words per element. Conversion to/from the arkworks format (32-bit limbs,
// R = 2^(32N)) happens once, when bases are registered and when the few result points are staged.
#pragma once
#include "fp_dev.h"

namespace mg {

// acc += a*b as ONE v_mad_u64_u32 the compiler may not re-associate: all multiply-adds of a column then form a single
// dependent chain (the C expression is split into an a*b chain and an m*p chain joined by a v_lshl_add_u64 per column:
// 28 more instructions per 14-limb product). Faster by 3.5-5.4 % at >= 2 wavefronts per SIMD, 35 % slower at one
// (profiles/r02_ubench4_fma_vs_int.txt) -- so only the throughput-bound accumulate kernel asks for it (CH = true), and
// there it pays only with SEVERAL links per asm block (below): -3 % kernel time on both curves.
MG_DEV void mad_chain_vv(u64 &acc, u32 a, u32 b) { asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b) : "vcc"); }
MG_DEV void mad_chain_vs(u64 &acc, u32 a, u32 k) { asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "s"(k) : "vcc"); }
// Several links of the chain in ONE asm block: the compiler puts an s_nop after every asm statement (it cannot see inside),
// which at one statement per multiply-add cost more than the single chain saved (~3 400 per loop body).
MG_DEV void mad_chain_pair(u64 &acc, u32 a, u32 b, u32 m, u32 k) { // a*b + m*k
    asm("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %3, %4, %0" : "+v"(acc) : "v"(a), "v"(b), "v"(m), "s"(k) : "vcc");
}
MG_DEV void mad_chain_pair2(u64 &acc, u32 a0, u32 b0, u32 m0, u32 k0, u32 a1, u32 b1, u32 m1, u32 k1) {
    asm("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %3, %4, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %5, %6, %0\n\tv_mad_u64_u32 %0, vcc, %7, %8, %0"
        : "+v"(acc)
        : "v"(a0), "v"(b0), "v"(m0), "s"(k0), "v"(a1), "v"(b1), "v"(m1), "s"(k1)
        : "vcc");
}
MG_DEV void mad_chain_vv2(u64 &acc, u32 a0, u32 b0, u32 a1, u32 b1) {
    asm("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %3, %4, %0" : "+v"(acc) : "v"(a0), "v"(b0), "v"(a1), "v"(b1) : "vcc");
}
MG_DEV void mad_chain_vs2(u64 &acc, u32 m0, u32 k0, u32 m1, u32 k1) {
    asm("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %3, %4, %0" : "+v"(acc) : "v"(m0), "s"(k0), "v"(m1), "s"(k1) : "vcc");
}
MG_DEV void mad_chain_triple(u64 &acc, u32 a, u32 b, u32 c, u32 d, u32 m, u32 k) { // a*b + c*d + m*k
    asm("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %3, %4, %0\n\tv_mad_u64_u32 %0, vcc, %5, %6, %0"
        : "+v"(acc)
        : "v"(a), "v"(b), "v"(c), "v"(d), "v"(m), "s"(k)
        : "vcc");
}
MG_DEV void mad_chain_triple2(u64 &acc, u32 a0, u32 b0, u32 c0, u32 d0, u32 m0, u32 k0, u32 a1, u32 b1, u32 c1, u32 d1, u32 m1, u32 k1) {
    asm("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %3, %4, %0\n\tv_mad_u64_u32 %0, vcc, %5, %6, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %7, %8, %0\n\tv_mad_u64_u32 %0, vcc, %9, %10, %0\n\tv_mad_u64_u32 %0, vcc, %11, %12, %0"
        : "+v"(acc)
        : "v"(a0), "v"(b0), "v"(c0), "v"(d0), "v"(m0), "s"(k0), "v"(a1), "v"(b1), "v"(c1), "v"(d1), "v"(m1), "s"(k1)
        : "vcc");
}

template <class C> struct FpR {
    static constexpr int K = C::RR_K, LB = C::RR_LB;
    static constexpr int N = K; // words per element in memory
    static constexpr bool EXT = false;
    static constexpr bool LAZY = true;
    static constexpr u32 MASK = (1u << LB) - 1;
    // Products of two LB-bit limbs a 64-bit column accumulator holds next to a carry-in (< 2^36). 28- and 29-bit limbs never
    // come near it (a fused column has 3 K <= 42 of them); with 30-bit limbs -- round 4: BLS12-381 Fq as 13 x 30 bits, 2 K^2 + K =
    // 351 multiply-adds per product instead of the 406 of 14 x 28 -- sixteen 60-bit products overflow, so a column that holds more
    // is FLUSHED between its product groups: the accumulator's upper part is moved aside (one 64-bit shift, one mask) and joined
    // again when the column is done. Nine of a product's 25 columns need it; measured per product at the accumulate kernel's
    // occupancy (tools/ubench_limbs.hip, profiles/r04_ubench_limbs.txt): 1960 against 2183 cycles (-10 %).
    static constexpr int CAP = LB >= 30 ? 15 : (LB == 29 ? 60 : 250);
    static constexpr int nprod(int k) { return k < K ? k + 1 : 2 * K - 1 - k; } // limb products a_i b_(k-i) of column k
    // bound bookkeeping consumed by ec_dev.h's Bv<> wrapper (values are "< B*p")
    static constexpr int BM = 2;            // a product is < 2p ...
    static constexpr int LIM = C::RR_LIM;   // ... whenever the operand bounds multiply to <= LIM
    static constexpr int MULK = 1;
    static constexpr int MAXM = C::RR_MAXM; // largest tabulated multiple of p
    static constexpr int BX = 8, BY = 4;    // invariants of stored XYZZ X and Y coordinates
    static constexpr int BRED = 2;          // reduce<>() output bound
    typedef Fp<C> Std;
    u32 v[K];

    static MG_DEV FpR zero() {
        FpR r;
#pragma unroll
        for (int i = 0; i < K; ++i) r.v[i] = 0;
        return r;
    }
    static MG_DEV FpR one() {
        FpR r;
#pragma unroll
        for (int i = 0; i < K; ++i) r.v[i] = C::RR_ONE[i];
        return r;
    }
    MG_DEV bool is_zero_exact() const { // the all-zero representative (used as the infinity marker)
        u32 x = 0;
#pragma unroll
        for (int i = 0; i < K; ++i) x |= v[i];
        return x == 0;
    }
    // value == 0 (mod p), given value < B*p : value is one of 0, p, ..., (B-1)p
    template <int B> MG_DEV bool is_zero_mod() const {
        static_assert(B - 1 <= C::RR_MAXM, "multiple table too short");
        bool hit = false;
#pragma unroll
        for (int k = 0; k < B; ++k) {
            if (v[0] == C::RR_MULT[k][0]) { // cheap pre-filter on the low limb; the full compare is rare
                u32 d = 0;
#pragma unroll
                for (int i = 1; i < K; ++i) d |= v[i] ^ C::RR_MULT[k][i];
                hit |= (d == 0);
            }
        }
        return hit;
    }

    // ---- linear operations: limb-wise in signed 32-bit, then one carry-normalisation pass
    static MG_DEV FpR normalize(const int (&t)[K]) {
        FpR r;
        int c = 0;
#pragma unroll
        for (int i = 0; i < K - 1; ++i) {
            const int s = t[i] + c;
            c = s >> LB; // arithmetic shift: borrows propagate as -1
            r.v[i] = (u32)s & MASK;
        }
        r.v[K - 1] = (u32)(t[K - 1] + c); // value >= 0 and < 2^(LB*K): top limb needs no mask
        return r;
    }
    static MG_DEV FpR add(const FpR &a, const FpR &b) {
        int t[K];
#pragma unroll
        for (int i = 0; i < K; ++i) t[i] = (int)(a.v[i] + b.v[i]);
        return normalize(t);
    }
    static MG_DEV FpR dbl(const FpR &a) { return add(a, a); }
    // a + M*p - b   (requires b < M*p)
    template <int M> static MG_DEV FpR sub(const FpR &a, const FpR &b) {
        static_assert(M <= C::RR_MAXM, "multiple table too short");
        int t[K];
#pragma unroll
        for (int i = 0; i < K; ++i) t[i] = (int)a.v[i] + (int)C::RR_MULT[M][i] - (int)b.v[i];
        return normalize(t);
    }
    // a + M*p - b - 2c   (requires b + 2c < M*p)
    template <int M> static MG_DEV FpR sub2(const FpR &a, const FpR &b, const FpR &c) {
        static_assert(M <= C::RR_MAXM, "multiple table too short");
        if constexpr (LB >= 30) { // a + Mp - b - 2c leaves the 32-bit range limb-wise (-3 * 2^30): carry every limb in 64 bits
            FpR r;
            long long cy = 0;
#pragma unroll
            for (int i = 0; i < K - 1; ++i) {
                const long long sv = (long long)a.v[i] + (long long)C::RR_MULT[M][i] - (long long)b.v[i] - 2 * (long long)c.v[i] + cy;
                cy = sv >> LB;
                r.v[i] = (u32)sv & MASK;
            }
            r.v[K - 1] = (u32)((long long)a.v[K - 1] + (long long)C::RR_MULT[M][K - 1] - (long long)b.v[K - 1] - 2 * (long long)c.v[K - 1] + cy);
            return r;
        } else {
            int t[K];
#pragma unroll
            for (int i = 0; i < K; ++i) t[i] = (int)a.v[i] + (int)C::RR_MULT[M][i] - (int)b.v[i] - 2 * (int)c.v[i];
            return normalize(t);
        }
    }
    template <int M> static MG_DEV FpR neg(const FpR &a) { return sub<M>(zero(), a); }

    // ---- carry-free subtraction: a + M*p - b with NO normalisation pass. M*p is taken in the redundant limb form
    //   q_0 = m_0 + 2^LB,  q_i = m_i + 2^LB - 1 (0 < i < K-1),  q_{K-1} = m_{K-1} - 1      (same value: each limb lends one
    // unit of 2^LB to the limb below), so that q_i - b_i >= 0 for every NORMALISED b (limbs < 2^LB, and b < (M-1)*p keeps
    // the top limb non-negative: m_{K-1}(M) - 1 >= m_{K-1}(M-1) + p_top - 2 >= b_top). The result's limbs are < 3 * 2^LB
    // ("lazy limbs"); it is a legal multiplication operand wherever the column accumulators have the room (LAZY_LIMBS
    // below), which saves the 3-instructions-per-limb carry pass of `sub`.
    static constexpr u32 multb(int M, int i) {
        return i == 0 ? C::RR_MULT[M][0] + (1u << LB) : (i < K - 1 ? C::RR_MULT[M][i] + MASK : C::RR_MULT[M][K - 1] - 1u);
    }
    // worst column of a fused product with lazy operands: K * (3*3 + 3*1 + 1) * 2^(2 LB) must stay below 2^64
    static constexpr bool LAZY_LIMBS = (double)K * 13.0 * (double)(1ull << LB) * (double)(1ull << LB) < 18446744073709551616.0;
    template <int M> static MG_DEV FpR subl(const FpR &a, const FpR &b) { // a may itself have lazy limbs only if a is normalised: see call sites
        static_assert(M <= C::RR_MAXM, "multiple table too short");
        FpR r;
#pragma unroll
        for (int i = 0; i < K; ++i) r.v[i] = a.v[i] + multb(M, i) - b.v[i];
        return r;
    }
    template <int M> static MG_DEV FpR negl(const FpR &b) { return subl<M>(zero(), b); }
    static MG_DEV FpR normalize_u(const FpR &a) { // lazy limbs (non-negative) -> normalised
        FpR r;
        u32 c = 0;
#pragma unroll
        for (int i = 0; i < K - 1; ++i) {
            const u32 t = a.v[i] + c;
            c = t >> LB;
            r.v[i] = t & MASK;
        }
        r.v[K - 1] = a.v[K - 1] + c;
        return r;
    }
    // (fused products with NORMALISED operands need K * 3 * 2^(2 LB) < 2^64: true for both fields)
    static_assert(3 * K <= CAP || LB >= 30, "fused column overflow");
    // ---- fused almost-Montgomery product (a*b + c*d) * R'^-1 with ONE reduction: two double-width products share the
    // column accumulators and the m*p pass -- 3 K^2 + K multiply-adds instead of 4 K^2 + 2 K. Value < 2p whenever
    // Ba*Bb + Bc*Bd <= LIM; limbs of the operands may be lazy when LAZY_LIMBS (column bound above).
    static MG_DEV FpR mul_add(const FpR &a, const FpR &b, const FpR &c, const FpR &d) {
        if constexpr (3 * K > CAP) return mul_add_flushed(a, b, c, d);
        u64 acc = 0;
        u32 m[K];
        FpR t;
#pragma unroll
        for (int k = 0; k < K; ++k) {
#pragma unroll
            for (int i = 0; i < k; ++i) {
                acc += (u64)a.v[i] * b.v[k - i];
                acc += (u64)c.v[i] * d.v[k - i];
                acc += (u64)m[i] * C::RR_P[k - i];
            }
            acc += (u64)a.v[k] * b.v[0];
            acc += (u64)c.v[k] * d.v[0];
            m[k] = ((u32)acc * C::RR_INV) & MASK;
            acc += (u64)m[k] * C::RR_P[0];
            acc >>= LB;
        }
#pragma unroll
        for (int k = K; k < 2 * K - 1; ++k) {
#pragma unroll
            for (int i = k - K + 1; i < K; ++i) {
                acc += (u64)a.v[i] * b.v[k - i];
                acc += (u64)c.v[i] * d.v[k - i];
                acc += (u64)m[i] * C::RR_P[k - i];
            }
            t.v[k - K] = (u32)acc & MASK;
            acc >>= LB;
        }
        t.v[K - 1] = (u32)acc;
        return t;
    }
    // ---- the three product routines for limbs wide enough that a column overflows (CAP): the products of a column are added group
    // by group -- a*b, then c*d, then m*p -- and the accumulator is flushed before a group that would not fit (see CAP)
    static MG_DEV void flush(u64 &acc, u64 &spill) {
        spill += acc >> LB;
        acc &= (u64)MASK;
    }
    static MG_DEV FpR mul_add_flushed(const FpR &a, const FpR &b, const FpR &c, const FpR &d) {
        u64 acc = 0;
        u32 m[K];
        FpR t;
#pragma unroll
        for (int k = 0; k < 2 * K - 1; ++k) {
            const int lo = k < K ? 0 : k - K + 1, hi = k < K ? k : K - 1, n = hi - lo + 1;
            u64 spill = 0;
#pragma unroll
            for (int i = lo; i <= hi; ++i) acc += (u64)a.v[i] * b.v[k - i];
            if (2 * n > CAP) flush(acc, spill);
#pragma unroll
            for (int i = lo; i <= hi; ++i) acc += (u64)c.v[i] * d.v[k - i];
            if (2 * n > CAP || 3 * n > CAP) flush(acc, spill);
#pragma unroll
            for (int i = lo; i <= hi; ++i)
                if (!(k < K && i == k)) acc += (u64)m[i] * C::RR_P[k - i];
            if (k < K) {
                m[k] = ((u32)acc * C::RR_INV) & MASK;
                acc += (u64)m[k] * C::RR_P[0];
            } else {
                t.v[k - K] = (u32)acc & MASK;
            }
            acc >>= LB;
            acc += spill;
        }
        t.v[K - 1] = (u32)acc;
        return t;
    }
    static MG_DEV FpR mul_flushed(const FpR &a, const FpR &b) {
        u64 acc = 0;
        u32 m[K];
        FpR t;
#pragma unroll
        for (int k = 0; k < 2 * K - 1; ++k) {
            const int lo = k < K ? 0 : k - K + 1, hi = k < K ? k : K - 1, n = hi - lo + 1;
            u64 spill = 0;
#pragma unroll
            for (int i = lo; i <= hi; ++i) acc += (u64)a.v[i] * b.v[k - i];
            if (2 * n > CAP) flush(acc, spill);
#pragma unroll
            for (int i = lo; i <= hi; ++i)
                if (!(k < K && i == k)) acc += (u64)m[i] * C::RR_P[k - i];
            if (k < K) {
                m[k] = ((u32)acc * C::RR_INV) & MASK;
                acc += (u64)m[k] * C::RR_P[0];
            } else {
                t.v[k - K] = (u32)acc & MASK;
            }
            acc >>= LB;
            acc += spill;
        }
        t.v[K - 1] = (u32)acc;
        return t;
    }
    static MG_DEV FpR sqr_flushed(const FpR &a) {
        u64 acc = 0;
        u32 m[K], a2[K];
        FpR t;
#pragma unroll
        for (int i = 0; i < K; ++i) a2[i] = a.v[i] << 1; // (< 2^31: a doubled cross product counts as two)
#pragma unroll
        for (int k = 0; k < 2 * K - 1; ++k) {
            const int lo = k < K ? 0 : k - K + 1, hi = k < K ? k : K - 1, n = hi - lo + 1;
            u64 spill = 0;
#pragma unroll
            for (int i = lo; i <= hi; ++i) {
                const int j = k - i;
                if (i < j) acc += (u64)a2[i] * a.v[j];
                if (i == j) acc += (u64)a.v[i] * a.v[i];
            }
            if (2 * n > CAP) flush(acc, spill);
#pragma unroll
            for (int i = lo; i <= hi; ++i)
                if (!(k < K && i == k)) acc += (u64)m[i] * C::RR_P[k - i];
            if (k < K) {
                m[k] = ((u32)acc * C::RR_INV) & MASK;
                acc += (u64)m[k] * C::RR_P[0];
            } else {
                t.v[k - K] = (u32)acc & MASK;
            }
            acc >>= LB;
            acc += spill;
        }
        t.v[K - 1] = (u32)acc;
        return t;
    }
    // value < A*p (A <= 16)  ->  same residue, < 2p: subtract floor-estimate(value / p) * p, the
    // quotient estimated from the top limb (never too large, at most one too small)
    // (valid for any A with A*p < 2^(LB*K): with T the top limb and D = p_top + 1, the estimate q' = floor(T*RECIP/2^32) is
    // never above value/p and lies less than 1/8 + (A + 1)/p_top < 1 below T/D <= value/p, p_top >= 2^16 for all four fields)
    template <int A> static MG_DEV FpR reduce(const FpR &a) {
        static_assert(A <= 64 && A <= C::RR_LIM, "reduce: A*p must fit the representation");
        const u32 q = (u32)(((u64)a.v[K - 1] * C::RR_RECIP) >> 32);
        FpR r;
        long long c = 0;
#pragma unroll
        for (int i = 0; i < K - 1; ++i) {
            const long long s = (long long)a.v[i] - (long long)((u64)q * C::RR_P[i]) + c;
            c = s >> LB;
            r.v[i] = (u32)s & MASK;
        }
        r.v[K - 1] = (u32)((long long)a.v[K - 1] - (long long)((u64)q * C::RR_P[K - 1]) + c);
        return r;
    }

    template <int A, int B> static MG_DEV FpR mulb(const FpR &a, const FpR &b) { return mul(a, b); }
    // ---- almost-Montgomery product: a*b*R'^-1 mod p, result < 2p (see header for the input bounds)
    static MG_DEV FpR mul(const FpR &a, const FpR &b) {
        if constexpr (2 * K > CAP) return mul_flushed(a, b);
        u64 acc = 0;
        u32 m[K];
        FpR t;
#pragma unroll
        for (int k = 0; k < K; ++k) {
#pragma unroll
            for (int i = 0; i < k; ++i) {
                acc += (u64)a.v[i] * b.v[k - i];
                acc += (u64)m[i] * C::RR_P[k - i];
            }
            acc += (u64)a.v[k] * b.v[0];
            m[k] = ((u32)acc * C::RR_INV) & MASK;
            acc += (u64)m[k] * C::RR_P[0];
            acc >>= LB;
        }
#pragma unroll
        for (int k = K; k < 2 * K - 1; ++k) {
#pragma unroll
            for (int i = k - K + 1; i < K; ++i) {
                acc += (u64)a.v[i] * b.v[k - i];
                acc += (u64)m[i] * C::RR_P[k - i];
            }
            t.v[k - K] = (u32)acc & MASK;
            acc >>= LB;
        }
        t.v[K - 1] = (u32)acc;
        return t;
    }
    // single-chain codings (see mad_chain_vv) of mul / sqr / mul_add
    template <bool CH> static MG_DEV FpR mul_t(const FpR &a, const FpR &b) {
        if constexpr (!CH || 2 * K > CAP) return mul(a, b); // (the single-chain codings below have no flushes)
        u64 acc = 0;
        u32 m[K];
        FpR t;
#pragma unroll
        for (int k = 0; k < K; ++k) {
#pragma unroll
            for (int i = 0; i + 1 < k; i += 2)
                mad_chain_pair2(acc, a.v[i], b.v[k - i], m[i], C::RR_P[k - i], a.v[i + 1], b.v[k - i - 1], m[i + 1], C::RR_P[k - i - 1]);
            if (k & 1) mad_chain_pair(acc, a.v[k - 1], b.v[1], m[k - 1], C::RR_P[1]);
            mad_chain_vv(acc, a.v[k], b.v[0]);
            m[k] = ((u32)acc * C::RR_INV) & MASK;
            mad_chain_vs(acc, m[k], C::RR_P[0]);
            acc >>= LB;
        }
#pragma unroll
        for (int k = K; k < 2 * K - 1; ++k) {
            const int lo = k - K + 1, n = K - lo; // pairs i = lo .. K-1
#pragma unroll
            for (int q = 0; q + 1 < n; q += 2) {
                const int i = lo + q;
                mad_chain_pair2(acc, a.v[i], b.v[k - i], m[i], C::RR_P[k - i], a.v[i + 1], b.v[k - i - 1], m[i + 1], C::RR_P[k - i - 1]);
            }
            if (n & 1) mad_chain_pair(acc, a.v[K - 1], b.v[k - K + 1], m[K - 1], C::RR_P[k - K + 1]);
            t.v[k - K] = (u32)acc & MASK;
            acc >>= LB;
        }
        t.v[K - 1] = (u32)acc;
        return t;
    }
    template <bool CH> static MG_DEV FpR sqr_t(const FpR &a) {
        if constexpr (!CH || 2 * K > CAP) return sqr(a);
        u64 acc = 0;
        u32 m[K], a2[K];
        FpR t;
#pragma unroll
        for (int i = 0; i < K; ++i) a2[i] = a.v[i] << 1;
#pragma unroll
        for (int k = 0; k < 2 * K - 1; ++k) {
            // cross products a_i a_j (i < j, i + j = k) against the doubled operand, two per asm block
            const int ilo = k < K ? 0 : k - K + 1, ihi = (k - 1) / 2; // i = ilo .. ihi with i < k - i
#pragma unroll
            for (int i = ilo; i + 1 <= ihi; i += 2) mad_chain_vv2(acc, a2[i], a.v[k - i], a2[i + 1], a.v[k - i - 1]);
            if (k >= 1 && ((ihi - ilo + 1) & 1) && ihi >= ilo) mad_chain_vv(acc, a2[ihi], a.v[k - ihi]);
            if (!(k & 1)) mad_chain_vv(acc, a.v[k / 2], a.v[k / 2]);
            // the m_i p_{k-i} terms already known, two per block
            const int mlo = k < K ? 0 : k - K + 1, mhi = k < K ? k - 1 : K - 1;
#pragma unroll
            for (int i = mlo; i + 1 <= mhi; i += 2) mad_chain_vs2(acc, m[i], C::RR_P[k - i], m[i + 1], C::RR_P[k - i - 1]);
            if (mhi >= mlo && ((mhi - mlo + 1) & 1)) mad_chain_vs(acc, m[mhi], C::RR_P[k - mhi]);
            if (k < K) {
                m[k] = ((u32)acc * C::RR_INV) & MASK;
                mad_chain_vs(acc, m[k], C::RR_P[0]);
            } else {
                t.v[k - K] = (u32)acc & MASK;
            }
            acc >>= LB;
        }
        t.v[K - 1] = (u32)acc;
        return t;
    }
    template <bool CH> static MG_DEV FpR mul_add_t(const FpR &a, const FpR &b, const FpR &c, const FpR &d) {
        if constexpr (!CH || 2 * K > CAP) return mul_add(a, b, c, d);
        u64 acc = 0;
        u32 m[K];
        FpR t;
#pragma unroll
        for (int k = 0; k < K; ++k) {
#pragma unroll
            for (int i = 0; i + 1 < k; i += 2)
                mad_chain_triple2(acc, a.v[i], b.v[k - i], c.v[i], d.v[k - i], m[i], C::RR_P[k - i], a.v[i + 1], b.v[k - i - 1],
                                  c.v[i + 1], d.v[k - i - 1], m[i + 1], C::RR_P[k - i - 1]);
            if (k & 1) mad_chain_triple(acc, a.v[k - 1], b.v[1], c.v[k - 1], d.v[1], m[k - 1], C::RR_P[1]);
            mad_chain_vv(acc, a.v[k], b.v[0]);
            mad_chain_vv(acc, c.v[k], d.v[0]);
            m[k] = ((u32)acc * C::RR_INV) & MASK;
            mad_chain_vs(acc, m[k], C::RR_P[0]);
            acc >>= LB;
        }
#pragma unroll
        for (int k = K; k < 2 * K - 1; ++k) {
            const int lo = k - K + 1, n = K - lo;
#pragma unroll
            for (int q = 0; q + 1 < n; q += 2) {
                const int i = lo + q;
                mad_chain_triple2(acc, a.v[i], b.v[k - i], c.v[i], d.v[k - i], m[i], C::RR_P[k - i], a.v[i + 1], b.v[k - i - 1],
                                  c.v[i + 1], d.v[k - i - 1], m[i + 1], C::RR_P[k - i - 1]);
            }
            if (n & 1) mad_chain_triple(acc, a.v[K - 1], b.v[k - K + 1], c.v[K - 1], d.v[k - K + 1], m[K - 1], C::RR_P[k - K + 1]);
            t.v[k - K] = (u32)acc & MASK;
            acc >>= LB;
        }
        t.v[K - 1] = (u32)acc;
        return t;
    }
    // the same as real functions: for callers whose register budget cannot take the inlined products
    static __device__ __noinline__ FpR mul_call(const FpR a, const FpR b) { return mul(a, b); }
    static __device__ __noinline__ FpR sqr_call(const FpR a) { return sqr(a); }
    // (operands by REFERENCE: with four 14-limb structs passed by value -- 56 VGPRs of arguments -- hipcc of ROCm 7.2 produced a
    // function that returned wrong products on gfx950: BLS12-381 G2 MSMs failed against the oracle, the by-reference build of the
    // same source passes; found by bisecting builds on the GPU, round 4)
    static __device__ __noinline__ FpR mul_add_call(const FpR &a, const FpR &b, const FpR &c, const FpR &d) { return mul_add(a, b, c, d); }
    template <int A> static MG_DEV FpR sqrb(const FpR &a) { return sqr(a); }
    // square: cross products once, against the doubled operand
    static MG_DEV FpR sqr(const FpR &a) {
        if constexpr (2 * K > CAP) return sqr_flushed(a);
        u64 acc = 0;
        u32 m[K], a2[K];
        FpR t;
#pragma unroll
        for (int i = 0; i < K; ++i) a2[i] = a.v[i] << 1;
#pragma unroll
        for (int k = 0; k < 2 * K - 1; ++k) {
#pragma unroll
            for (int i = 0; i < K; ++i) {
                const int j = k - i;
                if (j < 0 || j >= K) continue;
                if (i < j) acc += (u64)a2[i] * a.v[j];
                if (i == j) acc += (u64)a.v[i] * a.v[i];
                if (k < K) {
                    if (i < k) acc += (u64)m[i] * C::RR_P[k - i];
                } else {
                    acc += (u64)m[i] * C::RR_P[k - i];
                }
            }
            if (k < K) {
                m[k] = ((u32)acc * C::RR_INV) & MASK;
                acc += (u64)m[k] * C::RR_P[0];
            } else {
                t.v[k - K] = (u32)acc & MASK;
            }
            acc >>= LB;
        }
        t.v[K - 1] = (u32)acc;
        return t;
    }
    static __device__ __noinline__ FpR inv(const FpR &a) { // a^(p-2); slow, one-off use only
        FpR acc = one();
        for (int i = 32 * C::N - 1; i >= 0; --i) {
            acc = sqr(acc);
            u32 w = 0;
#pragma unroll
            for (int j = 0; j < C::N; ++j)
                if (j == (i >> 5)) w = C::PM2[j];
            if ((w >> (i & 31)) & 1) acc = mul(acc, a);
        }
        return acc;
    }

    // ---- cheap conversions for fields with R'/R = 2^(LB*K - 32N) small (Fr: 2^261 / 2^256 = 2^5): the arkworks Montgomery
    // form a*R is turned into a*R' by SHIFTING the integer left while repacking its limbs (value < 2^5 p) and one reduce --
    // no multiplication; and a*R' becomes the canonical integer a by one almost-Montgomery product with the integer 1.
    static MG_DEV FpR from_std_shift(const Std &s) {
        constexpr int SH = LB * K - 32 * C::N; // R' = 2^SH * R
        static_assert(SH >= 0 && SH < 8 && (1 << SH) <= 64 && (1 << SH) <= C::RR_LIM, "shift conversion needs a small R'/R");
        FpR r;
#pragma unroll
        for (int i = 0; i < K; ++i) {
            const int bit = i * LB - SH; // limb i of (value << SH) = bits [bit, bit + LB) of value
            u64 x = 0;
            const int w = bit >= 0 ? (bit >> 5) : -1, o = bit >= 0 ? (bit & 31) : 0;
            if (bit >= 0) {
                x = w < C::N ? s.v[w] : 0;
                if (w + 1 < C::N) x |= (u64)s.v[w + 1] << 32;
                r.v[i] = (u32)(x >> o) & MASK;
            } else { // only limb 0: its low SH bits are zero
                r.v[i] = ((u32)s.v[0] << (-bit)) & MASK;
            }
        }
        return reduce<(1 << SH)>(r);
    }
    // value (< 4p, any representative) -> the canonical integer a, as 32-bit words (ark-ff into_repr)
    MG_DEV Std to_canonical() const {
        FpR one_plain = zero();
        one_plain.v[0] = 1;
        return mul(*this, one_plain).canonical_words();
    }
    // a value < 2p with normalised limbs -> fully reduced, repacked into 32-bit words
    MG_DEV Std canonical_words() const {
        FpR y = *this;
        int t[K];
#pragma unroll
        for (int i = 0; i < K; ++i) t[i] = (int)y.v[i] - (int)C::RR_P[i];
        int cy = 0;
        u32 d[K];
#pragma unroll
        for (int i = 0; i < K; ++i) {
            const int sum = t[i] + cy;
            cy = sum >> LB;
            d[i] = (u32)sum & MASK;
        }
        const bool ge = cy == 0;
#pragma unroll
        for (int i = 0; i < K; ++i) y.v[i] = ge ? d[i] : y.v[i];
        Std r;
#pragma unroll
        for (int w = 0; w < C::N; ++w) {
            u64 x = 0;
#pragma unroll
            for (int i = 0; i < K; ++i) {
                const int lo = i * LB - w * 32;
                if (lo > -LB && lo < 32) x |= lo >= 0 ? ((u64)y.v[i] << lo) : ((u64)y.v[i] >> (-lo));
            }
            r.v[w] = (u32)x;
        }
        return r;
    }
    // ---- conversions to / from the arkworks (32-bit limb, R = 2^(32N)) Montgomery form
    static MG_DEV FpR from_std(const Std &s) {
        Std c;
#pragma unroll
        for (int i = 0; i < C::N; ++i) c.v[i] = C::RR_TO[i];
        const Std t = Std::mul(s, c); // a*R * R' * R^-1 = a*R' mod p, canonical
        FpR r;
#pragma unroll
        for (int i = 0; i < K; ++i) {
            const int bit = i * LB, w = bit >> 5, o = bit & 31;
            u64 x = w < C::N ? t.v[w] : 0;
            if (w + 1 < C::N) x |= (u64)t.v[w + 1] << 32;
            r.v[i] = (u32)(x >> o) & MASK;
        }
        return r;
    }
    MG_DEV Std to_std() const {
        FpR c;
#pragma unroll
        for (int i = 0; i < K; ++i) c.v[i] = C::RR_FROM[i];
        FpR y = mul(*this, c); // a*R' * R * R'^-1 = a*R mod p, in [0, 2p)
        // one conditional subtraction of p
        int t[K];
#pragma unroll
        for (int i = 0; i < K; ++i) t[i] = (int)y.v[i] - (int)C::RR_P[i];
        int cy = 0;
        u32 d[K];
#pragma unroll
        for (int i = 0; i < K; ++i) {
            const int s = t[i] + cy;
            cy = s >> LB;
            d[i] = (u32)s & MASK;
        }
        const bool ge = cy == 0; // no final borrow: y >= p
#pragma unroll
        for (int i = 0; i < K; ++i) y.v[i] = ge ? d[i] : y.v[i];
        Std r;
#pragma unroll
        for (int w = 0; w < C::N; ++w) { // gather 32-bit words from the LB-bit limbs
            u64 x = 0;
#pragma unroll
            for (int i = 0; i < K; ++i) {
                const int lo = i * LB - w * 32; // position of limb i relative to word w
                if (lo > -LB && lo < 32) x |= lo >= 0 ? ((u64)y.v[i] << lo) : ((u64)y.v[i] >> (-lo));
            }
            r.v[w] = (u32)x;
        }
        return r;
    }

    static MG_DEV FpR load(const u32 *p) {
        FpR r;
#pragma unroll
        for (int i = 0; i < K; ++i) r.v[i] = p[i];
        return r;
    }
    MG_DEV void store(u32 *p) const {
#pragma unroll
        for (int i = 0; i < K; ++i) p[i] = v[i];
    }
    // 32 N-bit form of a NORMALISED value below 2^(32 N) (any lazily reduced value < 2p): the LB-bit limbs repacked into N
    // words, no arithmetic. What an NTT pass leaves in HBM for the next one: 32 B instead of 36 per element, and two adjacent
    // elements are one aligned 64 B sector where the 72 B of the limb form straddle two.
    MG_DEV void store_packed(u32 *p) const {
#pragma unroll
        for (int w = 0; w < C::N; ++w) {
            u64 x = 0;
#pragma unroll
            for (int i = 0; i < K; ++i) {
                const int lo = i * LB - w * 32; // position of limb i relative to word w
                if (lo > -LB && lo < 32) x |= lo >= 0 ? ((u64)v[i] << lo) : ((u64)v[i] >> (-lo));
            }
            p[w] = (u32)x;
        }
    }
    static MG_DEV FpR load_packed(const u32 *p) {
        u32 t[C::N];
#pragma unroll
        for (int w = 0; w < C::N; ++w) t[w] = p[w];
        FpR r;
#pragma unroll
        for (int i = 0; i < K; ++i) {
            const int bit = i * LB, w = bit >> 5, o = bit & 31;
            u64 x = w < C::N ? t[w] : 0;
            if (w + 1 < C::N) x |= (u64)t[w + 1] << 32;
            r.v[i] = (u32)(x >> o) & MASK;
        }
        return r;
    }
    // ---- memory format of AFFINE base coordinates (the records the accumulate kernels gather). A coordinate is < 2p with
    // normalised limbs, so it fits the 32 N-bit packed form: BN254 G1 = 2 x 32 B = ONE aligned 64 B sector per gathered point
    // where the 72 B limb form (96 B stride) touched two; the 18 v_alignbit/v_and of the unpacking are
    // 0.6 % of a mixed addition's instructions. MG_PACK_AFFINE: 0 = limb form everywhere (the round-1..5 layout), 1 (default) =
    // packed where the packed record is a whole number of 64 B sectors and smaller in sectors than the padded limb record
    // (BN254: 64 against 96 B, G2 128 against 160 B), 2 = BLS12-381 too (96 against 128 B: fewer bytes, the same two sectors).
#ifndef MG_PACK_AFFINE
#define MG_PACK_AFFINE 1
#endif
    static constexpr bool PACK_AFF = MG_PACK_AFFINE == 2 || (MG_PACK_AFFINE == 1 && C::N == 8);
    static constexpr int AFF_N = PACK_AFF ? C::N : K; // words per affine coordinate in memory
    static MG_DEV FpR load_aff(const u32 *p) {
        if constexpr (PACK_AFF) return load_packed(p);
        else return load(p);
    }
    MG_DEV void store_aff(u32 *p) const {
        if constexpr (PACK_AFF) store_packed(p);
        else store(p);
    }
    // limb i at p[i * stride]: LDS exchange areas keep one limb of all 64 lanes together (no bank conflicts)
    static MG_DEV FpR load_strided(const u32 *p, int stride) {
        FpR r;
#pragma unroll
        for (int i = 0; i < K; ++i) r.v[i] = p[i * stride];
        return r;
    }
    MG_DEV void store_strided(u32 *p, int stride) const {
#pragma unroll
        for (int i = 0; i < K; ++i) p[i * stride] = v[i];
    }
    // LDS exchange format of the cooperative addition: quads of limbs, one 16-byte access per lane and quad
    // (quad q of lane l at word (q*64 + l)*4): a 9-limb element is 3 ds_*_b128 instead of 9 ds_*_b32
    static constexpr int XQ = (K + 3) / 4;         // quads per element
    static constexpr int XWORDS = XQ * 64 * 4;     // words of one exchange slot (64 lanes)
    MG_DEV void store_x(u32 *slot, int lane) const {
#pragma unroll
        for (int q = 0; q < XQ; ++q) {
            uint4 t;
            t.x = v[4 * q];
            t.y = 4 * q + 1 < K ? v[4 * q + 1] : 0u;
            t.z = 4 * q + 2 < K ? v[4 * q + 2] : 0u;
            t.w = 4 * q + 3 < K ? v[4 * q + 3] : 0u;
            *reinterpret_cast<uint4 *>(slot + ((size_t)q * 64 + lane) * 4) = t;
        }
    }
    static MG_DEV FpR load_x(const u32 *slot, int lane) {
        FpR r;
#pragma unroll
        for (int q = 0; q < XQ; ++q) {
            const uint4 t = *reinterpret_cast<const uint4 *>(slot + ((size_t)q * 64 + lane) * 4);
            r.v[4 * q] = t.x;
            if (4 * q + 1 < K) r.v[4 * q + 1] = t.y;
            if (4 * q + 2 < K) r.v[4 * q + 2] = t.z;
            if (4 * q + 3 < K) r.v[4 * q + 3] = t.w;
        }
        return r;
    }
    static MG_DEV FpR select(bool c, const FpR &a, const FpR &b) {
        FpR r;
#pragma unroll
        for (int i = 0; i < K; ++i) r.v[i] = c ? a.v[i] : b.v[i];
        return r;
    }
    static MG_DEV FpR shfl(const FpR &a, int src_lane) {
        FpR r;
#pragma unroll
        for (int i = 0; i < K; ++i) r.v[i] = __shfl(a.v[i], src_lane, 64);
        return r;
    }
};

// Fp2 = FpR[u]/(u^2+1), components lazily reduced. Products are schoolbook (4 base products): the sum
// a0+a1 a Karatsuba product needs would double the operand bounds and overrun BN254's 7 bits of
// Montgomery headroom. Component bounds: product < 4p (c0 = a0 b0 + 2p - a1 b1, c1 = a0 b1 + a1 b0).
template <class C> struct Fp2R {
    typedef FpR<C> B;
    static constexpr int N = 2 * B::N;
    static constexpr bool EXT = true;
    static constexpr bool LAZY = true;
    static constexpr int BM = 4, LIM = B::LIM, MULK = 1, MAXM = B::MAXM, BX = 4, BY = 4, BRED = 2;
    typedef Fp2<C> Std;
    B c0, c1;
    static MG_DEV Fp2R zero() { return Fp2R{B::zero(), B::zero()}; }
    static MG_DEV Fp2R one() { return Fp2R{B::one(), B::zero()}; }
    MG_DEV bool is_zero_exact() const { return c0.is_zero_exact() && c1.is_zero_exact(); }
    template <int A> MG_DEV bool is_zero_mod() const { return c0.template is_zero_mod<A>() && c1.template is_zero_mod<A>(); }
    static MG_DEV Fp2R add(const Fp2R &a, const Fp2R &b) { return Fp2R{B::add(a.c0, b.c0), B::add(a.c1, b.c1)}; }
    static MG_DEV Fp2R dbl(const Fp2R &a) { return add(a, a); }
    template <int M> static MG_DEV Fp2R sub(const Fp2R &a, const Fp2R &b) {
        return Fp2R{B::template sub<M>(a.c0, b.c0), B::template sub<M>(a.c1, b.c1)};
    }
    template <int M> static MG_DEV Fp2R sub2(const Fp2R &a, const Fp2R &b, const Fp2R &c) {
        return Fp2R{B::template sub2<M>(a.c0, b.c0, c.c0), B::template sub2<M>(a.c1, b.c1, c.c1)};
    }
    template <int M> static MG_DEV Fp2R neg(const Fp2R &a) { return Fp2R{B::template neg<M>(a.c0), B::template neg<M>(a.c1)}; }
    template <int A> static MG_DEV Fp2R reduce(const Fp2R &a) {
        return Fp2R{B::template reduce<A>(a.c0), B::template reduce<A>(a.c1)};
    }
    // 14-limb products (BLS12-381 in rounds 1-3) were calls here: four of them inlined per Fp2 product overflowed the
    // register file in the group operations
    // (round 4: with 13 limbs the inlined form compiles -- 442 unified registers, one wavefront per SIMD like the calls build -- and
    // the BLS12-381 G2 accumulate kernel is 1.65x faster without the scratch traffic of the calls: 2^20 G2 MSM 24.0 -> 14.8 ms
    // uniform, 13.6 -> 8.1 ms witness-like; MG_FP2_CALLS restores the calls)
#ifdef MG_FP2_CALLS
    static constexpr bool CALLS = B::K > 9;
#else
    static constexpr bool CALLS = B::K > 13;
#endif
    static MG_DEV B bmul(const B &x, const B &y) {
        if constexpr (CALLS) return B::mul_call(x, y);
        else return B::mul(x, y);
    }
    static MG_DEV B bsqr(const B &x) {
        if constexpr (CALLS) return B::sqr_call(x);
        else return B::sqr(x);
    }
    static MG_DEV Fp2R mul(const Fp2R &a, const Fp2R &b) {
        const B v0 = bmul(a.c0, b.c0), v1 = bmul(a.c1, b.c1);
        const B x = bmul(a.c0, b.c1), y = bmul(a.c1, b.c0);
        return Fp2R{B::template sub<2>(v0, v1), B::add(x, y)};
    }
    // product of operands with component bounds A, B. Where the headroom allows the operand sums
    // (4 A B <= R'/p) this is Karatsuba -- 3 base products + one cheap reduction of c1 back below 2p -- else
    // the 4-product schoolbook form above. Either way the components of the result are < 4p.
    static MG_DEV B bmul_add(const B &x, const B &y, const B &z, const B &w) {
        if constexpr (CALLS) return B::mul_add_call(x, y, z, w);
        else return B::mul_add(x, y, z, w);
    }
    // Round 4: c0 = a0 b0 + a1 (M p - b1) and c1 = a0 b1 + a1 b0 as TWO FUSED products (FpR::mul_add: both double-width
    // products of a component share the column accumulators and ONE Montgomery reduction) -- 6 K^2 + 2 K multiply-adds, the
    // same as Karatsuba's three products (3 (2 K^2 + K)), but none of Karatsuba's linear work: two operand sums, the sum
    // v0 + v1, two subtractions and a reduce<6> (about 200 VALU instructions per product at K = 9) become ONE negation.
    // The G2 accumulate kernel issued 63 % non-multiply instructions (34 % of the multiply-add issue peak against 67 % for
    // G1: gpurun r4f). Components of the result are < 2p. Needs (A Bb + A Bb) <= LIM.
#ifndef MG_FP2_KARATSUBA // A/B builds: the round-1..3 Karatsuba / schoolbook forms
    template <int A, int Bb> static MG_DEV Fp2R mulb(const Fp2R &a, const Fp2R &b) {
#ifdef MG_FP2_NO_MULB
        if constexpr (false) {
#else
        if constexpr (2L * A * Bb <= LIM && (A <= MAXM || Bb <= MAXM)) {
#endif
            if constexpr (Bb <= A || A > MAXM) { // negate the component with the smaller bound (the smaller multiple of p)
                const B nb1 = B::template neg<Bb>(b.c1);
                return Fp2R{bmul_add(a.c0, b.c0, a.c1, nb1), bmul_add(a.c0, b.c1, a.c1, b.c0)};
            } else {
                const B na1 = B::template neg<A>(a.c1);
                return Fp2R{bmul_add(a.c0, b.c0, na1, b.c1), bmul_add(a.c0, b.c1, a.c1, b.c0)};
            }
        } else {
            return mulb_karatsuba<A, Bb>(a, b);
        }
    }
    // (a0 + a1)(a0 - a1) and 2 a0 a1: two products where the operand sums fit the headroom (4 A^2 <= LIM), else a0^2 - a1^2 as
    // one fused product; either way one reduction per component and a result < 2p
    template <int A> static MG_DEV Fp2R sqrb(const Fp2R &a) {
#ifdef MG_FP2_NO_SQRB
        return sqr(a);
#endif
        if constexpr (4L * A * A <= LIM && A <= MAXM) {
            const B s = B::add(a.c0, a.c1), d = B::template sub<A>(a.c0, a.c1);
            return Fp2R{bmul(s, d), bmul(B::dbl(a.c0), a.c1)};
        } else if constexpr (2L * A * A <= LIM && A <= MAXM) {
            return Fp2R{bmul_add(a.c0, a.c0, a.c1, B::template neg<A>(a.c1)), bmul(B::dbl(a.c0), a.c1)};
        } else {
            return sqr(a);
        }
    }
#else
    template <int A, int Bb> static MG_DEV Fp2R mulb(const Fp2R &a, const Fp2R &b) { return mulb_karatsuba<A, Bb>(a, b); }
    template <int A> static MG_DEV Fp2R sqrb(const Fp2R &a) { return sqr(a); }
#endif
    template <int A, int Bb> static MG_DEV Fp2R mulb_karatsuba(const Fp2R &a, const Fp2R &b) {
        if constexpr (4L * A * Bb <= LIM && 2 * A <= 16 && 2 * Bb <= 16) {
            const B v0 = bmul(a.c0, b.c0), v1 = bmul(a.c1, b.c1);
            const B s = bmul(B::add(a.c0, a.c1), B::add(b.c0, b.c1));
            const B c1 = B::template reduce<6>(B::template sub<4>(s, B::add(v0, v1))); // s + 4p - v0 - v1 < 6p
            return Fp2R{B::template sub<2>(v0, v1), c1};
        } else {
            return mul(a, b);
        }
    }
    static MG_DEV Fp2R sqr(const Fp2R &a) {
        const B v0 = bsqr(a.c0), v1 = bsqr(a.c1), x = bmul(a.c0, a.c1);
        return Fp2R{B::template sub<2>(v0, v1), B::dbl(x)};
    }
    static __device__ __noinline__ Fp2R inv(const Fp2R &a) { // one-off use only
        const B n = B::inv(B::add(B::sqr(a.c0), B::sqr(a.c1)));
        return Fp2R{B::mul(a.c0, n), B::template neg<2>(B::mul(a.c1, n))};
    }
    static MG_DEV Fp2R from_std(const Std &s) { return Fp2R{B::from_std(s.c0), B::from_std(s.c1)}; }
    MG_DEV Std to_std() const { return Std{c0.to_std(), c1.to_std()}; }
    static constexpr int XWORDS = 2 * B::XWORDS;
    MG_DEV void store_x(u32 *slot, int lane) const {
        c0.store_x(slot, lane);
        c1.store_x(slot + B::XWORDS, lane);
    }
    static MG_DEV Fp2R load_x(const u32 *slot, int lane) { return Fp2R{B::load_x(slot, lane), B::load_x(slot + B::XWORDS, lane)}; }
    static MG_DEV Fp2R load_strided(const u32 *p, int stride) {
        return Fp2R{B::load_strided(p, stride), B::load_strided(p + B::N * stride, stride)};
    }
    MG_DEV void store_strided(u32 *p, int stride) const {
        c0.store_strided(p, stride);
        c1.store_strided(p + B::N * stride, stride);
    }
    static constexpr int AFF_N = 2 * B::AFF_N;
    static MG_DEV Fp2R load_aff(const u32 *p) { return Fp2R{B::load_aff(p), B::load_aff(p + B::AFF_N)}; }
    MG_DEV void store_aff(u32 *p) const {
        c0.store_aff(p);
        c1.store_aff(p + B::AFF_N);
    }
    static MG_DEV Fp2R load(const u32 *p) { return Fp2R{B::load(p), B::load(p + B::N)}; }
    MG_DEV void store(u32 *p) const {
        c0.store(p);
        c1.store(p + B::N);
    }
    static MG_DEV Fp2R select(bool c, const Fp2R &a, const Fp2R &b) {
        return Fp2R{B::select(c, a.c0, b.c0), B::select(c, a.c1, b.c1)};
    }
    static MG_DEV Fp2R shfl(const Fp2R &a, int src_lane) { return Fp2R{B::shfl(a.c0, src_lane), B::shfl(a.c1, src_lane)}; }
};

} // namespace mg
