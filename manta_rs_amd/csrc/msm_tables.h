// Kernels around the MSM: base conversion, window / full tables, batched affine conversion, fixed-base multiplication, group NTT,
// element-wise group operations. Part of msm_impl.h.
#pragma once
#include "msm_common.h"

namespace mg {

// arkworks-format affine bases -> internal representation (identity copy when the two coincide)
template <class F>
__global__ __launch_bounds__(256) void bases_to_internal(const u32 *__restrict__ in, size_t n, u32 *__restrict__ out,
                                                         u32 astride) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    typedef typename F::Std S;
    const Affine<S> a = Affine<S>::load(in + i * Affine<S>::WORDS);
    Affine<F> r;
    if (a.is_inf()) {
        r.x = F::zero();
        r.y = F::zero();
    } else {
        r.x = F::from_std(a.x);
        r.y = F::from_std(a.y);
    }
    r.store(out + i * astride);
}

// --------------------------------------------------------------------------------------------
// precompute: table[w*n + i] = 2^(c w) * P_i (affine). Two kernels: doubling chains into XYZZ, then
// batched conversion to affine with Montgomery's trick (one Fermat inversion per KB points).
// --------------------------------------------------------------------------------------------
template <class F>
__global__ __launch_bounds__(256) void precompute_chain(const u32 *__restrict__ base, u32 astride, u32 n, int c, int W,
                                                        u32 *__restrict__ xyzz_out /* (W-1)*n */) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    XYZZ<F> p = XYZZ<F>::from_affine(Affine<F>::load(base + (size_t)i * astride));
    for (int w = 1; w < W; ++w) {
        for (int k = 0; k < c; ++k) p = XYZZ<F>::dbl(p);
        p.store(xyzz_out + ((size_t)(w - 1) * n + i) * XYZZ<F>::WORDS);
    }
}
// full tables: the multiples m Q, m = 1 .. B, of `cnt` window bases Q = 2^(c w) P (affine, from entry j0 on) as XYZZ points,
// entry (t B + m - 1) -- B - 1 mixed additions per lane (the first is the doubling Q + Q: madd's exact exceptional cases)
template <class F>
__global__ __launch_bounds__(256) void full_table_chain(const u32 *__restrict__ win, u32 astride, size_t j0, u32 cnt, u32 B,
                                                        u32 *__restrict__ xyzz_out) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= cnt) return;
    const Affine<F> q = Affine<F>::load(win + (j0 + t) * astride);
    XYZZ<F> acc = XYZZ<F>::from_affine(q);
    for (u32 m = 0; m < B; ++m) {
        if (m) acc.madd(q, false);
        acc.store(xyzz_out + ((size_t)t * B + m) * XYZZ<F>::WORDS);
    }
}
template <class F> struct FieldInv; // Fermat inversion on the device (slow, one-off use only)
template <class C> struct FieldInv<Fp<C>> {
    static __device__ Fp<C> inv(const Fp<C> &a) { return Fp<C>::inv(a); }
};
template <class C> struct FieldInv<FpR<C>> {
    static __device__ FpR<C> inv(const FpR<C> &a) { return FpR<C>::inv(a); }
};
template <class C> struct FieldInv<Fp2R<C>> {
    static __device__ Fp2R<C> inv(const Fp2R<C> &a) { return Fp2R<C>::inv(a); }
};
template <class C> struct FieldInv<Fp2<C>> {
    static __device__ Fp2<C> inv(const Fp2<C> &a) {
        typedef Fp<C> B;
        B n = B::inv(B::add(B::sqr(a.c0), B::sqr(a.c1)));
        return Fp2<C>{B::mul(a.c0, n), B::neg(B::mul(a.c1, n))};
    }
};
template <class F, int KB>
__global__ __launch_bounds__(256) void xyzz_to_affine_batch(const u32 *__restrict__ xyzz, size_t n,
                                                            u32 *__restrict__ aff, u32 astride) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t b0 = t * KB;
    if (b0 >= n) return;
    // prefix products of d_k = zz*zzz (1 for infinity) are parked at the front of the affine output record (F::N limb words: a
    // record is 2 F::AFF_N >= F::N words in either format), then replaced by the record itself
    static_assert(Affine<F>::WORDS >= F::N, "record too small to park a field element");
    F run = F::one();
    for (int k = 0; k < KB && b0 + k < n; ++k) {
        const u32 *src = xyzz + (b0 + k) * XYZZ<F>::WORDS;
        F zz = F::load(src + 2 * F::N), zzz = F::load(src + 3 * F::N);
        F d = zz.is_zero_exact() ? F::one() : F::mul(zz, zzz);
        run.store(aff + (b0 + k) * astride); // prefix before k
        run = F::mul(run, d);
    }
    F inv = FieldInv<F>::inv(run);
    int last = KB - 1;
    if (b0 + KB > n) last = (int)(n - b0) - 1;
    for (int k = last; k >= 0; --k) {
        const u32 *src = xyzz + (b0 + k) * XYZZ<F>::WORDS;
        u32 *dst = aff + (b0 + k) * astride;
        F zz = F::load(src + 2 * F::N), zzz = F::load(src + 3 * F::N);
        if (zz.is_zero_exact()) {
            Affine<F>{F::zero(), F::zero()}.store(dst);
            continue;
        }
        F pre = F::load(dst);
        F dinv = F::mul(inv, pre); // 1/(zz*zzz)
        inv = F::mul(inv, F::mul(zz, zzz));
        F x = F::load(src), y = F::load(src + F::N);
        F izz = F::mul(dinv, zzz), izzz = F::mul(dinv, zz);
        Affine<F>{F::mul(x, izz), F::mul(y, izzz)}.store(dst); // products: < 2p, normalised -- what the packed format needs
    }
}

// [k_i] * base, k canonical; output XYZZ (converted by xyzz_to_affine_batch)
template <class F>
__global__ __launch_bounds__(256) void fixed_base_mul_kernel(const u32 *__restrict__ base_aff,
                                                             const u32 *__restrict__ scalars, size_t n,
                                                             u32 *__restrict__ out_xyzz) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Affine<F> b = Affine<F>::load(base_aff);
    XYZZ<F> acc = XYZZ<F>::inf();
    for (int limb = 7; limb >= 0; --limb) {
        const u32 w = scalars[i * 8 + limb];
        for (int bit = 31; bit >= 0; --bit) {
            acc = XYZZ<F>::dbl(acc);
            if ((w >> bit) & 1) acc.madd(b, false);
        }
    }
    acc.store(out_xyzz + i * XYZZ<F>::WORDS);
}

// Windowed fixed-base multiplication (key generation: every element of a Groth16 key is a multiple of a generator;
// ark-groth16 generate_parameters uses FixedBaseMSM the same way). Table T[w][d-1] = d * 2^(8w) * B for 32 windows of 8
// bits, d = 1..255 (8160 affine points, ~0.5 / 0.8 MB in G1: L2-resident); [k]B = at most 32 mixed additions, no doubling.
template <class F>
__global__ __launch_bounds__(256) void fixed_base_table_kernel(const u32 *__restrict__ base_aff, u32 *__restrict__ out_xyzz) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 32 * 255) return;
    const u32 w = t / 255, d = t % 255 + 1;
    const Affine<F> b = Affine<F>::load(base_aff);
    XYZZ<F> acc = XYZZ<F>::inf();
    for (int bit = 7; bit >= 0; --bit) { // d * B
        acc = XYZZ<F>::dbl(acc);
        if ((d >> bit) & 1) acc.madd(b, false);
    }
    for (u32 k = 0; k < 8 * w; ++k) acc = XYZZ<F>::dbl(acc); // * 2^(8w)
    acc.store(out_xyzz + (size_t)t * XYZZ<F>::WORDS);
}
template <class F>
__global__ __launch_bounds__(256) void fixed_base_mul_table_kernel(const u32 *__restrict__ table_aff, const u32 *__restrict__ scalars,
                                                                   size_t n, u32 *__restrict__ out_xyzz) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    XYZZ<F> acc = XYZZ<F>::inf();
    for (int limb = 0; limb < 8; ++limb) {
        const u32 s = scalars[i * 8 + limb];
#pragma unroll 1
        for (int k = 0; k < 4; ++k) {
            const u32 d = (s >> (8 * k)) & 255u;
            if (d) acc.madd(Affine<F>::load(table_aff + ((size_t)(limb * 4 + k) * 255 + d - 1) * Affine<F>::WORDS), false);
        }
    }
    acc.store(out_xyzz + i * XYZZ<F>::WORDS);
}

// Radix-2 NTT over GROUP elements (`Radix2EvaluationDomain::{fft, ifft}` applied to a vector of points:
// manta-trusted-setup/src/groth16/mpc.rs:378-381 turns powers of tau into the Lagrange basis this way). One butterfly
// per lane and stage on XYZZ points in HBM: t = [w] b (double-and-add, w canonical from the Fr twiddle table),
// a' = a + t, b' = a - t. Input in bit-reversed order, output natural (decimation in time).
template <class F, class FrC>
__global__ __launch_bounds__(256) void group_ntt_stage_kernel(u32 *__restrict__ pts, const u32 *__restrict__ tw_mont, unsigned lg,
                                                              unsigned s) {
    const u32 k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= (1u << (lg - 1))) return;
    const u32 half = 1u << (s - 1), j = k & (half - 1), g = k >> (s - 1);
    const size_t i0 = ((size_t)g << s) | j, i1 = i0 + half;
    constexpr size_t XW = XYZZ<F>::WORDS;
    XYZZ<F> a = XYZZ<F>::load(pts + i0 * XW);
    const XYZZ<F> b = XYZZ<F>::load(pts + i1 * XW);
    XYZZ<F> t = b;
    if (s > 1) { // twiddle w_n^(j * n / 2^s); stage 1 has w = 1
        const Fp<FrC> wc = Fp<FrC>::from_mont(Fp<FrC>::load(tw_mont + ((size_t)j << (lg - s)) * 8));
        t = XYZZ<F>::inf();
        for (int limb = 7; limb >= 0; --limb) {
            u32 w = 0;
#pragma unroll
            for (int q = 0; q < 8; ++q) w = (q == limb) ? wc.v[q] : w;
            for (int bit = 31; bit >= 0; --bit) {
                t = XYZZ<F>::dbl(t);
                if ((w >> bit) & 1) t.add(b);
            }
        }
    }
    XYZZ<F> d = a;
    a.add(t);
    if (!t.is_inf()) {
        t.y = b_neg(bv<XYZZ<F>::BY>(t.y)).v;
        d.add(t);
    }
    a.store(pts + i0 * XW);
    d.store(pts + i1 * XW);
}
// affine (arkworks format) -> XYZZ internal at the bit-reversed position; and XYZZ internal -> scaled by a scalar -> std XYZZ
template <class F>
__global__ __launch_bounds__(256) void group_ntt_load_kernel(const u32 *__restrict__ in_aff, unsigned lg, u32 *__restrict__ out) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (1u << lg)) return;
    typedef typename F::Std S;
    const u32 j = lg ? (__brev(i) >> (32 - lg)) : 0;
    const Affine<S> s = Affine<S>::load(in_aff + (size_t)j * Affine<S>::WORDS);
    XYZZ<F> p = XYZZ<F>::inf();
    if (!s.is_inf()) p = XYZZ<F>{F::from_std(s.x), F::from_std(s.y), F::one(), F::one()};
    p.store(out + (size_t)i * XYZZ<F>::WORDS);
}
template <class F>
__global__ __launch_bounds__(256) void group_scale_store_kernel(const u32 *__restrict__ pts, const u32 *__restrict__ scalar_canon,
                                                                size_t n, u32 *__restrict__ out_xyzz_std) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    typedef typename F::Std S;
    XYZZ<F> p = XYZZ<F>::load(pts + i * XYZZ<F>::WORDS);
    if (scalar_canon) { // ifft: times n^-1
        const XYZZ<F> b = p;
        p = XYZZ<F>::inf();
        for (int limb = 7; limb >= 0; --limb) {
            const u32 w = scalar_canon[limb];
            for (int bit = 31; bit >= 0; --bit) {
                p = XYZZ<F>::dbl(p);
                if ((w >> bit) & 1) p.add(b);
            }
        }
    }
    p.store_std(out_xyzz_std + i * XYZZ<S>::WORDS);
}

// Element-wise group operations on arrays of affine points (arkworks format in, XYZZ in arkworks format out,
// normalised by xyzz_to_affine_batch) computed with the MSM kernels' own device functions in their internal
// field representation -- the primitive menu of manta-benchmark/src/ecc.rs:30-128 (mixed add :69-74, projective
// add :78-83, scalar multiplication :87-101, batch normalisation :114-119) as a parity-test surface.
//   op 0: P + Q via madd (projective += affine)      op 1: P + Q via the general add (projective += projective)
//   op 2: 2P                                          op 3: [k]P, k = 4 x u64 canonical (double-and-add over madd)
//   op 4: P - Q via madd with the negate flag            op 5: [k]P with ONE scalar k for all points (`batch_mul_fixed_scalar`,
//                                                              manta-trusted-setup/src/util.rs:440-445): uniform control flow
//   op 6 (internal, ec_mul_xyzz_begin): [k1 + lambda k2]P, 64-bit k1 and k2, through the endomorphism (G1 only)
template <class F>
__global__ __launch_bounds__(256) void ec_elementwise_kernel(int op, const u32 *__restrict__ a, const u32 *__restrict__ b,
                                                             size_t n, u32 *__restrict__ out_xyzz_std) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    typedef typename F::Std S;
    auto load_affine = [](const u32 *p) {
        const Affine<S> s = Affine<S>::load(p);
        Affine<F> r;
        if (s.is_inf()) {
            r.x = F::zero();
            r.y = F::zero();
        } else {
            r.x = F::from_std(s.x);
            r.y = F::from_std(s.y);
        }
        return r;
    };
    const Affine<F> pa = load_affine(a + i * Affine<S>::WORDS);
    XYZZ<F> acc = XYZZ<F>::from_affine(pa);
    if (op == 0 || op == 4) {
        acc.madd(load_affine(b + i * Affine<S>::WORDS), op == 4);
    } else if (op == 1) {
        acc.add(XYZZ<F>::from_affine(load_affine(b + i * Affine<S>::WORDS)));
    } else if (op == 2) {
        acc = XYZZ<F>::dbl(acc);
    } else if (op == 6) {
        // [k1 + lambda k2] P with 64-bit k1, k2 (the low two u64 of the lane's scalar) through the curve's endomorphism
        // phi(x, y) = (beta x, y) = lambda (x, y): ONE chain of 64 doublings with additions of P, phi(P) or P + phi(P) --
        // the general addition on a table entry picked by selects, so that every lane runs the same instruction stream
        // (128 doublings + 64 mixed additions for a 128-bit multiplier otherwise). beta: arkworks-format words behind the
        // n scalars in b. The batch verifier's random coefficients (verify.cpp).
        const S beta_std = S::load(b + n * 8);
        const F beta = F::from_std(beta_std);
        const XYZZ<F> t1 = acc;
        XYZZ<F> t2 = acc;
        if (!t1.is_inf()) t2.x = (bv<F::BM>(pa.x) * bv<F::BM>(beta)).v;
        XYZZ<F> t3 = t1;
        t3.add(t2);
        const u64 k1 = (u64)b[i * 8] | ((u64)b[i * 8 + 1] << 32), k2 = (u64)b[i * 8 + 2] | ((u64)b[i * 8 + 3] << 32);
        acc = XYZZ<F>::inf();
        for (int bit = 63; bit >= 0; --bit) {
            acc = XYZZ<F>::dbl(acc);
            const int sel = (int)((k1 >> bit) & 1) | ((int)((k2 >> bit) & 1) << 1);
            XYZZ<F> o;
            o.x = F::select(sel == 3, t3.x, F::select(sel == 2, t2.x, t1.x));
            o.y = F::select(sel == 3, t3.y, t1.y); // (phi keeps y)
            o.zz = F::select(sel == 3, t3.zz, t1.zz);
            o.zzz = F::select(sel == 3, t3.zzz, t1.zzz);
            if (sel) acc.add(o);
        }
    } else {
        acc = XYZZ<F>::inf();
        const size_t si = op == 5 ? 0 : i; // op 5: every lane reads the same scalar
        for (int limb = 7; limb >= 0; --limb) {
            const u32 w = b[si * 8 + limb];
            for (int bit = 31; bit >= 0; --bit) {
                acc = XYZZ<F>::dbl(acc);
                if ((w >> bit) & 1) acc.madd(pa, false);
            }
        }
    }
    acc.store_std(out_xyzz_std + i * XYZZ<S>::WORDS);
}

// per-thread partial sums of affine points (strided), output XYZZ partials
template <class F>
__global__ __launch_bounds__(256) void sum_affine_kernel(const u32 *__restrict__ pts, size_t n, u32 T,
                                                         u32 *__restrict__ out) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    XYZZ<F> acc = XYZZ<F>::inf();
    for (size_t i = t; i < n; i += T) acc.madd(Affine<F>::load(pts + i * Affine<F>::WORDS), false);
    acc.store(out + (size_t)t * XYZZ<F>::WORDS);
}

} // namespace mg
