// Scalar-field kernels on gfx950: radix-2 NTT, CSR SpMV, QAP pointwise step. Templated on the Fr
// parameters; instantiated in fr_bn254.hip / fr_bls381.hip.
//
// Replaces: ark-poly ^0.3.0 Radix2EvaluationDomain::{fft,ifft,coset_fft,
// coset_ifft}_in_place and ark-groth16 ^0.3.0 R1CStoQAP::witness_map's evaluate_constraint /
// mul_polynomials_in_evaluation_domain / divide_by_vanishing_poly_on_coset (reached from
// manta-crypto/src/arkworks/groth16.rs:597). Conventions restated from SURVEY.md App. A.1/B.3:
// omega_D = omega_{2^s}^(2^(s - log D)); ifft scales by D^-1; coset shift g = Fr multiplicative
// generator; natural order in and out.
#pragma once
#include "fp_dev.h"
#include "tuning.h"
#include "fpr_dev.h"
#include "host_ec.h"
#include "params_gen.h"
#include "prover.h"
#include <algorithm>
#include <type_traits>
#include <map>
#include <mutex>

namespace mg {

// out[i] = base^i for i < n, given sq[k] = base^(2^k) (device array of `bits` elements)
template <class FrC>
__global__ __launch_bounds__(256) void powers_kernel(u32 *__restrict__ out, const u32 *__restrict__ sq, u32 n,
                                                     int bits, const u32 *__restrict__ scale /* 8 words or null */) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    typedef Fp<FrC> F;
    F acc = scale ? F::load(scale) : F::one();
    for (int k = 0; k < bits; ++k)
        if ((i >> k) & 1) acc = F::mul(acc, F::load(sq + 8 * k));
    acc.store(out + (size_t)i * 8);
}

// ---------------------------------------------------------------------------------------------------
// Reduced-radix scalar field (round 2). Inside the NTT / witness-map pipeline an Fr element is FpR<FrC>: 9 limbs of
// 29 bits, Montgomery radix R' = 2^261 = 2^5 R, lazily reduced (fpr_dev.h) -- the butterfly's product is 171 carry-free
// v_mad_u64_u32 instead of 136 multiply-adds + 136 carry additions, additions are limb-wise with one normalisation
// pass. Elements are 9 words (36 B) in HBM and in LDS. Conversions at the ends cost no multiplication going in (the
// arkworks form a*R becomes a*R' by a 5-bit shift folded into the limb repacking + one reduce, from_std_shift) and
// one product coming out (to_std / to_canonical). Tables (twiddles, scale tables) hold canonical values (< p).
//
// LDS-fused NTT pass: `ns` consecutive butterfly stages (global stages s0 .. s0+ns-1) of up to three
// vectors (blockIdx.y) in one launch. A workgroup owns the 2^ns elements that differ only in index bits
// [s0-1, s0-1+ns), for 2^cb adjacent values of the low bits; the tile lives in LDS as limb-major SoA
// (conflict-free ds_read_b32) and is read from / written to HBM exactly once per pass.
//   DIT (DIF = false): stages ascending, input bit-reversed -> output natural (Cooley-Tukey)
//   DIF (DIF = true) : stages descending, input natural -> output bit-reversed (Gentleman-Sande)
// Bounds (multiples of p; uniform per stage, tracked in `B`): tile values enter < 4p. DIT: t = b w < 2p, a' = a + t,
// b' = a + 2p - t, so B grows by 2 per stage (<= 4 + 5 + 2*9 = 27 over ten stages, within LIM = 70 / 169 for the only
// product, b w) and one reduce at the end of the pass brings everything below 2p. DIF: x = a + b doubles B, so x is
// reduced whenever 2B would exceed 8; y = (a + 9p - b) w < 2p. `post` (optional, last pass) multiplies element i by
// post[i] on the way out (n^-1 and coset powers, tabulated in the order the pass leaves the data in).
// The public transform (mg_ntt / mg_ntt_device: arkworks format in and out) folds its format conversions into its first and
// last pass instead of running them as kernels of their own (27 + 17 us of a 218 us 2^20 transform): with `in_std` the tile
// is gathered from the arkworks-format input at the bit-reversed index and converted on the way into LDS (times pre[j] for
// the forward coset transform); with `out_std` the finished tile is scaled (n^-1 of the plain inverse transform), converted
// back and written in the arkworks format. Both null inside the witness map, which stays in the work form throughout.
// the butterfly's product: the single-chain coding of fpr_dev.h (mad_chain_*: every multiply-add of a column in one dependent
// chain, no 64-bit joins) -- 112 instead of 131 VGPRs in the register kernel (four wavefronts per SIMD) and -10 % on the 2^20
// transform (profiles/r04_ntt_register_stages.txt); MG_NTT_TWO_CHAINS restores the compiler's coding for re-measurement
#ifdef MG_NTT_TWO_CHAINS
#define MG_NTT_MUL(x, y) R::mul(x, y)
#else
#define MG_NTT_MUL(x, y) R::template mul_t<true>(x, y)
#endif
#define MG_NTT_MUL_RR(x, y) MG_NTT_MUL(x, y) // (the stage-per-round-trip kernel: 82 -> 64 VGPRs, DIF passes -4 %)
// The witness map is the head of the chain that bounds a single proof (witness map -> h MSM) and runs beside the accumulate kernels
// of the other MSMs: its wavefronts ask for issue priority 2 -- above the accumulate kernels (0), below the MSMs' tail kernels (3).
// Sequential PrivateTransfer proofs, sparse / W / dense, library variants alternating on one box (profiles/r05_single_proof_ab.txt):
// level 0: 0.728 / 0.872 / 1.290 ms, 1: 0.714 / 0.867 / 1.253, 2: 0.707 / 0.864 / 1.249, 3: 0.716 / 0.925 / 1.245 (level 3 takes issue
// slots from the a | b_g1 | l and G2 chains' own tails: W +6 % -- what round 3 saw when it tried level 3 and dropped it);
// batched passes do not move. MG_WM_PRIO = the s_setprio level at build time (0: off).
#ifndef MG_WM_PRIO
#define MG_WM_PRIO 2
#endif
#if MG_WM_PRIO
#define MG_PRIO_WM() __builtin_amdgcn_s_setprio(MG_WM_PRIO)
#else
#define MG_PRIO_WM() ((void)0)
#endif
struct NttIo {
    const u32 *in_std, *pre_rr;
    u32 *out_std;
    const u32 *scale_rr;
    // the work vector between two passes of the public transform (a scratch vector, not the caller's) in the packed 32 B form
    // (FpR::store_packed): bit 0 = this pass reads it, bit 1 = this pass writes it. A strided pass moves two adjacent elements
    // per butterfly column: 64 B = one aligned sector instead of 72 B across two (2^20: passes 77.6 + 101 -> 75 + 97 us)
    u32 packed;
};
template <class FrC, bool DIF>
__global__ __launch_bounds__(1024) void ntt_pass_rr(u32 *__restrict__ d0, u32 *__restrict__ d1, u32 *__restrict__ d2,
                                                   const u32 *__restrict__ tw, unsigned lg, unsigned s0, unsigned ns,
                                                   unsigned cb, const u32 *__restrict__ post, NttIo io) {
    MG_PRIO_WM();
    extern __shared__ __attribute__((aligned(16))) u32 sm[];
    typedef FpR<FrC> R;
    constexpr int K = R::K;
    static_assert(K == 9 && R::LIM >= 64, "bound analysis above");
    u32 *__restrict__ data = (blockIdx.y == 0 ? d0 : (blockIdx.y == 1 ? d1 : d2)) + ((size_t)blockIdx.z << lg) * K;
    const u32 E = 1u << ns, TOT = E << cb, CM = (1u << cb) - 1;
    const u32 lo_bits = s0 - 1;
    const u32 nlo = (1u << lo_bits) >> cb;
    const u32 lo_base = (blockIdx.x % nlo) << cb, hi = blockIdx.x / nlo;
    const size_t base = ((size_t)hi << (lo_bits + ns)) + lo_base;
    // Round 5, single-column tiles (cb = 0: the passes of ONE proof's witness map, a chain of six launches with ~1.5 wavefronts per
    // SIMD): the tile's 2^ns - 1 twiddles -- entry 2^(tl-1) - 1 + jl for local stage tl -- are fetched ONCE, next to the tile, into
    // LDS behind it. Before, every stage waited for its own 48 B gather from a table that the accumulate kernels' GB of table
    // traffic keep out of the L2: eight dependent memory latencies per pass (io.packed bit 2; MANTA_NTT_TWL=0 turns it off).
    const bool twl = (io.packed & 4u) != 0;
    u32 *__restrict__ stw = sm + (size_t)K * TOT;
    if (twl) {
        for (u32 x = threadIdx.x + 1; x < E; x += blockDim.x) {
            const unsigned tl = 32u - (unsigned)__clz(x);
            const unsigned s = s0 + tl - 1;
            if (s > 1) {
                const size_t j = ((size_t)(x - (1u << (tl - 1))) << lo_bits) | lo_base;
                const uint4 *q = reinterpret_cast<const uint4 *>(tw + (j << (lg - s)) * 12);
                const uint4 q0 = q[0], q1 = q[1], q2 = q[2];
                u32 *o = stw + (x - 1);
                o[0 * E] = q0.x, o[1 * E] = q0.y, o[2 * E] = q0.z, o[3 * E] = q0.w, o[4 * E] = q1.x, o[5 * E] = q1.y, o[6 * E] = q1.z,
                o[7 * E] = q1.w, o[8 * E] = q2.x;
            }
        }
    }
    for (u32 t = threadIdx.x; t < TOT; t += blockDim.x) {
        const size_t gi = base + ((size_t)(t >> cb) << lo_bits) + (t & CM);
        if (io.in_std) { // arkworks format at the bit-reversed index -> work form (< 2p)
            const u32 j = __brev((u32)gi) >> (32 - lg);
            R r = R::from_std_shift(Fp<FrC>::load(io.in_std + (size_t)j * 8));
            if (io.pre_rr) r = R::mul(r, R::load(io.pre_rr + (size_t)j * K));
#pragma unroll
            for (int l = 0; l < K; ++l) sm[l * TOT + t] = r.v[l];
            continue;
        }
        if (io.packed & 1u) {
            const R r = R::load_packed(data + gi * 8);
#pragma unroll
            for (int l = 0; l < K; ++l) sm[l * TOT + t] = r.v[l];
            continue;
        }
        const u32 *p = data + gi * K;
#pragma unroll
        for (int l = 0; l < K; ++l) sm[l * TOT + t] = p[l];
    }
    __syncthreads();
    int B = 4; // every value in the tile is < B*p
    for (unsigned st = 0; st < ns; ++st) {
        const unsigned tl = DIF ? ns - st : st + 1; // local stage 1..ns
        const unsigned s = s0 + tl - 1;             // global stage
        const u32 half = 1u << (tl - 1);
        const bool has_w = s > 1;
        const bool red = DIF && 2 * B > 8; // uniform
        for (u32 k = threadIdx.x; k < TOT / 2; k += blockDim.x) {
            const u32 c = k & CM, kk = k >> cb;
            const u32 jl = kk & (half - 1), g = kk >> (tl - 1);
            const u32 i0 = ((((g << tl) | jl)) << cb) + c, i1 = i0 + (half << cb);
            R a, b;
#pragma unroll
            for (int l = 0; l < K; ++l) a.v[l] = sm[l * TOT + i0], b.v[l] = sm[l * TOT + i1];
            R w;
            if (has_w && twl) {
                const u32 *o = stw + (half - 1 + jl);
#pragma unroll
                for (int l = 0; l < K; ++l) w.v[l] = o[l * E];
            } else if (has_w) {
                const size_t j = ((size_t)jl << lo_bits) | (lo_base + c);
                const uint4 *q = reinterpret_cast<const uint4 *>(tw + (j << (lg - s)) * 12); // 48 B records
                const uint4 q0 = q[0], q1 = q[1], q2 = q[2];
                w.v[0] = q0.x, w.v[1] = q0.y, w.v[2] = q0.z, w.v[3] = q0.w, w.v[4] = q1.x, w.v[5] = q1.y, w.v[6] = q1.z,
                w.v[7] = q1.w, w.v[8] = q2.x;
            }
            R x, y;
            if (DIF) {
                x = R::add(a, b);                    // < 2B p
                y = R::template sub<9>(a, b);        // a + 9p - b (b < 8p) < (B + 9) p
                if (has_w) y = MG_NTT_MUL_RR(y, w);  // < 2p
                else y = R::template reduce<17>(y);  // last DIF stage (w = 1)
                if (red) x = R::template reduce<16>(x);
            } else {
                if (has_w) {
                    b = MG_NTT_MUL_RR(b, w); // < 2p
                    x = R::add(a, b);
                    y = R::template sub<2>(a, b);
                } else { // first DIT stage (w = 1): b < 4p
                    x = R::add(a, b);
                    y = R::template sub<5>(a, b);
                }
            }
#pragma unroll
            for (int l = 0; l < K; ++l) sm[l * TOT + i0] = x.v[l], sm[l * TOT + i1] = y.v[l];
        }
        if (DIF) B = red ? 2 : 2 * B;
        else B = has_w ? B + 2 : B + 5;
        __syncthreads();
    }
    for (u32 t = threadIdx.x; t < TOT; t += blockDim.x) {
        const size_t gi = base + ((size_t)(t >> cb) << lo_bits) + (t & CM);
        R v;
#pragma unroll
        for (int l = 0; l < K; ++l) v.v[l] = sm[l * TOT + t];
        if (post) v = R::mul(v, R::load(post + gi * K)); // any B <= 27 times a canonical table entry: < 2p
        else if (DIF ? B > 4 : true) v = R::template reduce<32>(v);
        if (io.out_std) { // work form (< 4p) -> arkworks format, times the constant of the plain inverse transform first
            if (io.scale_rr) v = R::mul(v, R::load(io.scale_rr));
            v.to_std().store(io.out_std + gi * 8);
            continue;
        }
        if (io.packed & 2u) { // (v < 2p and normalised: a product or a reduce made it)
            v.store_packed(data + gi * 8);
            continue;
        }
        u32 *p = data + gi * K;
#pragma unroll
        for (int l = 0; l < K; ++l) p[l] = v.v[l];
    }
}

// ---------------------------------------------------------------------------------------------------
// Round 5: pass FUSION for the witness map of ONE proof. With single-column tiles the low pass of the inverse
// transform and the low pass of the forward coset transform that follows it work on the SAME contiguous tile, and so do the high pass
// of the forward transform, the pointwise step and the high pass of the last inverse transform on the same strided tile: 7 launches
// (2 + 2 + pointwise + 2) become 4, three LDS fill / drain pairs and three trips through HBM of a | b | c go away -- on a chain whose
// every launch is latency (six launches of 30-45 us at ~1.5 wavefronts per SIMD).
// ntt_tile_stages = the stage loop of ntt_pass_rr at cb = 0 over one tile already in LDS (limb-major, stride E).
template <class FrC, bool DIF>
MG_DEV void ntt_tile_stages(u32 *__restrict__ sm, u32 E, unsigned ns, unsigned s0, unsigned lg, u32 lo_bits, u32 lo_base,
                            const u32 *__restrict__ tw, int &B, u32 tid, u32 nthr) {
    typedef FpR<FrC> R;
    constexpr int K = R::K;
    for (unsigned st = 0; st < ns; ++st) {
        const unsigned tl = DIF ? ns - st : st + 1; // local stage 1..ns
        const unsigned s = s0 + tl - 1;             // global stage
        const u32 half = 1u << (tl - 1);
        const bool has_w = s > 1;
        const bool red = DIF && 2 * B > 8; // uniform
        for (u32 k = tid; k < E / 2; k += nthr) {
            const u32 jl = k & (half - 1), g = k >> (tl - 1);
            const u32 i0 = (g << tl) | jl, i1 = i0 + half;
            R a, b;
#pragma unroll
            for (int l = 0; l < K; ++l) a.v[l] = sm[l * E + i0], b.v[l] = sm[l * E + i1];
            R w;
            if (has_w) {
                const size_t j = ((size_t)jl << lo_bits) | lo_base;
                const uint4 *q = reinterpret_cast<const uint4 *>(tw + (j << (lg - s)) * 12); // 48 B records
                const uint4 q0 = q[0], q1 = q[1], q2 = q[2];
                w.v[0] = q0.x, w.v[1] = q0.y, w.v[2] = q0.z, w.v[3] = q0.w, w.v[4] = q1.x, w.v[5] = q1.y, w.v[6] = q1.z,
                w.v[7] = q1.w, w.v[8] = q2.x;
            }
            R x, y;
            if (DIF) {
                x = R::add(a, b);                    // < 2B p
                y = R::template sub<9>(a, b);        // a + 9p - b (b < 8p)
                if (has_w) y = MG_NTT_MUL_RR(y, w);  // < 2p
                else y = R::template reduce<17>(y);  // last DIF stage (w = 1)
                if (red) x = R::template reduce<16>(x);
            } else {
                if (has_w) {
                    b = MG_NTT_MUL_RR(b, w); // < 2p
                    x = R::add(a, b);
                    y = R::template sub<2>(a, b);
                } else { // first DIT stage (w = 1): b < 4p
                    x = R::add(a, b);
                    y = R::template sub<5>(a, b);
                }
            }
#pragma unroll
            for (int l = 0; l < K; ++l) sm[l * E + i0] = x.v[l], sm[l * E + i1] = y.v[l];
        }
        if (DIF) B = red ? 2 : 2 * B;
        else B = has_w ? B + 2 : B + 5;
        __syncthreads();
    }
}
// low pass of the inverse transform (DIF stages ns .. 1, then the table `post`) + low pass of the forward transform (DIT stages 1 .. ns)
// on one contiguous tile of 2^ns elements; grid (n >> ns, vectors, batch), E / 2 threads
template <class FrC>
__global__ __launch_bounds__(512) void ntt_fused_low(u32 *__restrict__ d0, u32 *__restrict__ d1, u32 *__restrict__ d2,
                                                     const u32 *__restrict__ tw_inv, const u32 *__restrict__ tw_fwd, unsigned lg,
                                                     unsigned ns, const u32 *__restrict__ post) {
    MG_PRIO_WM();
    extern __shared__ __attribute__((aligned(16))) u32 sm[];
    typedef FpR<FrC> R;
    constexpr int K = R::K;
    static_assert(K == 9 && R::LIM >= 64, "bound analysis of ntt_pass_rr");
    u32 *__restrict__ data = (blockIdx.y == 0 ? d0 : (blockIdx.y == 1 ? d1 : d2)) + ((size_t)blockIdx.z << lg) * K;
    const u32 E = 1u << ns;
    const size_t base = (size_t)blockIdx.x << ns;
    for (u32 t = threadIdx.x; t < E; t += blockDim.x) {
        const u32 *p = data + (base + t) * K;
#pragma unroll
        for (int l = 0; l < K; ++l) sm[l * E + t] = p[l];
    }
    __syncthreads();
    int B = 4; // every value of the tile is < B p (the high pass left < 4p)
    ntt_tile_stages<FrC, true>(sm, E, ns, 1, lg, 0, 0, tw_inv, B, threadIdx.x, blockDim.x);
    for (u32 t = threadIdx.x; t < E; t += blockDim.x) { // n^-1 g^bitrev(i): any B <= 27 times a canonical entry -> < 2p
        R v;
#pragma unroll
        for (int l = 0; l < K; ++l) v.v[l] = sm[l * E + t];
        v = R::mul(v, R::load(post + (base + t) * K));
#pragma unroll
        for (int l = 0; l < K; ++l) sm[l * E + t] = v.v[l];
    }
    __syncthreads();
    B = 2;
    ntt_tile_stages<FrC, false>(sm, E, ns, 1, lg, 0, 0, tw_fwd, B, threadIdx.x, blockDim.x);
    for (u32 t = threadIdx.x; t < E; t += blockDim.x) {
        R v;
#pragma unroll
        for (int l = 0; l < K; ++l) v.v[l] = sm[l * E + t];
        v = R::template reduce<32>(v);
        u32 *p = data + (base + t) * K;
#pragma unroll
        for (int l = 0; l < K; ++l) p[l] = v.v[l];
    }
}
// high pass of the forward transform on a, b, c (DIT stages lo + 1 .. lg; a third of the workgroup each) + (a b - c) / Z + high pass of
// the inverse transform on the result (DIF stages lg .. lo + 1), one strided tile (column blockIdx.x) of 2^ns elements per vector;
// grid (n >> ns, 1, batch), 3 E / 2 threads
template <class FrC>
__global__ __launch_bounds__(1024) void ntt_fused_high_pw(u32 *__restrict__ a, u32 *__restrict__ b, u32 *__restrict__ c,
                                                          const u32 *__restrict__ tw_fwd, const u32 *__restrict__ tw_inv, unsigned lg,
                                                          unsigned ns, const u32 *__restrict__ zinv_rr) {
    MG_PRIO_WM();
    extern __shared__ __attribute__((aligned(16))) u32 sm[];
    typedef FpR<FrC> R;
    constexpr int K = R::K;
    const u32 E = 1u << ns, lo_bits = lg - ns, s0 = lo_bits + 1, lo_base = blockIdx.x;
    const u32 per = blockDim.x / 3, vec = threadIdx.x / per, tid = threadIdx.x - vec * per;
    u32 *__restrict__ data = (vec == 0 ? a : (vec == 1 ? b : c)) + ((size_t)blockIdx.z << lg) * K;
    u32 *__restrict__ tile = sm + (size_t)vec * K * E;
    for (u32 t = tid; t < E; t += per) {
        const u32 *p = data + ((size_t)lo_base + ((size_t)t << lo_bits)) * K;
#pragma unroll
        for (int l = 0; l < K; ++l) tile[l * E + t] = p[l];
    }
    __syncthreads();
    int B = 4;
    ntt_tile_stages<FrC, false>(tile, E, ns, s0, lg, lo_bits, lo_base, tw_fwd, B, tid, per);
    const R zinv = R::load(zinv_rr);
    u32 *ta = sm, *tb = sm + (size_t)K * E, *tc = sm + (size_t)2 * K * E;
    for (u32 t = threadIdx.x; t < E; t += blockDim.x) { // h = (a b - c) (g^D - 1)^-1 on the coset, < 2p
        R x, y, z;
#pragma unroll
        for (int l = 0; l < K; ++l) x.v[l] = ta[l * E + t], y.v[l] = tb[l * E + t], z.v[l] = tc[l * E + t];
        x = R::template reduce<32>(x), y = R::template reduce<32>(y), z = R::template reduce<32>(z);
        R ab = R::mul(x, y);
        ab = R::template sub<5>(ab, z);
        ab = R::mul(ab, zinv);
#pragma unroll
        for (int l = 0; l < K; ++l) ta[l * E + t] = ab.v[l];
    }
    __syncthreads();
    B = 2;
    ntt_tile_stages<FrC, true>(ta, E, ns, s0, lg, lo_bits, lo_base, tw_inv, B, threadIdx.x, blockDim.x);
    u32 *__restrict__ out = a + ((size_t)blockIdx.z << lg) * K;
    for (u32 t = threadIdx.x; t < E; t += blockDim.x) {
        R v;
#pragma unroll
        for (int l = 0; l < K; ++l) v.v[l] = ta[l * E + t];
        if (B > 4) v = R::template reduce<32>(v);
        u32 *p = out + ((size_t)lo_base + ((size_t)t << lo_bits)) * K;
#pragma unroll
        for (int l = 0; l < K; ++l) p[l] = v.v[l];
    }
}

// ---------------------------------------------------------------------------------------------------
// Round 4: the same pass with the butterflies of up to THREE consecutive stages done in REGISTERS. PMC of the kernel above
// (profiles/r04_pmc_ntt.txt, 2^20): 394 VALU instructions per butterfly of which 171 are multiply-adds, 40 LDS instructions per
// butterfly and bank conflicts in 54 % of the LDS-active cycles (stages whose partner distance is below 64 elements touch every
// other bank) -- the pass runs at 70-80 % of the issue bound of ITS OWN instruction stream, so the stream has to shrink. Here a
// thread owns the 2^LR (8) tile elements that differ in LR consecutive index bits, fetches them from LDS once, runs the LR stages
// that pair exactly those bits (12 butterflies, 7 distinct twiddles: index arithmetic and twiddle addressing once per thread and
// group, not once per butterfly and stage) and puts them back: ceil(ns / LR) LDS round trips and barriers per pass instead of
// ns. The tile is stored with one pad word per 32 (address t + t / 32): the strided accesses of the low groups -- 8 elements per
// lane, lane-to-lane stride 8 -- and the unit-stride ones of the high groups are both conflict-free.
// Same arguments, same bounds bookkeeping and same I/O options as ntt_pass_rr; tiles of at least 64 * 2^LR elements.
// one butterfly stage on the 2^LR elements a lane holds in registers: pairs the owned bit Q; 2^Q distinct twiddles (the owned
// bits below Q), each shared by the 2^(LR-1-Q) butterflies that differ in the owned bits above Q
// lazy (DIT only): the sums a + t and a + 3p - t are left with unnormalised limbs (< 2^31) -- legal wherever they only feed the
// next stage's product and its normalising additions: 9 x 2^31 x 2^29 + 9 x 2^58 < 2^64 -- which saves the two carry passes
// (54 of ~330 VALU instructions per butterfly); the subtraction then adds 3p (fpr_dev.h `subl`), one p more than the normalised form
// only the FIRST of two consecutive in-register DIT stages of a two-bit group: its lazy outputs (limbs < 2^29 + 2^30) meet one
// normalising stage before they are stored; groups of three would stack two lazy stages (limbs up to 2^29 + 2 x 2^30: too wide)
template <bool DIF> MG_DEV bool ntt_lazy_stage(int q, int lr, unsigned bit, unsigned first_bit, unsigned gs, unsigned s) {
#ifdef MG_NTT_NO_LAZY
    return false;
#else
    return !DIF && lr == 2 && q == 0 && gs == 2 && bit == first_bit && s > 1;
#endif
}
template <class FrC, bool DIF, int LR, int Q>
MG_DEV void ntt_reg_stage(FpR<FrC> (&v)[1 << LR], const u32 *__restrict__ tw, bool has_w, bool red, unsigned b0, u32 e_lo,
                          u32 lo_bits, u32 col, unsigned tw_shift, bool lazy = false, const u32 *__restrict__ stw = nullptr, u32 ntw = 0) {
    typedef FpR<FrC> R;
#pragma unroll
    for (int u = 0; u < (1 << Q); ++u) {
        R w;
        if (has_w && stw) { // the tile's twiddles staged in LDS (single-column tiles): entry 2^bit - 1 + jl, bit = b0 + Q
            const u32 *o = stw + ((1u << (b0 + (unsigned)Q)) - 1u + (((u32)u << b0) | e_lo));
#pragma unroll
            for (int l = 0; l < R::K; ++l) w.v[l] = o[l * ntw];
        } else if (has_w) {
            const u32 jl = ((u32)u << b0) | e_lo; // the stage bits below the paired one
            const size_t j = ((size_t)jl << lo_bits) | col;
            const uint4 *qp = reinterpret_cast<const uint4 *>(tw + (j << tw_shift) * 12); // 48 B records
            const uint4 q0 = qp[0], q1 = qp[1], q2 = qp[2];
            w.v[0] = q0.x, w.v[1] = q0.y, w.v[2] = q0.z, w.v[3] = q0.w, w.v[4] = q1.x, w.v[5] = q1.y, w.v[6] = q1.z, w.v[7] = q1.w,
            w.v[8] = q2.x;
        }
#pragma unroll
        for (int h = 0; h < (1 << (LR - 1 - Q)); ++h) {
            constexpr int dummy = 0;
            (void)dummy;
            const int i0 = (h << (Q + 1)) | u, i1 = i0 | (1 << Q);
            const R a = v[i0], b = v[i1];
            if (DIF) {
                R x = R::add(a, b);                   // < 2B p
                R y = R::template sub<9>(a, b);       // a + 9p - b (b < 8p) < (B + 9) p
                if (has_w) y = MG_NTT_MUL(y, w);      // < 2p
                else y = R::template reduce<17>(y);   // last DIF stage (w = 1)
                if (red) x = R::template reduce<16>(x);
                v[i0] = x, v[i1] = y;
            } else if (has_w) {
                const R t = MG_NTT_MUL(b, w); // < 2p
                if (lazy) {
#pragma unroll
                    for (int l = 0; l < R::K; ++l) v[i0].v[l] = a.v[l] + t.v[l];
                    v[i1] = R::template subl<3>(a, t);
                } else {
                    v[i0] = R::add(a, t);
                    v[i1] = R::template sub<2>(a, t);
                }
            } else { // first DIT stage (w = 1): b < 4p
                v[i0] = R::add(a, b);
                v[i1] = R::template sub<5>(a, b);
            }
        }
    }
}

#ifndef MG_NTT_REG_WAVES
#define MG_NTT_REG_WAVES 2
#endif
template <class FrC, bool DIF, int LR>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(MG_NTT_REG_WAVES, 8))) void ntt_pass_reg(u32 *__restrict__ d0, u32 *__restrict__ d1, u32 *__restrict__ d2,
                                                   const u32 *__restrict__ tw, unsigned lg, unsigned s0, unsigned ns,
                                                   unsigned cb, const u32 *__restrict__ post, NttIo io) {
    MG_PRIO_WM();
    extern __shared__ __attribute__((aligned(16))) u32 sm[];
    typedef FpR<FrC> R;
    constexpr int K = R::K, NE = 1 << LR;
    static_assert(K == 9 && R::LIM >= 64, "bound analysis of ntt_pass_rr");
    u32 *__restrict__ data = (blockIdx.y == 0 ? d0 : (blockIdx.y == 1 ? d1 : d2)) + ((size_t)blockIdx.z << lg) * K;
    const u32 E = 1u << ns, TOT = E << cb, CM = (1u << cb) - 1, TOTP = TOT + (TOT >> 5);
    const u32 lo_bits = s0 - 1;
    const u32 nlo = (1u << lo_bits) >> cb;
    const u32 lo_base = (blockIdx.x % nlo) << cb, hi_blk = blockIdx.x / nlo;
    const size_t base = ((size_t)hi_blk << (lo_bits + ns)) + lo_base;
    auto pad = [](u32 t) { return t + (t >> 5); };
    const bool twl = (io.packed & 4u) != 0; // single-column tiles: the tile's 2^ns - 1 twiddles in LDS behind it (see ntt_pass_rr)
    u32 *__restrict__ stw = sm + (size_t)K * TOTP;
    if (twl) {
        for (u32 x = threadIdx.x + 1; x < E; x += blockDim.x) {
            const unsigned tl = 32u - (unsigned)__clz(x);
            const unsigned s = s0 + tl - 1;
            if (s > 1) {
                const size_t j = ((size_t)(x - (1u << (tl - 1))) << lo_bits) | lo_base;
                const uint4 *q = reinterpret_cast<const uint4 *>(tw + (j << (lg - s)) * 12);
                const uint4 q0 = q[0], q1 = q[1], q2 = q[2];
                u32 *o = stw + (x - 1);
                o[0 * E] = q0.x, o[1 * E] = q0.y, o[2 * E] = q0.z, o[3 * E] = q0.w, o[4 * E] = q1.x, o[5 * E] = q1.y, o[6 * E] = q1.z,
                o[7 * E] = q1.w, o[8 * E] = q2.x;
            }
        }
    }
    for (u32 t = threadIdx.x; t < TOT; t += blockDim.x) {
        const size_t gi = base + ((size_t)(t >> cb) << lo_bits) + (t & CM);
        const u32 o = pad(t);
        R r;
        if (io.in_std) { // arkworks format at the bit-reversed index -> work form (< 2p)
            const u32 j = __brev((u32)gi) >> (32 - lg);
            r = R::from_std_shift(Fp<FrC>::load(io.in_std + (size_t)j * 8));
            if (io.pre_rr) r = R::mul(r, R::load(io.pre_rr + (size_t)j * K));
        } else if (io.packed & 1u) {
            r = R::load_packed(data + gi * 8);
        } else {
            r = R::load(data + gi * K);
        }
#pragma unroll
        for (int l = 0; l < K; ++l) sm[l * TOTP + o] = r.v[l];
    }
    __syncthreads();
    int B = 4; // every value in the tile is < B*p
    const u32 nthr = TOT >> LR;
    for (unsigned done = 0; done < ns;) {
        const unsigned gs = ns - done < (unsigned)LR ? ns - done : (unsigned)LR;
        // the LR stage bits this group's threads own: [b0, b0 + LR), the gs unprocessed ones among them are bits
        //   DIT (ascending):  done .. done + gs - 1          DIF (descending): ns - done - gs .. ns - done - 1
        const unsigned first_bit = DIF ? ns - done - gs : done;
        unsigned b0 = first_bit + gs >= (unsigned)LR ? (DIF ? first_bit : (first_bit + LR <= ns ? first_bit : ns - LR)) : 0;
        if (b0 + LR > ns) b0 = ns >= (unsigned)LR ? ns - LR : 0;
        const unsigned p0 = cb + b0;
        const int B_in = B;
        for (u32 vt = threadIdx.x; vt < nthr; vt += blockDim.x) { // (one trip when the workgroup has a thread per 2^LR elements)
            B = B_in;
            const u32 lo = vt & ((1u << p0) - 1), hi = vt >> p0;
            const u32 tb = (hi << (p0 + LR)) | lo;
            const u32 c = lo & CM, e_lo = lo >> cb; // column; the stage bits below b0
            R v[NE];
#pragma unroll
            for (int j = 0; j < NE; ++j) {
                const u32 o = pad(tb + ((u32)j << p0));
#pragma unroll
                for (int l = 0; l < K; ++l) v[j].v[l] = sm[l * TOTP + o];
            }
            // the stages of this group, one call per owned bit with the bit as a template constant (every register index static)
            auto run = [&](auto qc) {
                constexpr int Q = decltype(qc)::value;
                const unsigned bit = b0 + (unsigned)Q; // local stage bit; local stage number tl = bit + 1
                if (bit < first_bit || bit >= first_bit + gs) return; // (uniform) not one of this group's stages
                const unsigned s = s0 + bit;               // global stage
                // DIT: a stage whose outputs are consumed by another stage of this group (not stored) may leave them lazy
                const bool lazy = ntt_lazy_stage<DIF>(Q, LR, bit, first_bit, gs, s);
                ntt_reg_stage<FrC, DIF, LR, Q>(v, tw, s > 1, DIF && 2 * B > 8, b0, e_lo, lo_bits, lo_base + c, lg - s, lazy,
                                               twl ? stw : (const u32 *)nullptr, E);
                const bool red = DIF && 2 * B > 8;
                if (DIF) B = red ? 2 : 2 * B;
                else B = s > 1 ? B + (lazy ? 3 : 2) : B + 5;
            };
            if constexpr (DIF) {
                if constexpr (LR > 2) run(std::integral_constant<int, 2>());
                run(std::integral_constant<int, 1>());
                run(std::integral_constant<int, 0>());
            } else {
                run(std::integral_constant<int, 0>());
                run(std::integral_constant<int, 1>());
                if constexpr (LR > 2) run(std::integral_constant<int, 2>());
            }
#pragma unroll
            for (int j = 0; j < NE; ++j) {
                const u32 o = pad(tb + ((u32)j << p0));
#pragma unroll
                for (int l = 0; l < K; ++l) sm[l * TOTP + o] = v[j].v[l];
            }
        }
        { // the bound after this group's stages, the same for every thread (also for those without elements)
            B = B_in;
            for (unsigned k = 0; k < gs; ++k) {
                const unsigned bit = DIF ? first_bit + gs - 1 - k : first_bit + k;
                const bool has_w = s0 + bit > 1, red = DIF && 2 * B > 8;
                const bool lazy = ntt_lazy_stage<DIF>((int)(bit - b0), LR, bit, first_bit, gs, s0 + bit);
                if (DIF) B = red ? 2 : 2 * B;
                else B = has_w ? B + (lazy ? 3 : 2) : B + 5;
            }
        }
        __syncthreads();
        done += gs;
    }
    for (u32 t = threadIdx.x; t < TOT; t += blockDim.x) {
        const size_t gi = base + ((size_t)(t >> cb) << lo_bits) + (t & CM);
        const u32 o = pad(t);
        R v;
#pragma unroll
        for (int l = 0; l < K; ++l) v.v[l] = sm[l * TOTP + o];
        if (post) v = R::mul(v, R::load(post + gi * K)); // any B <= 27 times a canonical table entry: < 2p
        else if (DIF ? B > 4 : true) v = R::template reduce<32>(v);
        if (io.out_std) { // work form (< 4p) -> arkworks format, times the constant of the plain inverse transform first
            if (io.scale_rr) v = R::mul(v, R::load(io.scale_rr));
            v.to_std().store(io.out_std + gi * 8);
            continue;
        }
        if (io.packed & 2u) { // (v < 2p and normalised: a product or a reduce made it)
            v.store_packed(data + gi * 8);
            continue;
        }
        v.store(data + gi * K);
    }
}

// arkworks-format table (Montgomery, 8 words) -> canonical reduced-radix table with `stride` words per entry
template <class FrC>
__global__ __launch_bounds__(256) void std_to_rr_table_kernel(const u32 *__restrict__ in, u32 n, u32 *__restrict__ out, u32 stride) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    typedef FpR<FrC> R;
    const R r = R::from_std(Fp<FrC>::load(in + (size_t)i * 8)); // canonical (< p)
#pragma unroll
    for (int l = 0; l < R::K; ++l) out[(size_t)i * stride + l] = r.v[l];
    for (u32 l = R::K; l < stride; ++l) out[(size_t)i * stride + l] = 0;
}
// public-API entry: out_rr[i] = in_std[bitrev(i)] (x pre[bitrev(i)] for the forward coset transform), < 2p
template <class FrC>
__global__ __launch_bounds__(256) void ntt_load_rr_kernel(const u32 *__restrict__ in_std, unsigned lg, const u32 *__restrict__ pre_rr,
                                                          u32 *__restrict__ out_rr) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (1u << lg)) return;
    typedef FpR<FrC> R;
    const u32 j = lg ? (__brev(i) >> (32 - lg)) : 0;
    R r = R::from_std_shift(Fp<FrC>::load(in_std + (size_t)j * 8));
    if (pre_rr) r = R::mul(r, R::load(pre_rr + (size_t)j * R::K));
    r.store(out_rr + (size_t)i * R::K);
}
// reduced radix (< 4p) -> arkworks format, optionally times a constant first
template <class FrC>
__global__ __launch_bounds__(256) void rr_to_std_kernel(const u32 *__restrict__ in_rr, size_t n, const u32 *__restrict__ scale_rr,
                                                        u32 *__restrict__ out_std) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    typedef FpR<FrC> R;
    R r = R::load(in_rr + i * R::K);
    if (scale_rr) r = R::mul(r, R::load(scale_rr));
    r.to_std().store(out_std + i * 8);
}

// out[p] = in[bitrev(p)]
template <class FrC>
__global__ __launch_bounds__(256) void permute_bitrev_kernel(u32 *__restrict__ out, const u32 *__restrict__ in, unsigned lg) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (1u << lg)) return;
    const u32 j = lg ? (__brev(i) >> (32 - lg)) : 0;
    const uint4 *p = reinterpret_cast<const uint4 *>(in + (size_t)j * 8);
    uint4 *q = reinterpret_cast<uint4 *>(out + (size_t)i * 8);
    q[0] = p[0];
    q[1] = p[1];
}

// the same for three matrices at once (blockIdx.z): A z, B z, C z of the witness map. Outputs are written in the
// reduced-radix form of the NTT pipeline (9 words per element); matrix 0 (A) also fills the input-consistency rows
// a[m + j] = z_j, j < P (mpc.rs:299-312).
struct Csr3 {
    const u32 *row_ptr[3], *col[3], *val[3];
    u32 *out[3];
};
template <class FrC>
__global__ __launch_bounds__(256) void spmv3_kernel(Csr3 M, const u32 *__restrict__ z, u32 m, u32 P, size_t z_stride, size_t out_stride,
                                                    u32 rows_total) {
    MG_PRIO_WM();
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    const int t = blockIdx.z;
    z += (size_t)blockIdx.y * z_stride; // batch member
    u32 *__restrict__ out = M.out[t] + (size_t)blockIdx.y * out_stride;
    typedef Fp<FrC> F;
    typedef FpR<FrC> R;
    if (i >= m + (t == 0 ? P : 0)) {
        // rows past the constraints (and, in A, past the input-consistency rows) up to the domain size are zero -- written here, by the
        // kernel that owns the vector, not by a memset node in front of it (round 5: a hipMemsetAsync node of a LINEAR captured graph
        // replays with a wrong fill pattern once other work has gone through the runtime: profiles/r05_linear_graph_defect.txt)
        if (i < rows_total) {
#pragma unroll
            for (int k = 0; k < R::K; ++k) out[(size_t)i * R::K + k] = 0u;
        }
        return;
    }
    F acc = F::zero();
    if (i >= m) {
        acc = F::load(z + (size_t)(i - m) * 8);
    } else {
        const u32 *__restrict__ row_ptr = M.row_ptr[t], *__restrict__ col = M.col[t], *__restrict__ val = M.val[t];
        const F one = F::one();
        for (u32 k = row_ptr[i]; k < row_ptr[i + 1]; ++k) {
            F c = F::load(val + (size_t)k * 8);
            F x = F::load(z + (size_t)col[k] * 8);
            acc = F::add(acc, c == one ? x : F::mul(x, c));
        }
    }
    R::from_std_shift(acc).store(out + (size_t)i * R::K);
}

// a[i] = (a[i] b[i] - c[i]) * (g^D - 1)^-1, reduced-radix vectors (inputs < 4p, output < 2p)
template <class FrC>
__global__ __launch_bounds__(256) void qap_pointwise_kernel(u32 *__restrict__ a, const u32 *__restrict__ b,
                                                            const u32 *__restrict__ c, const u32 *__restrict__ zinv_rr,
                                                            u32 n) {
    MG_PRIO_WM();
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    typedef FpR<FrC> R;
    const size_t o = ((size_t)blockIdx.y * n + i) * R::K; // batch member
    R ab = R::mul(R::load(a + o), R::load(b + o));            // 4 * 4 <= LIM: < 2p
    ab = R::template sub<5>(ab, R::load(c + o));              // + 5p - c (c < 4p): < 7p
    R::mul(ab, R::load(zinv_rr)).store(a + o);
}

template <class FrC> class FrEngineT : public FrEngine {
  public:
    typedef host::HFp<FrC> HF;
    struct Domain {
        u32 *tw_fwd = nullptr, *tw_inv = nullptr; // omega^k, omega^-k, k < n/2
        u32 *coset_fwd = nullptr;                 // g^i
        u32 *coset_inv = nullptr;                 // n^-1 * g^-i
        u32 *consts = nullptr;                    // [0] n^-1, [1] (g^n - 1)^-1
        u32 *t1_br = nullptr;                     // n^-1 * g^bitrev(p)   (between ifft and coset fft)
        u32 *t2_br = nullptr;                     // n^-1 * g^-bitrev(p)  (after the final coset ifft)
        // the same tables in the NTT pipeline's reduced-radix form (canonical values): twiddles 12 words per entry
        // (three 16-byte loads), everything else 9
        u32 *tw_fwd_rr = nullptr, *tw_inv_rr = nullptr, *coset_fwd_rr = nullptr, *coset_inv_rr = nullptr, *consts_rr = nullptr,
            *t1_br_rr = nullptr, *t2_br_rr = nullptr;
        u32 *scratch_rr = nullptr; // one work vector for the public in-place transform (guarded by scratch_mu_)
    };
    static constexpr int RK = FpR<FrC>::K; // words per reduced-radix element
    int two_adicity() const override { return FrC::TWO_ADICITY; }

    static HF host_load(const u32 *w) {
        HF x;
        x.load_words(w);
        return x;
    }
    static HF hpow2k(HF x, int k) {
        for (int i = 0; i < k; ++i) x = HF::sqr(x);
        return x;
    }
    static HF from_u64(u64 v) {
        HF x = HF::zero();
        x.v[0] = v;
        return HF::to_mont(x);
    }

    int get_domain(unsigned log_n, Domain **out) {
        std::lock_guard<std::mutex> g(mu_);
        auto it = domains_.find(log_n);
        if (it != domains_.end()) {
            *out = &it->second;
            return MG_OK;
        }
        if ((int)log_n > FrC::TWO_ADICITY) return MG_ERR_DOMAIN;
        const size_t n = (size_t)1 << log_n;
        Domain d;
        HF root = hpow2k(host_load(FrC::ROOT), FrC::TWO_ADICITY - (int)log_n);
        HF root_inv = HF::inv(root);
        HF gen = host_load(FrC::GEN), gen_inv = HF::inv(gen);
        HF n_inv = HF::inv(from_u64((u64)n));
        HF zinv = HF::inv(HF::sub(hpow2k(gen, (int)log_n), HF::one()));
        const size_t half = n > 1 ? n / 2 : 1;
        MG_HIP(hipMalloc((void **)&d.tw_fwd, half * 32));
        MG_HIP(hipMalloc((void **)&d.tw_inv, half * 32));
        MG_HIP(hipMalloc((void **)&d.coset_fwd, n * 32));
        MG_HIP(hipMalloc((void **)&d.coset_inv, n * 32));
        MG_HIP(hipMalloc((void **)&d.consts, 64));
        // squarings of the four bases on the host, powers on the device
        u32 *d_sq = nullptr;
        MG_HIP(hipMalloc((void **)&d_sq, 4 * 33 * 32 + 32));
        std::vector<u32> hsq(4 * 33 * 8 + 8);
        HF bases[4] = {root, root_inv, gen, gen_inv};
        for (int b = 0; b < 4; ++b) {
            HF x = bases[b];
            for (int k = 0; k < 33; ++k) {
                x.store_words(&hsq[((size_t)b * 33 + k) * 8]);
                x = HF::sqr(x);
            }
        }
        n_inv.store_words(&hsq[4 * 33 * 8]);
        MG_HIP(memcpy_sync(d_sq, hsq.data(), hsq.size() * 4, hipMemcpyHostToDevice));
        const int bits = (int)log_n;
        const u32 gh = (u32)((half + 255) / 256), gn = (u32)((n + 255) / 256);
        hipLaunchKernelGGL((powers_kernel<FrC>), dim3(gh), dim3(256), 0, setup_stream(), d.tw_fwd, d_sq, (u32)half, bits,
                           (const u32 *)nullptr);
        hipLaunchKernelGGL((powers_kernel<FrC>), dim3(gh), dim3(256), 0, setup_stream(), d.tw_inv, d_sq + 33 * 8, (u32)half, bits,
                           (const u32 *)nullptr);
        hipLaunchKernelGGL((powers_kernel<FrC>), dim3(gn), dim3(256), 0, setup_stream(), d.coset_fwd, d_sq + 2 * 33 * 8, (u32)n,
                           bits + 1, (const u32 *)nullptr);
        hipLaunchKernelGGL((powers_kernel<FrC>), dim3(gn), dim3(256), 0, setup_stream(), d.coset_inv, d_sq + 3 * 33 * 8, (u32)n,
                           bits + 1, (const u32 *)(d_sq + 4 * 33 * 8));
        u32 hc[16];
        n_inv.store_words(hc);
        zinv.store_words(hc + 8);
        MG_HIP(memcpy_sync(d.consts, hc, 64, hipMemcpyHostToDevice));
        // bit-reversed scale tables of the permutation-free witness-map pipeline
        MG_HIP(hipMalloc((void **)&d.t1_br, n * 32));
        MG_HIP(hipMalloc((void **)&d.t2_br, n * 32));
        {
            u32 *tmp = nullptr;
            MG_HIP(hipMalloc((void **)&tmp, n * 32));
            hipLaunchKernelGGL((powers_kernel<FrC>), dim3(gn), dim3(256), 0, setup_stream(), tmp, d_sq + 2 * 33 * 8, (u32)n, bits + 1,
                               (const u32 *)(d_sq + 4 * 33 * 8)); // n^-1 * g^i
            hipLaunchKernelGGL((permute_bitrev_kernel<FrC>), dim3(gn), dim3(256), 0, setup_stream(), d.t1_br, tmp, log_n);
            hipLaunchKernelGGL((permute_bitrev_kernel<FrC>), dim3(gn), dim3(256), 0, setup_stream(), d.t2_br, d.coset_inv, log_n);
            MG_HIP(setup_sync());
            hipFree(tmp);
        }
        MG_HIP(setup_sync());
        hipFree(d_sq);
        { // reduced-radix copies
            struct T {
                u32 **dst;
                const u32 *src;
                size_t cnt;
                u32 stride;
            } tabs[] = {{&d.tw_fwd_rr, d.tw_fwd, half, 12},    {&d.tw_inv_rr, d.tw_inv, half, 12},
                        {&d.coset_fwd_rr, d.coset_fwd, n, RK}, {&d.coset_inv_rr, d.coset_inv, n, RK},
                        {&d.consts_rr, d.consts, 2, RK},       {&d.t1_br_rr, d.t1_br, n, RK},
                        {&d.t2_br_rr, d.t2_br, n, RK}};
            for (T &t : tabs) {
                MG_HIP(hipMalloc((void **)t.dst, t.cnt * t.stride * 4));
                hipLaunchKernelGGL((std_to_rr_table_kernel<FrC>), dim3((u32)((t.cnt + 255) / 256)), dim3(256), 0, setup_stream(), t.src, (u32)t.cnt,
                                   *t.dst, t.stride);
            }
            MG_HIP(setup_sync());
            // the saturated copies the pipeline no longer reads are released (the public NTT keeps none of them either)
            // (tw_fwd / tw_inv stay: the group-domain NTT reads them)
            hipFree(d.t1_br), hipFree(d.t2_br), hipFree(d.coset_fwd), hipFree(d.coset_inv);
            d.t1_br = d.t2_br = d.coset_fwd = d.coset_inv = nullptr;
        }
        auto ins = domains_.emplace(log_n, d);
        *out = &ins.first->second;
        return MG_OK;
    }

    // stages per LDS round trip of the NTT passes. Measured (profiles/r04_ntt_register_stages.txt): the DIT passes gain from two
    // stages per round trip (2^20 public transform 0.191 -> 0.17 ms; the 3-vector passes of a 32-proof witness map 268 + 428 ->
    // 248 + 313 us), the DIF passes do not (318 + 415 -> 340 + 433 us: their product comes AFTER the subtraction, the extra live
    // registers cost occupancy) and three stages per round trip lose everywhere (215-256 VGPRs). MANTA_NTT_R = 0 / 2 / 3 forces one.
    static int ntt_reg_bits(bool dif) {
        static const int v = [] {
            const int x = ab_knob("MANTA_NTT_R", -1);
            return x == 0 || x == 2 || x == 3 ? x : -1;
        }();
        return v >= 0 ? v : (dif ? 0 : 2);
    }
    static u32 ntt_threads() {
        static const u32 v = [] {
            return (u32)ab_knob("MANTA_NTT_THREADS", 0);
        }();
        return v;
    }
    // all stages of one transform over up to 3 reduced-radix vectors as LDS-fused passes of <= 10 stages
    template <bool DIF>
    static void run_passes(u32 *d0, u32 *d1, u32 *d2, int nvec, const u32 *tw_rr, unsigned lg, const u32 *post_rr, hipStream_t s,
                           u32 batch = 1, NttIo io = NttIo{nullptr, nullptr, nullptr, nullptr, 0u}) {
        if (lg == 0) return;
        static const bool attr_set = [] { // tiles of 2048 elements x 36 B = 72 KB: above the 64 KB default of dynamic LDS
            hipFuncSetAttribute(reinterpret_cast<const void *>(&ntt_pass_rr<FrC, DIF>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                36 * 2048);
            hipFuncSetAttribute(reinterpret_cast<const void *>(&ntt_pass_reg<FrC, DIF, 3>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                36 * (2048 + 64));
            hipFuncSetAttribute(reinterpret_cast<const void *>(&ntt_pass_reg<FrC, DIF, 2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                36 * (2048 + 64));
            return true;
        }();
        (void)attr_set;
        const unsigned npass = (lg + 9) / 10;
        unsigned done = 0;
        for (unsigned p = 0; p < npass; ++p) {
            const unsigned ns = (lg - done + (npass - p) - 1) / (npass - p); // balanced split
            // DIT walks the stages upwards, DIF downwards
            const unsigned s0 = DIF ? (lg - done - ns + 1) : (done + 1);
            const unsigned lo_bits = s0 - 1;
            unsigned cb = ns >= 11 ? 0 : 11 - ns; // tile <= 2048 elements = 72 KB of LDS (two workgroups per CU)
            if (cb > lo_bits) cb = lo_bits;
            if (cb > 3) cb = 3;
            // a SINGLE proof's witness map is a latency chain of six passes over three 2^16-element vectors: with 2048-element
            // tiles a pass is 32-96 workgroups on 256 CUs. Narrower column groups = more, smaller workgroups (the vectors live in
            // L2: the coalescing the wide groups buy is worth nothing here). MANTA_NTT_MIN_WGS = workgroups a pass should have.
            static const u32 min_wgs = [] {
                const int v = ab_knob("MANTA_NTT_MIN_WGS", 512);
                return (u32)(v >= 1 ? v : 1);
            }();
            while (cb > 0 && ((((size_t)1 << lg) >> (ns + cb)) * (size_t)nvec * batch) < min_wgs && (1u << (ns + cb - 1)) >= 128u) --cb;
            const u32 blocks = (u32)(((size_t)1 << lg) >> (ns + cb));
            const size_t lds = ((size_t)(4 * RK) << (ns + cb));
            const bool last = p + 1 == npass;
            // one butterfly per thread and stage when the tile is full (2048 elements -> 1024 threads = 4 wavefronts
            // per SIMD): a pass is a chain of dependent multiplications, more resident waves hide its latency
            const u32 tot = 1u << (ns + cb);
            const u32 threads = ntt_threads() ? ntt_threads() : (tot >= 2048 ? 1024u : tot >= 128 ? tot / 2 : 64u);
            // public transform (arkworks format in and out, d0 = a scratch vector): the passes hand the vector on in the packed form
            const bool pack = io.in_std && io.out_std && npass > 1 && !DIF;
            static const bool twl_on = [] {
                return ab_knob("MANTA_NTT_TWL", 1) != 0;
            }();
            // single-column tiles of a LATENCY-bound launch (one proof's witness map: <= 2 workgroups per CU): twiddles staged in LDS
            // (E entries of 36 B more). A 2^20 transform is 1 024 workgroups per pass and throughput-bound: there the larger LDS
            // footprint costs 8 % (0.154 -> 0.166 ms, tools/ntt_ab_r5.sh), so it keeps the per-stage gathers.
            const bool twl = twl_on && cb == 0 && ns >= 2 && (size_t)blocks * nvec * batch <= 512;
            const size_t twl_bytes = twl ? ((size_t)(4 * RK) << ns) : 0;
            NttIo pio{p == 0 ? io.in_std : nullptr, p == 0 ? io.pre_rr : nullptr, last ? io.out_std : nullptr, last ? io.scale_rr : nullptr,
                      (pack ? (p > 0 ? 1u : 0u) | (last ? 0u : 2u) : 0u) | (twl ? 4u : 0u)};
            // round 4: three stages per LDS round trip in registers (ntt_pass_reg) for every tile with a full wavefront of
            // 8-element lanes; MANTA_NTT_R = 0 restores the stage-per-round-trip kernel, 2 = four elements per lane
            const int lr = ntt_reg_bits(DIF);
            if (lr && tot >= (64u << lr) && ns >= (unsigned)lr) {
                const size_t lds_p = (size_t)(4 * RK) * (tot + (tot >> 5)) + twl_bytes;
                const u32 th = std::min<u32>(512u, tot >> lr);
                if (lr == 3)
                    hipLaunchKernelGGL((ntt_pass_reg<FrC, DIF, 3>), dim3(blocks, nvec, batch), dim3(th), lds_p, s, d0, d1, d2, tw_rr, lg, s0,
                                       ns, cb, last ? post_rr : (const u32 *)nullptr, pio);
                else
                    hipLaunchKernelGGL((ntt_pass_reg<FrC, DIF, 2>), dim3(blocks, nvec, batch), dim3(th), lds_p, s, d0, d1, d2, tw_rr, lg, s0,
                                       ns, cb, last ? post_rr : (const u32 *)nullptr, pio);
            } else
            hipLaunchKernelGGL((ntt_pass_rr<FrC, DIF>), dim3(blocks, nvec, batch), dim3(threads), lds + twl_bytes, s, d0, d1, d2, tw_rr, lg, s0,
                               ns, cb, last ? post_rr : (const u32 *)nullptr, pio);
            done += ns;
        }
    }

    // Radix2EvaluationDomain::{fft, ifft, coset_fft, coset_ifft}_in_place on 2^log_n arkworks-format elements in HBM:
    // converted to the reduced-radix form on the way in (bit-reversal and the coset pre-scaling folded into that
    // kernel), DIT passes, scaled and converted back on the way out.
    int transform(u32 *d_data, unsigned log_n, bool inverse, bool coset, hipStream_t s) override {
        Domain *d;
        int rc = get_domain(log_n, &d);
        if (rc) return rc;
        const u32 n = 1u << log_n;
        const u32 gn = (n + 255) / 256;
        // the transform runs on a work vector owned by the domain: calls on one domain size are serialised (they are
        // synchronous at the ABI anyway), none pays an allocation
        std::lock_guard<std::mutex> g(scratch_mu_);
        if (!d->scratch_rr) MG_HIP(hipMalloc((void **)&d->scratch_rr, (size_t)n * RK * 4));
        u32 *tmp = d->scratch_rr;
        const bool timed = kernel_timing(); // bench.py's NTT leg: HIP events between the kernels of this call
        hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
        if (timed)
            for (auto &e : ev) MG_HIP(hipEventCreate(&e));
        if (timed) MG_HIP(hipEventRecord(ev[0], s));
        const u32 *post = inverse && coset && log_n > 0 ? d->coset_inv_rr : nullptr; // n^-1 g^-i, natural order
        if (log_n > 0) {
            // conversions folded into the first / last pass: arkworks format (bit-reversed gather, coset pre-scaling) -> passes
            // on the work vector -> n^-1 of the plain inverse transform and arkworks format out
            if (timed) MG_HIP(hipEventRecord(ev[1], s));
            NttIo io{d_data, (!inverse && coset) ? d->coset_fwd_rr : (const u32 *)nullptr, d_data,
                     (inverse && !coset) ? d->consts_rr : (const u32 *)nullptr, 0u};
            // (no hazard on d_data: with one pass the single workgroup has the whole vector in LDS before it writes; with several,
            // the first pass only reads it -- into the work vector -- and only the last one writes it)
            run_passes<false>(tmp, tmp, tmp, 1, inverse ? d->tw_inv_rr : d->tw_fwd_rr, log_n, post, s, 1, io);
            if (timed) MG_HIP(hipEventRecord(ev[2], s));
            if (timed) MG_HIP(hipEventRecord(ev[3], s));
        } else {
            hipLaunchKernelGGL((ntt_load_rr_kernel<FrC>), dim3(gn), dim3(256), 0, s, d_data, log_n, (const u32 *)nullptr, tmp);
            if (timed) MG_HIP(hipEventRecord(ev[1], s));
            if (timed) MG_HIP(hipEventRecord(ev[2], s));
            hipLaunchKernelGGL((rr_to_std_kernel<FrC>), dim3(gn), dim3(256), 0, s, tmp, (size_t)n, (const u32 *)nullptr, d_data);
            if (timed) MG_HIP(hipEventRecord(ev[3], s));
        }
        hipError_t e = hipStreamSynchronize(s); // the scratch vector is handed to the next caller after this
        if (timed) {
            float v[4] = {0, 0, 0, 0};
            if (e == hipSuccess) {
                hipEventElapsedTime(&v[0], ev[0], ev[3]);
                hipEventElapsedTime(&v[1], ev[0], ev[1]);
                hipEventElapsedTime(&v[2], ev[1], ev[2]);
                hipEventElapsedTime(&v[3], ev[2], ev[3]);
            }
            set_last_ntt_ms(v);
            for (auto &x : ev) hipEventDestroy(x);
        }
        if (e != hipSuccess) {
            set_last_hip_error(e, "ntt transform", __FILE__, __LINE__);
            return MG_ERR_HIP;
        }
        MG_HIP(hipGetLastError());
        return MG_OK;
    }
    // QAP quotient in place on reduced-radix vectors: a, b, c hold the constraint evaluations; on return a holds the
    // coefficients of h = (A B - C)/Z in BIT-REVERSED order (the h-query bases are stored in the same order), < 2p.
    // ifft = DIF (natural -> bit-reversed) with n^-1 g^i folded into its last pass, coset fft = DIT
    // (bit-reversed -> natural): no permutation pass, 2 x ceil(lg/10) launches per transform, a/b/c batched.
    int qap_quotient(u32 *a, u32 *b, u32 *c, unsigned lg, hipStream_t s, u32 batch = 1) override {
        Domain *d;
        int rc = get_domain(lg, &d);
        if (rc) return rc;
        const u32 n = 1u << lg;
        if (lg == 0) { // degenerate domain: h = (a b - c) / (g - 1) scaled as the general path would
            hipLaunchKernelGGL((qap_pointwise_kernel<FrC>), dim3(1, batch), dim3(256), 0, s, a, b, c, d->consts_rr + RK, n);
            MG_HIP(hipGetLastError());
            return MG_OK;
        }
        // single proofs (single-column tiles, two passes per transform): four fused launches instead of seven (MANTA_NTT_FUSE=0: A/B)
        static const int fuse_on = [] { // bit 0: the low passes (ntt_fused_low), bit 1: high pass + pointwise + high pass
            return ab_knob("MANTA_NTT_FUSE", 3) & 3;
        }();
        const unsigned L = lg / 2, H = lg - L; // low / high stages of every transform
        if (fuse_on && lg >= 12 && lg <= 18 && (size_t)batch * 3 * ((size_t)n >> (H + 1)) < 512) {
            static const bool attr_set = [] {
                hipFuncSetAttribute(reinterpret_cast<const void *>(&ntt_fused_high_pw<FrC>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    3 * 36 * 512);
                return true;
            }();
            (void)attr_set;
            const NttIo none{nullptr, nullptr, nullptr, nullptr, 0u};
            // inverse transform, high stages lg .. L + 1 (strided tiles of 2^H), a | b | c
            hipLaunchKernelGGL((ntt_pass_rr<FrC, true>), dim3(n >> H, 3, batch), dim3(std::max(64u, (1u << H) / 2)), (size_t)(4 * RK) << H, s, a, b, c,
                               d->tw_inv_rr, lg, L + 1, H, 0u, (const u32 *)nullptr, none);
            // its low stages + n^-1 g^i + the forward coset transform's low stages (contiguous tiles of 2^L)
            if (fuse_on & 1) {
                hipLaunchKernelGGL((ntt_fused_low<FrC>), dim3(n >> L, 3, batch), dim3(std::max(64u, (1u << L) / 2)), (size_t)(4 * RK) << L, s, a, b, c,
                                   d->tw_inv_rr, d->tw_fwd_rr, lg, L, d->t1_br_rr);
            } else {
                hipLaunchKernelGGL((ntt_pass_rr<FrC, true>), dim3(n >> L, 3, batch), dim3(std::max(64u, (1u << L) / 2)), (size_t)(4 * RK) << L, s, a, b, c,
                                   d->tw_inv_rr, lg, 1u, L, 0u, d->t1_br_rr, none);
                hipLaunchKernelGGL((ntt_pass_rr<FrC, false>), dim3(n >> L, 3, batch), dim3(std::max(64u, (1u << L) / 2)), (size_t)(4 * RK) << L, s, a, b, c,
                                   d->tw_fwd_rr, lg, 1u, L, 0u, (const u32 *)nullptr, none);
            }
            // forward high stages on a, b, c + (a b - c) / Z + the last inverse transform's high stages
            if (fuse_on & 2) {
                hipLaunchKernelGGL((ntt_fused_high_pw<FrC>), dim3(n >> H, 1, batch), dim3(3 * std::max(64u, (1u << H) / 2)), (size_t)(3 * 4 * RK) << H, s,
                                   a, b, c, d->tw_fwd_rr, d->tw_inv_rr, lg, H, d->consts_rr + RK);
            } else {
                hipLaunchKernelGGL((ntt_pass_rr<FrC, false>), dim3(n >> H, 3, batch), dim3(std::max(64u, (1u << H) / 2)), (size_t)(4 * RK) << H, s, a, b, c,
                                   d->tw_fwd_rr, lg, L + 1, H, 0u, (const u32 *)nullptr, none);
                hipLaunchKernelGGL((qap_pointwise_kernel<FrC>), dim3((n + 255) / 256, batch), dim3(256), 0, s, a, b, c, d->consts_rr + RK, n);
                hipLaunchKernelGGL((ntt_pass_rr<FrC, true>), dim3(n >> H, 1, batch), dim3(std::max(64u, (1u << H) / 2)), (size_t)(4 * RK) << H, s, a, a, a,
                                   d->tw_inv_rr, lg, L + 1, H, 0u, (const u32 *)nullptr, none);
            }
            // its low stages + n^-1 g^-i
            hipLaunchKernelGGL((ntt_pass_rr<FrC, true>), dim3(n >> L, 1, batch), dim3(std::max(64u, (1u << L) / 2)), (size_t)(4 * RK) << L, s, a, a, a,
                               d->tw_inv_rr, lg, 1u, L, 0u, d->t2_br_rr, none);
            MG_HIP(hipGetLastError());
            return MG_OK;
        }
        run_passes<true>(a, b, c, 3, d->tw_inv_rr, lg, d->t1_br_rr, s, batch);
        run_passes<false>(a, b, c, 3, d->tw_fwd_rr, lg, nullptr, s, batch);
        hipLaunchKernelGGL((qap_pointwise_kernel<FrC>), dim3((n + 255) / 256, batch), dim3(256), 0, s, a, b, c, d->consts_rr + RK, n);
        run_passes<true>(a, a, a, 1, d->tw_inv_rr, lg, d->t2_br_rr, s, batch);
        MG_HIP(hipGetLastError());
        return MG_OK;
    }
    int work_words() const override { return RK; }
    int spmv3(const DevCsr &A, const DevCsr &B, const DevCsr &C, const u32 *d_z, u32 *d_a, u32 *d_b, u32 *d_c, u64 m, u64 P,
              hipStream_t s, u32 batch = 1, size_t z_stride = 0, size_t out_stride = 0, u64 rows_total = 0) override {
        if (rows_total < m + P) rows_total = m + P; // (0: the caller has zeroed the vectors itself)
        if (rows_total == 0) return MG_OK; // (m == 0 with rows to write still launches: the kernel owns the zero rows up to rows_total)
        Csr3 M{{A.row_ptr, B.row_ptr, C.row_ptr}, {A.col, B.col, C.col}, {A.val, B.val, C.val}, {d_a, d_b, d_c}};
        hipLaunchKernelGGL((spmv3_kernel<FrC>), dim3((u32)((rows_total + 255) / 256), batch, 3), dim3(256), 0, s, M, d_z, (u32)m, (u32)P,
                           z_stride, out_stride, (u32)rows_total);
        MG_HIP(hipGetLastError());
        return MG_OK;
    }
    // reduced-radix vector -> arkworks format (tests / mg_witness_map)
    int work_to_std(const u32 *d_rr, size_t n, u32 *d_std, hipStream_t s) override {
        hipLaunchKernelGGL((rr_to_std_kernel<FrC>), dim3((u32)((n + 255) / 256)), dim3(256), 0, s, d_rr, n, (const u32 *)nullptr, d_std);
        MG_HIP(hipGetLastError());
        return MG_OK;
    }
    int domain_twiddles(unsigned log_n, bool inverse, const u32 **d_tw, u64 n_inv_canonical[4]) override {
        Domain *d;
        int rc = get_domain(log_n, &d);
        if (rc) return rc;
        *d_tw = inverse ? d->tw_inv : d->tw_fwd;
        const HF ninv = HF::from_mont(HF::inv(from_u64((u64)1 << log_n)));
        std::memcpy(n_inv_canonical, ninv.v, 32);
        return MG_OK;
    }
    void fr_mul(const u64 a[4], const u64 b[4], u64 out[4]) const override {
        HF x, y;
        std::memcpy(x.v, a, 32);
        std::memcpy(y.v, b, 32);
        HF r = HF::mul(x, y);
        std::memcpy(out, r.v, 32);
    }
    int setup_scalars(const mg_csr *A, const mg_csr *B, const mg_csr *Cm, u64 m, u64 V, u64 P, unsigned log_d,
                      const u64 *toxic5, std::vector<u64> &s1, std::vector<u64> &s2) const override {
        if ((int)log_d > FrC::TWO_ADICITY || V < 2 || P < 1 || P >= V) return MG_ERR_ARG;
        const size_t D = (size_t)1 << log_d;
        if (m + P > D) return MG_ERR_ARG;
        auto ld = [](const u64 *p) {
            HF x;
            std::memcpy(x.v, p, 32);
            return x;
        };
        const HF alpha = ld(toxic5), beta = ld(toxic5 + 4), gamma = ld(toxic5 + 8), delta = ld(toxic5 + 12),
                 tau = ld(toxic5 + 16);
        if (gamma.is_zero() || delta.is_zero()) return MG_ERR_ARG;
        HF w; // primitive D-th root of unity
        w.load_words(FrC::ROOT);
        for (unsigned i = log_d; i < (unsigned)FrC::TWO_ADICITY; ++i) w = HF::sqr(w);
        HF tD = tau;
        for (unsigned i = 0; i < log_d; ++i) tD = HF::sqr(tD);
        const HF Zt = HF::sub(tD, HF::one()); // Z(tau) = tau^D - 1
        // L_i(tau) = Z(tau)/D * w^i / (tau - w^i); the D denominators are inverted together
        std::vector<HF> pw(D), den(D), pref(D + 1);
        pw[0] = HF::one();
        for (size_t i = 1; i < D; ++i) pw[i] = HF::mul(pw[i - 1], w);
        pref[0] = HF::one();
        for (size_t i = 0; i < D; ++i) {
            den[i] = HF::sub(tau, pw[i]);
            if (den[i].is_zero()) return MG_ERR_ARG; // tau inside the domain
            pref[i + 1] = HF::mul(pref[i], den[i]);
        }
        HF inv = HF::inv(pref[D]);
        HF dD = HF::one();
        for (unsigned i = 0; i < log_d; ++i) dD = HF::dbl(dD);
        const HF zd = HF::mul(Zt, HF::inv(dD));
        std::vector<HF> L(D);
        for (size_t i = D; i-- > 0;) {
            const HF di = HF::mul(inv, pref[i]);
            inv = HF::mul(inv, den[i]);
            L[i] = HF::mul(HF::mul(zd, pw[i]), di);
        }
        std::vector<HF> av(V, HF::zero()), bv(V, HF::zero()), cv(V, HF::zero());
        const mg_csr *Ms[3] = {A, B, Cm};
        std::vector<HF> *acc[3] = {&av, &bv, &cv};
        for (int t = 0; t < 3; ++t) {
            const mg_csr *M = Ms[t];
            if (!M || !M->row_ptr || (M->nnz && (!M->col || !M->val)) || M->row_ptr[0] != 0 || M->row_ptr[m] != M->nnz)
                return MG_ERR_ARG;
            for (u64 i = 0; i < m; ++i) // the loop below walks row_ptr[i] .. row_ptr[i+1] on the host: monotone or rejected
                if (M->row_ptr[i] > M->row_ptr[i + 1]) return MG_ERR_ARG;
            for (u64 i = 0; i < m; ++i)
                for (u64 k = M->row_ptr[i]; k < M->row_ptr[i + 1]; ++k) {
                    if (M->col[k] >= V) return MG_ERR_ARG;
                    HF &x = (*acc[t])[M->col[k]];
                    x = HF::add(x, HF::mul(ld(M->val + 4 * k), L[i]));
                }
        }
        for (u64 j = 0; j < P; ++j) av[j] = HF::add(av[j], L[m + j]); // input-consistency rows (mpc.rs:299-312)
        const HF ginv = HF::inv(gamma), dinv = HF::inv(delta);
        s1.assign((3 + P + 2 * V + (D - 1) + (V - P)) * 4, 0);
        s2.assign((3 + V) * 4, 0);
        auto put = [](std::vector<u64> &dst, size_t idx, const HF &x) {
            const HF c = HF::from_mont(x);
            std::memcpy(&dst[idx * 4], c.v, 32);
        };
        put(s1, 0, alpha), put(s1, 1, beta), put(s1, 2, delta);
        put(s2, 0, beta), put(s2, 1, gamma), put(s2, 2, delta);
        size_t o_abc = 3, o_a = o_abc + P, o_b = o_a + V, o_h = o_b + V, o_l = o_h + (D - 1);
        for (u64 j = 0; j < V; ++j) {
            const HF ext = HF::add(HF::add(HF::mul(beta, av[j]), HF::mul(alpha, bv[j])), cv[j]);
            put(s1, o_a + j, av[j]);
            put(s1, o_b + j, bv[j]);
            put(s2, 3 + j, bv[j]);
            if (j < P) put(s1, o_abc + j, HF::mul(ext, ginv));
            else put(s1, o_l + (j - P), HF::mul(ext, dinv));
        }
        HF cur = HF::mul(Zt, dinv);
        for (size_t i = 0; i + 1 < D; ++i) {
            put(s1, o_h + i, cur);
            cur = HF::mul(cur, tau);
        }
        return MG_OK;
    }
    void fr_to_canonical(const u64 a[4], u64 out[4]) const override {
        HF x;
        std::memcpy(x.v, a, 32);
        HF r = HF::from_mont(x);
        std::memcpy(out, r.v, 32);
    }

  private:
    std::mutex mu_, scratch_mu_;
    std::map<unsigned, Domain> domains_;
};

} // namespace mg
