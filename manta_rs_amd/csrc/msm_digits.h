// Pippenger MSM, stage K5: scalars -> signed window digits -> (bucket key, base index | sign) pairs. Part of msm_impl.h.
#pragma once
#include "msm_common.h"

namespace mg {

// --------------------------------------------------------------------------------------------
// K5: digits
// --------------------------------------------------------------------------------------------
// One lane per (stored base, scalar vector of the batch): W signed c-bit digits -> (bucket key, base index | sign)
// pairs. Zero digits produce NO pair: real witnesses are 40 % zeros and 25 % ones, so two thirds of all digits
// vanish here instead of being carried through the sort. The surviving pairs are appended to the arrays in
// wave-sized, window-major groups (one atomicAdd on `count` per wavefront, positions by ballot/popcount: the
// order is irrelevant, the sort follows); every later stage reads the pair count from the device.
// (A count -> scan -> write version without the atomic was measured too: the kernel is bound by the scalar loads and the
// Montgomery conversion, not by the append, so running it twice costs more than the atomics do -- 2 x 105 + 46 us against
// 127 us for 32 x 2^15 scalars.)
// count == nullptr selects the fixed layout o = w*n + i with an `invalid` key for zero digits (library-sort path).
template <class FrC>
__global__ __launch_bounds__(1024) void digits_kernel(const u32 *__restrict__ scalars, u32 n, int c, int W, u32 B,
                                                     int precomp, u32 tstride, int mont, u32 invalid,
                                                     u32 *__restrict__ keys, u32 *__restrict__ vals,
                                                     const u32 *__restrict__ map, u32 n_scalars,
                                                     size_t scalar_stride, u32 seg_keys, u32 *__restrict__ count,
                                                     u32 n_sets = 1, u32 set_len = 0, u32 i_first = 0) {
    MG_PRIO_HIGH();
    const u32 i = i_first + blockIdx.x * blockDim.x + threadIdx.x; // (i_first: one launch per query, lanes [i_first, n))
    // blockIdx.y = scalar vector of a batch: its own scalars, its own range of bucket keys; the bases (and so
    // the values) are shared
    scalars += (size_t)blockIdx.y * scalar_stride;
    u32 src = (i < n) ? (map ? map[i] : i) : 0xffffffffu; // which scalar belongs to stored base i
    // concatenated queries (BaseSet::n_sets): original entry j = query j / set_len, scalar j % set_len; every (vector, query)
    // pair has its own range of bucket keys
    u32 set = 0;
    if (n_sets > 1 && i < n) {
        set = src / set_len;
        src -= set * set_len;
    }
    const u32 key0 = (blockIdx.y * n_sets + set) * seg_keys;
    const bool have = i < n && src < n_scalars; // the scalar vector may be shorter than the base set: zip
    u32 s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (have && mont == 2) { // the witness map's reduced-radix work form (9 words): one product with the integer 1
        typedef FpR<FrC> R;
        const Fp<FrC> f = R::load(scalars + (size_t)src * R::K).to_canonical();
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] = f.v[j];
    } else if (have) {
        const uint4 *p = reinterpret_cast<const uint4 *>(scalars + (size_t)src * 8);
        uint4 a = p[0], b = p[1];
        s[0] = a.x, s[1] = a.y, s[2] = a.z, s[3] = a.w, s[4] = b.x, s[5] = b.y, s[6] = b.z, s[7] = b.w;
        if (mont) { // ark-ff into_repr on the device
            Fp<FrC> f;
#pragma unroll
            for (int j = 0; j < 8; ++j) f.v[j] = s[j];
            f = Fp<FrC>::from_mont(f);
#pragma unroll
            for (int j = 0; j < 8; ++j) s[j] = f.v[j];
        }
    }
    // k P = (r - k)(-P): the smaller of k and r - k is below 2^(BITS - 1), so ceil(BITS / c) signed windows hold it -- one fewer
    // than the ceil((BITS + 1) / c) a scalar up to r - 1 needs whenever c divides BITS (BLS12-381, 255 bits: 15 windows of 17
    // bits instead of 16). The sign of every digit flips with the scalar.
    // A scalar that is NOT below r (the ABI says canonical, arkworks' multi_scalar_mul takes any BigInteger256 and treats it as
    // the integer it is) is first reduced: k P = (k mod r) P, and 2^256 < 6 r on both curves. Without this its top window could
    // exceed B and drop a carry. Wave-uniform early exit: canonical input pays one borrow chain.
    for (int it = 0; it < 6; ++it) {
        u32 t[8], bw = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const u64 d = (u64)s[j] - FrC::P[j] - bw;
            t[j] = (u32)d;
            bw = (u32)(d >> 63);
        }
        if (!__any(!bw)) break; // every lane's scalar is below r
        if (!bw) {
#pragma unroll
            for (int j = 0; j < 8; ++j) s[j] = t[j];
        }
    }
    u32 flip = 0;
    {
        u32 t[8], bw = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const u64 d = (u64)FrC::P[j] - s[j] - bw;
            t[j] = (u32)d;
            bw = (u32)(d >> 63);
        }
        bool lt = false; // r - k < k
#pragma unroll
        for (int j = 0; j < 8; ++j) lt = t[j] != s[j] ? t[j] < s[j] : lt;
        if (!bw && lt) {
            flip = 1;
#pragma unroll
            for (int j = 0; j < 8; ++j) s[j] = t[j];
        }
    }
    const u32 mask = (1u << c) - 1;
    // signed digit of the NEXT window (0 = nothing to add): the low c bits, then the scalar moves right by c -- eight
    // funnel shifts instead of a dynamically indexed limb pair (~90 instructions per window in selects, which made this
    // kernel issue-bound at 5k instructions per scalar: 172 -> 127 us for 32 x 2^15 scalars, 40 -> 23 us for one 2^15)
    auto next_digit = [&](u32 (&t)[8], u32 &carry, u32 &neg) -> u32 {
        u32 d = (t[0] & mask) + carry;
#pragma unroll
        for (int j = 0; j < 7; ++j) t[j] = __funnelshift_r(t[j], t[j + 1], c);
        t[7] >>= c;
        neg = d > B;
        carry = neg;
        return neg ? (1u << c) - d : d;
    };
    if (!count) { // fixed layout
        if (i >= n) return;
        keys += (size_t)blockIdx.y * W * n;
        vals += (size_t)blockIdx.y * W * n;
        u32 carry = 0, neg;
        for (int w = 0; w < W; ++w) {
            const u32 d = next_digit(s, carry, neg); // s = 0 without a scalar
            const size_t o = (size_t)w * n + i;
            keys[o] = d ? key0 + (precomp == 2 ? 0u : precomp ? (d - 1) : ((u32)w * B + d - 1)) : invalid;
            vals[o] = d ? ((precomp == 2 ? ((u32)w * tstride + i) * B + (d - 1) : precomp ? ((u32)w * tstride + i) : i) | ((neg ^ flip) << 31)) : 0;
        }
        return;
    }
    // pass 1: how many pairs does this wavefront produce
    const int lane = threadIdx.x & 63;
    u32 total = 0;
    {
        u32 t[8], carry = 0, neg;
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = s[j];
        for (int w = 0; w < W; ++w) total += (u32)__popcll(__ballot(next_digit(t, carry, neg) != 0));
    }
    // one atomic per WORKGROUP (up to sixteen wavefronts add up through LDS): the counter is a single address shared
    // by the whole grid, and atomics on it serialise at ~50 ns each -- one per wavefront (16 384 at 2^20 scalars) made the
    // kernel 0.41 ms, one per 256 threads 0.28 ms
    __shared__ u32 wave_tot[16], block_base;
    const int wv = threadIdx.x >> 6, nwv = blockDim.x >> 6;
    if (lane == 0) wave_tot[wv] = total;
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 t = 0;
        for (int q = 0; q < nwv; ++q) t += wave_tot[q];
        block_base = t ? atomicAdd(count, t) : 0;
    }
    __syncthreads();
    u32 base = block_base;
    for (int q = 0; q < wv; ++q) base += wave_tot[q];
    // pass 2: write them, window-major inside the wavefront's slice
    const unsigned long long lt = (1ull << lane) - 1ull;
    u32 carry = 0, neg;
    for (int w = 0; w < W; ++w) {
        const u32 d = next_digit(s, carry, neg);
        const unsigned long long m = __ballot(d != 0);
        if (d) {
            const u32 o = base + (u32)__popcll(m & lt);
            keys[o] = key0 + (precomp == 2 ? 0u : precomp ? (d - 1) : ((u32)w * B + d - 1));
            vals[o] = (precomp == 2 ? ((u32)w * tstride + i) * B + (d - 1) : precomp ? ((u32)w * tstride + i) : i) | ((neg ^ flip) << 31);
        }
        base += (u32)__popcll(m);
    }
}

} // namespace mg
