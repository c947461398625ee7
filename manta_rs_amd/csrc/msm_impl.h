// Pippenger variable-base MSM on gfx950 -- kernels and host orchestration, templated on the curve
// and group. Instantiated once per (curve, group) in msm_<curve>_<group>.hip.
//
// Replaces ark-ec ^0.3.0 `VariableBaseMSM::multi_scalar_mul` (4x G1 + 1x G2 per proof, SURVEY.md rows
// a-7/a-8; reached from manta-crypto/src/arkworks/groth16.rs:597). The result is the same group
// element; the schedule is GPU-native and differs from arkworks' on purpose:
//
//   K5 digits      scalar -> W signed c-bit digits -> (bucket key, base index|sign) pairs, coalesced
//   K6 sort        one device-wide radix sort of the n*W pairs by bucket key (sort.hip)
//   K7a chunks     every lane walks L consecutive sorted pairs, mixed-adding gathered bases into an
//                  XYZZ accumulator kept in VGPRs; bucket runs that start and end inside the chunk are
//                  stored straight to their bucket, the chunk's first/last run become "partials"
//   K7b merge      partials (still sorted by key) are combined by a wavefront-wide segmented scan
//                  (shuffles, no LDS, no atomics), 64 -> 2 per wave and level, until one wave is left
//   K8 reduce      sum_k (k+1) B_k per bucket window as wavefront suffix-scan + reduction per 64-bucket
//                  tile, two levels; <= a few dozen points per window go to the host
//   K9 (host)      fold those points, Horner over windows (none when the bases carry precomputed
//                  2^(c w) multiples: all windows then share ONE bucket set and no doubling is left)
//
// Load balance never depends on the scalar distribution: work is split by sorted position, not by
// bucket, so a witness that is 25 % ones (one giant bucket) costs the same as uniform scalars.
// No global atomics on points; every bucket is written exactly once; results are deterministic.
#pragma once
#include "ec_dev.h"
#include "engine.h"
#include "fpr_dev.h"
#include "host_ec.h"
#include "params_gen.h"
#include "tuning.h"
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>

namespace mg {

// A/B switch: cap the registers of the tail kernels (merge, bucket reduce) so that they fit next to the two resident
// wavefronts of an accumulate kernel of a neighbouring MSM (160 VGPRs each: 192 are left per SIMD lane)
#ifdef MG_TAIL_WAVES
#define MG_TAIL_ATTR __attribute__((amdgpu_waves_per_eu(MG_TAIL_WAVES, MG_TAIL_WAVES)))
#define MG_SERIAL_ATTR MG_TAIL_ATTR
#ifdef MG_TAIL_COOP_SLIM
#define MG_TAIL_COOP_ATTR MG_TAIL_ATTR
#else
#define MG_TAIL_COOP_ATTR
#endif
#else
#define MG_TAIL_ATTR
#define MG_TAIL_COOP_ATTR
// serial_reduce holds three points (acc, sum, the loaded item): 266 VGPRs left alone = one wavefront per SIMD; capped at
// 256 (30 spilled) two fit, and the big first level (2^19 buckets at c = 20) runs at the issue rate of two wavefronts
#define MG_SERIAL_ATTR __attribute__((amdgpu_waves_per_eu(2, 2)))
#endif

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

// --------------------------------------------------------------------------------------------
// K7a: chunk accumulate
// --------------------------------------------------------------------------------------------
// (179 VGPRs for BLS12-381 G1 -> two wavefronts per SIMD, which already saturates the integer pipe; forcing
// three through the launch bounds spills and is slower, software-prefetching the gather changes nothing; BN254 G1 needs 130
// -> three per SIMD, and asking for four -- amdgpu_waves_per_eu(4, 4): 128 VGPRs, two spilled -- changes nothing either)
// PROBE = true is the measurement twin bench.py's roofline leg runs (kernel timing on): identical but for its first wavefront
// bracketing its whole run with the shader clock counter (s_memtime) and the constant-rate wall clock -- ticks per wall-clock
// second = the clock the kernel actually ran at. A template parameter, not a run-time test: the extra live values cost the
// product kernel six VGPRs when they were an `if`.
#ifdef MG_ACC_WAVES // per translation unit: cap the accumulate kernel's registers for this many wavefronts per SIMD
#define MG_ACC_ATTR __attribute__((amdgpu_waves_per_eu(MG_ACC_WAVES, MG_ACC_WAVES)))
#else
#define MG_ACC_ATTR
#endif
template <class F, bool PROBE = false>
__global__ __launch_bounds__(256) MG_ACC_ATTR void accumulate_chunks(const u32 *__restrict__ keys, const u32 *__restrict__ vals,
                                                         u32 M, u32 L, u32 invalid, const u32 *__restrict__ bases,
                                                         u32 astride, u32 *__restrict__ buckets,
                                                         u32 *__restrict__ pkeys, u32 *__restrict__ ppts, u32 T,
                                                         const u32 *__restrict__ count, unsigned long long *__restrict__ clk,
                                                         u32 adapt) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    long long c0 = 0;
    unsigned long long w0 = 0;
    if constexpr (PROBE)
        if (t == 0) c0 = clock64(), w0 = wall_clock64();
    if (count) {
        M = *count; // compacted pairs: lanes past the last pair have nothing to do
        // adapt: the host launched ONE round of lanes (T = what the chip holds at this kernel's occupancy) without knowing how many
        // pairs survived the compaction; the chunk length that spreads them over exactly those lanes is only known here
        if (adapt) {
            const u32 l = (M + T - 1) / T;
            L = l > L ? l : L;
        }
    }
    const size_t begin = (size_t)t * L;
    size_t end = begin + L;
    if (end > M) end = M;
    u32 cur = begin < M ? keys[begin] : invalid;
    if (cur == invalid) {
        pkeys[2 * t] = invalid;
        pkeys[2 * t + 1] = invalid;
        return;
    }
    XYZZ<F> acc = XYZZ<F>::inf();
    bool first = true;
    for (size_t j = begin; j < end; ++j) {
        const u32 k = keys[j];
        if (k != cur) {
            if (first) {
                pkeys[2 * t] = cur;
                acc.store(ppts + (size_t)(2 * t) * XYZZ<F>::WORDS);
                first = false;
            } else {
                acc.store(buckets + (size_t)cur * XYZZ<F>::WORDS);
            }
            acc = XYZZ<F>::inf();
            cur = k;
            if (k == invalid) break;
        }
        const u32 v = vals[j];
        const Affine<F> p = Affine<F>::load(bases + (size_t)(v & 0x7fffffffu) * astride);
        acc.madd_throughput(p, (v >> 31) != 0);
    }
    if (first) { // the whole chunk is one run
        pkeys[2 * t] = cur;
        acc.store(ppts + (size_t)(2 * t) * XYZZ<F>::WORDS);
        pkeys[2 * t + 1] = cur;
        XYZZ<F>::inf().store(ppts + (size_t)(2 * t + 1) * XYZZ<F>::WORDS);
    } else {
        pkeys[2 * t + 1] = cur; // may be `invalid` (then the point is never read as a summand)
        acc.store(ppts + (size_t)(2 * t + 1) * XYZZ<F>::WORDS);
    }
    if constexpr (PROBE)
        if (t == 0) clk[0] = (unsigned long long)(clock64() - c0), clk[1] = wall_clock64() - w0;
}

// Round 5 -- the accumulate stage of a SINGLE-KEY MSM (full tables, one scalar vector: every pair's key is 0 and the sum of all
// table entries IS the result -- the h MSM and the G2 MSM of a single proof). accumulate_chunks leaves two partials per lane and
// the first merge level then folds 16 of them serially per lane and scans: ~22 dependent additions on the chain that bounds a
// single proof (107 of the 168 us of merge kernels behind the h accumulate: gpurun timeline, round 5). With one key no run
// detection is needed and a workgroup can sum its own lanes: the accumulators go to LDS, ONE wavefront folds four of them per
// lane and runs six butterfly levels of shuffles -- nine dependent additions in one wavefront while the other three have retired
// (the accumulate kernel is throughput-bound: a first version that ran the butterfly in all four wavefronts added 8 wave-additions
// to the 21 of the main loop and LOST 10 % of a proof; this one adds ~2.5). One partial per WORKGROUP (768 entries instead of 393 216
// for a round of three wavefronts per SIMD) and two short merge levels behind it. Over Fp2 the lanes pair up through one shuffle
// first, which halves the LDS (72-104 words per point).
template <class F> struct AccSingle {
    static constexpr int XW = XYZZ<F>::WORDS;
    static constexpr int PAIR = F::EXT ? 1 : 0;
    static constexpr int SLOTS = 256 >> PAIR, PER = SLOTS / 64;
    typedef CoopAdd<F, false> Coop; // one exchange area: the footprint decides how many workgroups a CU holds
    static constexpr size_t LDS_BYTES = ((size_t)SLOTS * XW + Coop::LDS_WORDS) * 4; // dynamic: above 64 KB for the wide fields
};
template <class F>
__global__ __launch_bounds__(256) MG_TAIL_ATTR void accumulate_single(const u32 *__restrict__ vals, u32 M, u32 L, const u32 *__restrict__ bases,
                                                                      u32 astride, u32 *__restrict__ pkeys, u32 *__restrict__ ppts, u32 T,
                                                                      const u32 *__restrict__ count, u32 adapt, u32 invalid) {
    extern __shared__ __attribute__((aligned(16))) u32 acc_single_lds[];
    constexpr int XW = AccSingle<F>::XW, PAIR = AccSingle<F>::PAIR, SLOTS = AccSingle<F>::SLOTS, PER = AccSingle<F>::PER;
    u32 *xs = acc_single_lds;            // the workgroup's accumulators
    u32 *cx = acc_single_lds + SLOTS * XW; // exchange area of the cooperative additions
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (count) {
        M = *count;
        if (adapt) {
            const u32 l = (M + T - 1) / T;
            L = l > L ? l : L;
        }
    }
    XYZZ<F> acc = XYZZ<F>::inf();
    const size_t begin = (size_t)t * L;
    size_t end = begin + L;
    if (end > M) end = M;
    if (t < T)
        for (size_t j = begin; j < end; ++j) {
            const u32 v = vals[j];
            const Affine<F> p = Affine<F>::load(bases + (size_t)(v & 0x7fffffffu) * astride);
            acc.madd_throughput(p, (v >> 31) != 0);
        }
    if ((size_t)blockIdx.x * blockDim.x * L >= M) { // (uniform) no pair reached this workgroup
        if (threadIdx.x == 0) pkeys[blockIdx.x] = invalid;
        return;
    }
    if constexpr (PAIR) acc.add(XYZZ<F>::shfl(acc, lane ^ 1));
    if (!PAIR || !(lane & 1)) acc.store(xs + (size_t)(threadIdx.x >> PAIR) * XW);
    __syncthreads();
    // from here on the four wavefronts hold IDENTICAL copies of one 64-lane problem -- lane l folds accumulators PER l .. PER l +
    // PER - 1, then six butterfly levels -- and every addition is cooperative (ec_dev.h CoopAdd: each wavefront one of the four
    // independent products of a level): a dependent addition costs ~4 product-times instead of 14
    acc = XYZZ<F>::load(xs + (size_t)(lane * PER) * XW);
#pragma unroll 1
    for (int k = 1; k < PER; ++k) AccSingle<F>::Coop::add(acc, XYZZ<F>::load(xs + (size_t)(lane * PER + k) * XW), cx, wave, lane);
#pragma unroll 1
    for (int d = 1; d < 64; d <<= 1) AccSingle<F>::Coop::add(acc, XYZZ<F>::shfl(acc, lane ^ d), cx, wave, lane);
    if (threadIdx.x == 0) {
        pkeys[blockIdx.x] = 0u;
        acc.store(ppts + (size_t)blockIdx.x * XW);
    }
}

#ifdef MG_CALIBRATION
// Calibration twin of accumulate_chunks -- compiled ONLY into -DMG_CALIBRATION builds (tools/gather_calibration.py builds
// one with tools/build_variant.sh and selects it through MANTA_LIB; the shipped library has neither this kernel nor the
// MANTA_ACC_GATHER_ONLY switch, so no environment variable can make it return wrong results): the same lanes walk the same sorted (key, value) stream and gather the same base records, but instead of
// the mixed addition every loaded word is XORed into a register. Its duration is the memory side of the accumulate
// kernel alone -- how long the random 128 B gathers from the window tables take when no field arithmetic competes --
// and its PMC FETCH_SIZE calibrates the counter for this access pattern.
template <class F>
__global__ __launch_bounds__(256) void gather_only_chunks(const u32 *__restrict__ keys, const u32 *__restrict__ vals,
                                                          u32 M, u32 L, u32 invalid, const u32 *__restrict__ bases,
                                                          u32 astride, u32 *__restrict__ pkeys, u32 T,
                                                          const u32 *__restrict__ count) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    if (count) M = *count;
    const size_t begin = (size_t)t * L;
    size_t end = begin + L;
    if (end > M) end = M;
    u32 x = 0;
    for (size_t j = begin; j < end; ++j) {
        const u32 k = keys[j];
        if (k == invalid) break;
        const u32 v = vals[j];
        const uint4 *p = reinterpret_cast<const uint4 *>(bases + (size_t)(v & 0x7fffffffu) * astride);
#pragma unroll
        for (int q = 0; q < (int)(Affine<F>::WORDS + 3) / 4; ++q) {
            const uint4 w = p[q];
            x ^= w.x ^ w.y ^ w.z ^ w.w;
        }
        x ^= k;
    }
    pkeys[2 * t] = invalid; // no partials: the later stages see an empty list
    pkeys[2 * t + 1] = invalid;
    if (x == 0x9e3779b9u) pkeys[2 * t] = invalid - 1; // keep the loads alive
}
#endif

// --------------------------------------------------------------------------------------------
// K7b: merge of partials. The partial array is a key-sorted sequence of (key, point) entries, two per
// producer (head run, tail run; a producer whose whole range was one run emits (key, sum), (key, inf)).
// Every lane first folds G consecutive entries serially -- work-efficient: one addition per entry, and none at
// all for G = 2 on accumulate output, where the pair never shares a summable key -- which leaves it with a
// head run (parked in its own consumed input slot) and a tail run, or one run that spans the lane. The wave then
// runs ONE segmented scan over the tail runs (a run only crosses a lane if that lane is a single run, so
// equality of the sorted tail keys at distance d is the segment test) and one fix-up addition for the head
// runs. Runs that end inside the wave and do not touch its first element go to their bucket; the wave's first
// and last runs become the next level's two entries. 64*G entries -> 2 per wave.
// --------------------------------------------------------------------------------------------
template <class F>
__global__ __launch_bounds__(256) MG_TAIL_ATTR void merge_partials(u32 *__restrict__ pkeys, u32 *__restrict__ ppts, u32 cnt, u32 G,
                                                      u32 invalid, int final_level, u32 *__restrict__ buckets,
                                                      u32 *__restrict__ okeys, u32 *__restrict__ opts, u32 n_waves,
                                                      u32 *__restrict__ std_final) {
    MG_PRIO_FOR(F);
    constexpr size_t XW = XYZZ<F>::WORDS;
    const u32 wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (wave >= n_waves) return;
    const size_t b = ((size_t)wave * 64 + lane) * G;
    u32 kh = invalid, kt = invalid; // keys of the lane's first and last run
    bool single = true;             // the lane holds one run only (kh == kt)
    XYZZ<F> acc = XYZZ<F>::inf();   // sum of the last run
    if (b < cnt) {
        const size_t end = b + G < cnt ? b + G : cnt;
        u32 cur = pkeys[b];
        if (cur != invalid) {
            kh = cur;
            acc = XYZZ<F>::load(ppts + b * XW);
            for (size_t j = b + 1; j < end; ++j) {
                const u32 k = pkeys[j];
                if (k != cur) {
                    if (single) { // park the head run in slot b (already consumed; pkeys[b] == kh)
                        acc.store(ppts + b * XW);
                        single = false;
                    } else {
                        acc.store(buckets + (size_t)cur * XW);
                    }
                    cur = k;
                    acc = XYZZ<F>::inf();
                    if (k == invalid) break;
                    acc = XYZZ<F>::load(ppts + j * XW);
                } else {
                    const XYZZ<F> p = XYZZ<F>::load(ppts + j * XW);
                    if (!p.is_inf()) acc.add(p);
                }
            }
            kt = cur;
        }
    }
    // inclusive segmented scan over (kt, acc)
    for (int d = 1; d < 64; d <<= 1) {
        const u32 nk = __shfl_up(kt, d, 64);
        const bool take = (lane >= d) && (nk == kt) && (kt != invalid);
        if (!__any(take)) break;
        const XYZZ<F> o = XYZZ<F>::shfl(acc, lane - d < 0 ? lane : lane - d);
        if (take) acc.add(o);
    }
    const u32 prev_kt = __shfl_up(kt, 1, 64), next_kh = __shfl_down(kh, 1, 64);
    const u32 key0 = __shfl(kh, 0, 64);
    const bool need_in = !single && lane > 0 && prev_kt == kh; // the previous lane's last run flows into my head run
    const bool any_in = __any(need_in);
    XYZZ<F> prev = XYZZ<F>::inf();
    if (any_in) prev = XYZZ<F>::shfl(acc, lane > 0 ? lane - 1 : 0);
    // the run that ends at this lane's right edge
    const bool cont = lane < 63 && next_kh == kt;
    if (kt != invalid && !cont) {
        if (final_level) {
            // (std_final: one key in all -- a single MSM on full tables --, the last run IS the result: it leaves in the host's format)
            if (std_final) acc.store_std(std_final + (size_t)kt * XYZZ<typename F::Std>::WORDS);
            else acc.store(buckets + (size_t)kt * XW);
        } else if (kt == key0) {
            okeys[2 * wave] = kt;
            acc.store(opts + (size_t)(2 * wave) * XW);
            if (lane == 63) { // the whole wave is one run
                okeys[2 * wave + 1] = kt;
                XYZZ<F>::inf().store(opts + (size_t)(2 * wave + 1) * XW);
            }
        } else if (lane == 63) {
            okeys[2 * wave + 1] = kt;
            acc.store(opts + (size_t)(2 * wave + 1) * XW);
        } else {
            acc.store(buckets + (size_t)kt * XW);
        }
    }
    if (!final_level && lane == 63 && kt == invalid) { // all further entries are invalid too (sorted last)
        okeys[2 * wave + 1] = invalid;
        if (key0 == invalid) okeys[2 * wave] = invalid;
    }
    // the head run of a lane with several runs ends inside the lane
    if (!single) {
        XYZZ<F> h = XYZZ<F>::load(ppts + b * XW);
        if (need_in) h.add(prev);
        if (!final_level && kh == key0) {
            okeys[2 * wave] = kh;
            h.store(opts + (size_t)(2 * wave) * XW);
        } else if (final_level && std_final) {
            h.store_std(std_final + (size_t)kh * XYZZ<typename F::Std>::WORDS);
        } else {
            h.store(buckets + (size_t)kh * XW);
        }
    }
}

// K7b with cooperative additions (CoopAdd, ec_dev.h): one 64-entry-wide "logical wave" per 256-thread workgroup,
// its four wavefronts hold identical copies of the lanes' state and share every addition. Same contract as
// merge_partials; used for the levels with few logical waves, which are nothing but dependent additions.
template <class F>
__global__ __launch_bounds__(256) MG_TAIL_COOP_ATTR void merge_partials_coop(u32 *__restrict__ pkeys, u32 *__restrict__ ppts, u32 cnt, u32 G,
                                                           u32 invalid, int final_level, u32 *__restrict__ buckets,
                                                           u32 *__restrict__ okeys, u32 *__restrict__ opts,
                                                           u32 *__restrict__ std_final) {
    MG_PRIO_FOR(F);
    __shared__ __attribute__((aligned(16))) u32 lds[CoopAdd<F>::LDS_WORDS];
    constexpr size_t XW = XYZZ<F>::WORDS;
    const u32 wave = blockIdx.x; // logical wave
    const int lane = threadIdx.x & 63, pw = threadIdx.x >> 6;
    const bool writer = pw == 0; // identical data in the four wavefronts: one of them stores
    const size_t b = ((size_t)wave * 64 + lane) * G;
    u32 kh = invalid, cur = invalid;
    bool single = true, live = false;
    XYZZ<F> acc = XYZZ<F>::inf();
    if (b < cnt) {
        cur = pkeys[b];
        if (cur != invalid) {
            kh = cur;
            acc = XYZZ<F>::load(ppts + b * XW);
            live = true;
        }
    }
    for (u32 off = 1; off < G; ++off) { // uniform trip count: the additions below contain barriers
        const size_t j = b + off;
        const bool have = live && j < cnt;
        const u32 k = have ? pkeys[j] : invalid;
        XYZZ<F> p = XYZZ<F>::inf();
        if (have && k != invalid) p = XYZZ<F>::load(ppts + j * XW);
        const bool same = have && k == cur;
        if (have && k != cur) { // a run ended: the first one is parked in slot b, later ones are complete
            if (writer) acc.store(single ? ppts + b * XW : buckets + (size_t)cur * XW);
            single = false;
            cur = k;
            acc = p;
            if (k == invalid) live = false;
        }
        if (__any(same && !p.is_inf())) {
            const XYZZ<F> o = same ? p : XYZZ<F>::inf();
            CoopAdd<F>::add(acc, o, lds, pw, lane);
        }
    }
    const u32 kt = cur;
    __threadfence_block(); // the parked head runs are re-read by all four wavefronts
    __syncthreads();
    // inclusive segmented scan over (kt, acc)
    for (int d = 1; d < 64; d <<= 1) {
        const u32 nk = __shfl_up(kt, d, 64);
        const bool take = (lane >= d) && (nk == kt) && (kt != invalid);
        if (!__any(take)) break;
        XYZZ<F> o = XYZZ<F>::shfl(acc, lane - d < 0 ? lane : lane - d);
        if (!take) o = XYZZ<F>::inf();
        CoopAdd<F>::add(acc, o, lds, pw, lane);
    }
    const u32 prev_kt = __shfl_up(kt, 1, 64), next_kh = __shfl_down(kh, 1, 64);
    const u32 key0 = __shfl(kh, 0, 64);
    const bool need_in = !single && lane > 0 && prev_kt == kh;
    const bool any_in = __any(need_in);
    XYZZ<F> prev = XYZZ<F>::inf();
    if (any_in) prev = XYZZ<F>::shfl(acc, lane > 0 ? lane - 1 : 0);
    const bool cont = lane < 63 && next_kh == kt;
    if (writer && kt != invalid && !cont) {
        if (final_level) {
            // (std_final: one key in all -- a single MSM on full tables --, the last run IS the result: it leaves in the host's format)
            if (std_final) acc.store_std(std_final + (size_t)kt * XYZZ<typename F::Std>::WORDS);
            else acc.store(buckets + (size_t)kt * XW);
        } else if (kt == key0) {
            okeys[2 * wave] = kt;
            acc.store(opts + (size_t)(2 * wave) * XW);
            if (lane == 63) {
                okeys[2 * wave + 1] = kt;
                XYZZ<F>::inf().store(opts + (size_t)(2 * wave + 1) * XW);
            }
        } else if (lane == 63) {
            okeys[2 * wave + 1] = kt;
            acc.store(opts + (size_t)(2 * wave + 1) * XW);
        } else {
            acc.store(buckets + (size_t)kt * XW);
        }
    }
    if (writer && !final_level && lane == 63 && kt == invalid) {
        okeys[2 * wave + 1] = invalid;
        if (key0 == invalid) okeys[2 * wave] = invalid;
    }
    // head runs
    XYZZ<F> h = XYZZ<F>::inf();
    if (!single) h = XYZZ<F>::load(ppts + b * XW);
    if (any_in) {
        if (!need_in) prev = XYZZ<F>::inf();
        CoopAdd<F>::add(h, prev, lds, pw, lane);
    }
    if (writer && !single) {
        if (!final_level && kh == key0) {
            okeys[2 * wave] = kh;
            h.store(opts + (size_t)(2 * wave) * XW);
        } else if (final_level && std_final) {
            h.store_std(std_final + (size_t)kh * XYZZ<typename F::Std>::WORDS);
        } else {
            h.store(buckets + (size_t)kh * XW);
        }
    }
}

// --------------------------------------------------------------------------------------------
// K8: per-tile weighted sum. For the 64 items X_0..X_63 of a tile (missing items = infinity):
//   A = sum_j X_j,  S = sum_j (j+1) X_j  -- via suffix scan (acc_j = sum_{i>=j} X_i) then sum of acc_j.
// --------------------------------------------------------------------------------------------
template <class F>
__global__ __launch_bounds__(256) MG_TAIL_ATTR void tile_reduce(const u32 *__restrict__ in, u32 seg_stride /*points*/,
                                                   u32 item_off, u32 n_items, u32 tiles_per_seg, u32 n_waves,
                                                   u32 *__restrict__ outA, u32 *__restrict__ outS, int std_out) {
    MG_PRIO_FOR(F);
    const u32 wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (wave >= n_waves) return;
    const u32 seg = wave / tiles_per_seg, tile = wave % tiles_per_seg;
    const u32 idx = tile * 64 + lane;
    XYZZ<F> acc = XYZZ<F>::inf();
    if (idx < n_items) acc = XYZZ<F>::load(in + ((size_t)seg * seg_stride + item_off + idx) * XYZZ<F>::WORDS);
    int top = 1; // lanes actually populated in this tile, rounded up to a power of two
    {
        const u32 left = n_items - tile * 64;
        const int lim = left < 64 ? (int)left : 64;
        while (top < lim) top <<= 1;
    }
    for (int d = 1; d < top; d <<= 1) { // suffix scan
        const XYZZ<F> o = XYZZ<F>::shfl(acc, lane + d > 63 ? lane : lane + d);
        if (lane + d < 64) acc.add(o);
    }
    constexpr int SW = XYZZ<typename F::Std>::WORDS; // arkworks-format words per point (host staging)
    if (lane == 0) {
        if (std_out)
            acc.store_std(outA + (size_t)wave * SW);
        else
            acc.store(outA + (size_t)wave * XYZZ<F>::WORDS);
    }
    if (outS) {
        for (int d = top >> 1; d >= 1; d >>= 1) { // tree sum of the suffix sums
            const XYZZ<F> o = XYZZ<F>::shfl(acc, lane + d > 63 ? lane : lane + d);
            if (lane < d) acc.add(o);
        }
        if (lane == 0) {
            if (std_out)
                acc.store_std(outS + (size_t)wave * SW);
            else
                acc.store(outS + (size_t)wave * XYZZ<F>::WORDS);
        }
    }
}

// --------------------------------------------------------------------------------------------
// K8 front level (work-efficient): every LANE walks S consecutive items from the top with a running sum,
//   A = sum_i X_i,   Sx = sum_i (i+1) X_i   (i = 0 .. S-1 inside the lane's stretch)
// -- 2 (S-1) additions for S items where the wavefront scan of tile_reduce spends 12 per item. With lane t covering
// items tS .. tS+S-1:  sum_k (k+1) X_k = sum_t Sx_t + S * sum_{t>=1} t A_t, i.e. a plain sum of the Sx_t plus S times the
// SAME weighted sum over the A_t (t >= 1), S times shorter: levels of this kernel shrink a window of 2^19 buckets (c = 20)
// to a few thousand items for the scan kernels below, and make wide windows affordable (2^20 BLS12-381 G1: the c = 20
// accumulate kernel is 20 % shorter than the c = 16 one, and the scan-only reduce gave all of it back).
// outS == nullptr: plain partial sums (one addition per item), used for the sums of the Sx arrays.
// --------------------------------------------------------------------------------------------
template <class F>
__global__ __launch_bounds__(256) MG_SERIAL_ATTR void serial_reduce(const u32 *__restrict__ in, u32 seg_stride /*points*/, u32 item_off,
                                                     u32 n_items, u32 S, u32 lanes_per_seg, u32 n_lanes,
                                                     u32 *__restrict__ outA, u32 *__restrict__ outS) {
    MG_PRIO_FOR(F);
    constexpr size_t XW = XYZZ<F>::WORDS;
    const u32 g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_lanes) return;
    const u32 seg = g / lanes_per_seg, l = g % lanes_per_seg;
    const u32 i0 = l * S;
    u32 i1 = i0 + S;
    if (i1 > n_items) i1 = n_items;
    const u32 *base = in + ((size_t)seg * seg_stride + item_off) * XW;
    XYZZ<F> acc = XYZZ<F>::inf(), sum = XYZZ<F>::inf();
    for (u32 i = i1; i-- > i0;) {
        const XYZZ<F> x = XYZZ<F>::load(base + (size_t)i * XW);
        acc.add(x);
        if (outS) sum.add(acc);
    }
    acc.store(outA + (size_t)g * XW);
    if (outS) sum.store(outS + (size_t)g * XW);
}

// serial_reduce with cooperative additions (CoopAdd, ec_dev.h): one 64-lane logical wave per 256-thread workgroup, whose four
// wavefronts hold identical copies and share every addition (4 product-times instead of 14). For the levels with few lanes --
// from the second level on the front levels are chains of 2 (S-1) dependent additions and nothing else.
template <class F>
__global__ __launch_bounds__(256) MG_TAIL_COOP_ATTR void serial_reduce_coop(const u32 *__restrict__ in, u32 seg_stride /*points*/, u32 item_off,
                                                          u32 n_items, u32 S, u32 lanes_per_seg, u32 n_lanes,
                                                          u32 *__restrict__ outA, u32 *__restrict__ outS) {
    MG_PRIO_FOR(F);
    __shared__ __attribute__((aligned(16))) u32 lds[CoopAdd<F>::LDS_WORDS];
    constexpr size_t XW = XYZZ<F>::WORDS;
    const int lane = threadIdx.x & 63, pw = threadIdx.x >> 6;
    const u32 g = blockIdx.x * 64 + lane;
    const bool live = g < n_lanes;
    const u32 seg = live ? g / lanes_per_seg : 0, l = live ? g % lanes_per_seg : 0;
    const u32 i0 = l * S;
    const u32 *base = in + ((size_t)seg * seg_stride + item_off) * XW;
    XYZZ<F> acc = XYZZ<F>::inf(), sum = XYZZ<F>::inf();
    for (u32 j = S; j-- > 0;) { // uniform trip count: the additions contain barriers
        const u32 i = i0 + j;
        XYZZ<F> x = XYZZ<F>::inf();
        if (live && i < n_items) x = XYZZ<F>::load(base + (size_t)i * XW);
        CoopAdd<F>::add(acc, x, lds, pw, lane);
        if (outS) CoopAdd<F>::add(sum, acc, lds, pw, lane);
    }
    if (live && pw == 0) {
        acc.store(outA + (size_t)g * XW);
        if (outS) sum.store(outS + (size_t)g * XW);
    }
}

// The same with the additions spread over the four wavefronts of the workgroup (CoopAdd, ec_dev.h): one tile per
// workgroup, every wave holds the same 64 items. For the few-tile reduces of proof-sized MSMs, where the kernel is
// nothing but a chain of dependent additions.
template <class F>
__global__ __launch_bounds__(256) MG_TAIL_COOP_ATTR void tile_reduce_coop(const u32 *__restrict__ in, u32 seg_stride /*points*/,
                                                        u32 item_off, u32 n_items, u32 tiles_per_seg,
                                                        u32 *__restrict__ outA, u32 *__restrict__ outS, int std_out) {
    MG_PRIO_FOR(F);
    __shared__ __attribute__((aligned(16))) u32 lds[CoopAdd<F>::LDS_WORDS];
    const u32 tile_id = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u32 seg = tile_id / tiles_per_seg, tile = tile_id % tiles_per_seg;
    const u32 idx = tile * 64 + lane;
    XYZZ<F> acc = XYZZ<F>::inf();
    if (idx < n_items) acc = XYZZ<F>::load(in + ((size_t)seg * seg_stride + item_off + idx) * XYZZ<F>::WORDS);
    int top = 1;
    {
        const u32 left = n_items - tile * 64;
        const int lim = left < 64 ? (int)left : 64;
        while (top < lim) top <<= 1;
    }
    for (int d = 1; d < top; d <<= 1) { // suffix scan
        XYZZ<F> o = XYZZ<F>::shfl(acc, lane + d > 63 ? lane : lane + d);
        if (lane + d >= 64) o = XYZZ<F>::inf();
        CoopAdd<F>::add(acc, o, lds, wave, lane);
    }
    constexpr int SW = XYZZ<typename F::Std>::WORDS;
    if (threadIdx.x == 0) {
        if (std_out)
            acc.store_std(outA + (size_t)tile_id * SW);
        else
            acc.store(outA + (size_t)tile_id * XYZZ<F>::WORDS);
    }
    if (outS) {
        for (int d = top >> 1; d >= 1; d >>= 1) { // tree sum of the suffix sums
            XYZZ<F> o = XYZZ<F>::shfl(acc, lane + d > 63 ? lane : lane + d);
            if (lane >= d) o = XYZZ<F>::inf();
            CoopAdd<F>::add(acc, o, lds, wave, lane);
        }
        if (threadIdx.x == 0) {
            if (std_out)
                acc.store_std(outS + (size_t)tile_id * SW);
            else
                acc.store(outS + (size_t)tile_id * XYZZ<F>::WORDS);
        }
    }
}

// Second (last) reduce level for 2 <= T0 <= 64 tiles per window, ONE launch, two wavefronts per window on
// different SIMDs: wave 0 turns the tile totals A_t into X = sum_{t>=1} t*A_t (suffix scan + tree sum, only
// ceil(log2 T0) steps each), wave 1 sums the S_t. The host gets (X, sumS): window sum = sumS + 64*X.
// Together with the level-0 tile_reduce that is 12 + 2*log2(T0) dependent additions (20 for B = 1024)
// instead of 36 over three launches -- on a latency-bound tail the depth is what matters.
template <class F>
__global__ __launch_bounds__(128) MG_TAIL_ATTR void reduce_level1(const u32 *__restrict__ A0, const u32 *__restrict__ S0, u32 T0,
                                                     u32 *__restrict__ out_std) {
    MG_PRIO_FOR(F);
    constexpr int XW = XYZZ<F>::WORDS;
    constexpr int SW = XYZZ<typename F::Std>::WORDS;
    const u32 seg = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int top = 1;
    while (top < (int)T0) top <<= 1;
    XYZZ<F> acc = XYZZ<F>::inf();
    if (wave == 0) { // X = sum_{t>=1} t*A_t  =  sum_{j>=1} (sum_{t>=j} A_t)
        if (lane >= 1 && lane < (int)T0) acc = XYZZ<F>::load(A0 + ((size_t)seg * T0 + lane) * XW);
        for (int d = 1; d < top; d <<= 1) {
            const XYZZ<F> o = XYZZ<F>::shfl(acc, lane + d > 63 ? lane : lane + d);
            if (lane + d < 64) acc.add(o);
        }
        if (lane == 0) acc = XYZZ<F>::inf(); // lane 0's suffix (the total) carries weight 0
    } else {
        if (lane < (int)T0) acc = XYZZ<F>::load(S0 + ((size_t)seg * T0 + lane) * XW);
    }
    for (int d = top >> 1; d >= 1; d >>= 1) {
        const XYZZ<F> o = XYZZ<F>::shfl(acc, lane + d > 63 ? lane : lane + d);
        if (lane < d) acc.add(o);
    }
    if (lane == 0) acc.store_std(out_std + ((size_t)seg * 2 + wave) * SW);
}

// reduce_level1 with cooperative additions: two 256-thread workgroups per window (blockIdx.x & 1: 0 = the X part,
// 1 = the sum of the S_t), each spreading its additions over its four wavefronts.
template <class F>
__global__ __launch_bounds__(256) MG_TAIL_COOP_ATTR void reduce_level1_coop(const u32 *__restrict__ A0, const u32 *__restrict__ S0, u32 T0,
                                                          u32 *__restrict__ out_std) {
    MG_PRIO_FOR(F);
    __shared__ __attribute__((aligned(16))) u32 lds[CoopAdd<F>::LDS_WORDS];
    constexpr int XW = XYZZ<F>::WORDS;
    constexpr int SW = XYZZ<typename F::Std>::WORDS;
    const u32 seg = blockIdx.x >> 1;
    const int part = blockIdx.x & 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int top = 1;
    while (top < (int)T0) top <<= 1;
    XYZZ<F> acc = XYZZ<F>::inf();
    if (part == 0) { // X = sum_{t>=1} t*A_t  =  sum_{j>=1} (sum_{t>=j} A_t)
        if (lane >= 1 && lane < (int)T0) acc = XYZZ<F>::load(A0 + ((size_t)seg * T0 + lane) * XW);
        for (int d = 1; d < top; d <<= 1) {
            XYZZ<F> o = XYZZ<F>::shfl(acc, lane + d > 63 ? lane : lane + d);
            if (lane + d >= 64) o = XYZZ<F>::inf();
            CoopAdd<F>::add(acc, o, lds, wave, lane);
        }
        if (lane == 0) acc = XYZZ<F>::inf(); // lane 0's suffix (the total) carries weight 0
    } else {
        if (lane < (int)T0) acc = XYZZ<F>::load(S0 + ((size_t)seg * T0 + lane) * XW);
    }
    for (int d = top >> 1; d >= 1; d >>= 1) {
        XYZZ<F> o = XYZZ<F>::shfl(acc, lane + d > 63 ? lane : lane + d);
        if (lane >= d) o = XYZZ<F>::inf();
        CoopAdd<F>::add(acc, o, lds, wave, lane);
    }
    if (threadIdx.x == 0) acc.store_std(out_std + ((size_t)seg * 2 + part) * SW);
}

// --------------------------------------------------------------------------------------------
// K9 on the device: the fold msm_finish does on the host, for one bucket window per scalar vector (bases with precomputed
// multiples). One wavefront per vector; every lane computes the same chain (a dozen additions and doublings), lane 0 stores.
// Layouts (arkworks-format XYZZ points, as staged for the host): kind 0 = the window sum itself; kind 1 = (X, sumS) pairs,
// window = sumS + 2^6 X; kind 2 = A1[T1] | S1[T1] | P0[nP] blocks over all vectors, X = sum S1 + 2^6 sum_u u A1[u],
// window = sum P0 + 2^6 X. Front levels: window = 2^tail_shift * that + sum_e 2^shift_e * extra_e.
// --------------------------------------------------------------------------------------------
struct FoldDesc {
    const u32 *tail, *extra;
    u32 kind, T1, nP, segs, n_extra, tail_shift;
    u32 extra_shift[8];
};
template <class F>
__global__ __launch_bounds__(64) void fold_windows(FoldDesc d, u32 *__restrict__ out, size_t out_stride) {
    MG_PRIO_FOR(F);
    typedef typename F::Std S;
    constexpr int SW = XYZZ<S>::WORDS;
    const u32 q = blockIdx.x;
    auto ld = [](const u32 *p) {
        const XYZZ<S> s = XYZZ<S>::load(p);
        if (s.is_inf()) return XYZZ<F>::inf();
        return XYZZ<F>{F::from_std(s.x), F::from_std(s.y), F::from_std(s.zz), F::from_std(s.zzz)};
    };
    auto pow2 = [](XYZZ<F> p, u32 k) {
        for (u32 i = 0; i < k; ++i) p = XYZZ<F>::dbl(p);
        return p;
    };
    XYZZ<F> win = XYZZ<F>::inf();
    if (d.kind == 0) {
        win = ld(d.tail + (size_t)q * SW);
    } else if (d.kind == 1) {
        win = pow2(ld(d.tail + ((size_t)q * 2 + 0) * SW), 6);
        win.add(ld(d.tail + ((size_t)q * 2 + 1) * SW));
    } else {
        const u32 *A1 = d.tail + (size_t)q * d.T1 * SW;
        const u32 *S1 = d.tail + ((size_t)d.segs * d.T1 + (size_t)q * d.T1) * SW;
        const u32 *P0 = d.tail + ((size_t)d.segs * 2 * d.T1 + (size_t)q * d.nP) * SW;
        XYZZ<F> sumS = XYZZ<F>::inf(), run = XYZZ<F>::inf(), uA = XYZZ<F>::inf();
        for (int u = (int)d.T1 - 1; u >= 0; --u) {
            sumS.add(ld(S1 + (size_t)u * SW));
            if (u >= 1) {
                run.add(ld(A1 + (size_t)u * SW));
                uA.add(run);
            }
        }
        XYZZ<F> X = pow2(uA, 6);
        X.add(sumS);
        win = pow2(X, 6);
        for (u32 u = 0; u < d.nP; ++u) win.add(ld(P0 + (size_t)u * SW));
    }
    if (d.tail_shift) win = pow2(win, d.tail_shift);
    for (u32 e = 0; e < d.n_extra; ++e) win.add(pow2(ld(d.extra + ((size_t)e * d.segs + q) * SW), d.extra_shift[e]));
    if (threadIdx.x == 0) win.store_std(out + (size_t)q * out_stride);
}

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

// --------------------------------------------------------------------------------------------
// host orchestration
// --------------------------------------------------------------------------------------------
template <class Curve, int GROUP> struct GT;
template <class Curve> struct GT<Curve, 1> {
#ifdef MG_G1_SATURATED
    typedef Fp<typename Curve::Fq> F; // 32-bit saturated limbs everywhere (A/B reference build)
#else
    typedef FpR<typename Curve::Fq> F; // internal: reduced radix, lazily reduced
#endif
    typedef Fp<typename Curve::Fq> FIO; // arkworks memory format at the ABI
    typedef host::HFp<typename Curve::Fq> HF;
};
template <class Curve> struct GT<Curve, 2> {
    // G2 on the lazily-reduced Fp2R as well. Over BLS12-381 (an XYZZ point is 112 words) the 14-limb base
    // products inside Fp2R are calls (fpr_dev.h `CALLS`): fully inlined, those kernels need 256 VGPRs + 1.4 KB of
    // scratch per lane and -- observed on MI355X, ROCm 7.2 -- do not terminate. MG_G2_SATURATED keeps the
    // canonical 32-bit Fp2 path for A/B.
#ifdef MG_G2_SATURATED
    typedef Fp2<typename Curve::Fq> F;
#else
    typedef Fp2R<typename Curve::Fq> F;
#endif
    typedef Fp2<typename Curve::Fq> FIO;
    typedef host::HFp2<typename Curve::Fq> HF;
};

static inline u32 cdiv(size_t a, size_t b) { return (u32)((a + b - 1) / b); }

// The zero-fills of an MSM launch (pair counter, bucket array or direct result, timing words) as ONE kernel of ours instead of
// hipMemsetAsync calls: inside a stream capture those become memset nodes, and a memset node of a LINEAR captured graph was found
// to replay with a wrong fill pattern once other work had gone through the runtime (round 5: profiles/r05_linear_graph_defect.txt;
// the runtime pre-builds the AQL packets of such graphs, its own fill kernel included). No node of the library's graphs is a
// runtime-generated fill any more; one launch instead of two or three also shortens the chain.
// two word ranges device -> pinned host memory, a system-scope fence, then the token (msm_launch, MsmWorkspace::notify)
static __global__ __launch_bounds__(256) void stage_and_notify_kernel(const u32 *__restrict__ src0, u32 *__restrict__ dst0, u32 n0,
                                                                      const u32 *__restrict__ src1, u32 *__restrict__ dst1, u32 n1,
                                                                      u32 *__restrict__ flag) {
    for (u32 i = threadIdx.x; i < n0; i += 256) dst0[i] = src0[i];
    for (u32 i = threadIdx.x; i < n1; i += 256) dst1[i] = src1[i];
    __threadfence_system(); // every lane's stores are visible system-wide before it reaches the barrier ...
    __syncthreads();
    if (threadIdx.x == 0 && flag) {
        __atomic_store_n(flag, 1u, __ATOMIC_RELEASE); // ... and the token goes last
        __threadfence_system();
    }
}
struct ZeroRanges {
    u32 *p[3];
    u32 n[3]; // words
};
template <class F> __global__ __launch_bounds__(256) void zero_ranges(ZeroRanges r) {
    const u32 stride = gridDim.x * 256u, i0 = blockIdx.x * 256u + threadIdx.x;
#pragma unroll
    for (int t = 0; t < 3; ++t)
        for (u32 i = i0; i < r.n[t]; i += stride) r.p[t][i] = 0u;
}

template <class Curve, int CURVE_ID, int GROUP> class GroupEngineT : public GroupEngine {
  public:
    typedef typename GT<Curve, GROUP>::F F;
    typedef typename GT<Curve, GROUP>::FIO FIO;
    typedef typename GT<Curve, GROUP>::HF HF;
    typedef host::HPoint<HF> HP;
    typedef typename Curve::Fr FrC;
    static constexpr int AW = Affine<F>::WORDS, XW = XYZZ<F>::WORDS;           // internal formats
    static constexpr int AW_IO = Affine<FIO>::WORDS, XW_IO = XYZZ<FIO>::WORDS; // arkworks formats (ABI, staging)
    static constexpr bool SAME = std::is_same<F, FIO>::value;
    // stride of one point in a BaseSet: the internal affine record padded to a multiple of 32 B (BLS12-381
    // G1: 28 -> 32 words = one 128 B line per gathered point instead of a record straddling two)
    static constexpr int AWS = SAME ? AW : (AW + 7) / 8 * 8;
    static_assert(sizeof(HP) <= sizeof(HostPoint), "HostPoint too small");

    int curve() const override { return CURVE_ID; }
    int group() const override { return GROUP; }
    int affine_words() const override { return AW_IO; }
    int xyzz_words() const override { return XW_IO; }
    int scalar_bits() const override { return FrC::BITS; }
    int base_record_bytes() const override { return AWS * 4; }
    int point_bytes(bool compressed) const override { return compressed ? HF::BYTES : 2 * HF::BYTES; }

    static HP &hp(HostPoint *p) { return *reinterpret_cast<HP *>(p); }
    static const HP &hp(const HostPoint *p) { return *reinterpret_cast<const HP *>(p); }
    void hp_set_inf(HostPoint *p) const override { hp(p) = HP::inf(); }
    void hp_from_affine(HostPoint *p, const u32 *w) const override { hp(p) = HP::from_affine_words(w); }
    void hp_from_xyzz(HostPoint *p, const u32 *w) const override { hp(p) = HP::from_xyzz_words(w); }
    void hp_add(HostPoint *a, const HostPoint *o) const override { hp(a) = HP::add(hp(a), hp(o)); }
    void hp_neg(HostPoint *p) const override { hp(p) = hp(p).neg(); }
    void hp_mul(HostPoint *p, const u64 *k4) const override { hp(p) = HP::mul(hp(p), k4, 4); }
    void hp_mul2(const HostPoint *p, const u64 *k1, const HostPoint *q, const u64 *k2, HostPoint *out) const override {
        hp(out) = HP::mul2(hp(p), k1, hp(q), k2, 4);
    }
    void *hp_table_create(const HostPoint *base) const override {
        auto *t = new host::FixedBaseTable<HP>();
        t->build(hp(base));
        return t;
    }
    void hp_table_mul(const void *table, const u64 *k4, HostPoint *out) const override {
        hp(out) = static_cast<const host::FixedBaseTable<HP> *>(table)->mul(k4);
    }
    void hp_table_free(void *table) const override { delete static_cast<host::FixedBaseTable<HP> *>(table); }
    void hp_to_affine(const HostPoint *p, u32 *w) const override { hp(p).to_affine_words(w); }
    void hp_serialize(const HostPoint *p, unsigned char *out, bool compressed) const override {
        hp(p).serialize(out, compressed);
    }

    // ---------------------------------------------------------------- bases
    int bases_create(const u32 *pts_in, size_t n_in, bool src_on_device, int pre_c, BaseSet **out,
                     bool drop_infinity = false, u32 n_sets = 1) override {
        if (!pts_in || !n_in || !out || n_sets == 0 || n_in % n_sets) return MG_ERR_ARG;
        const u32 *pts = pts_in;
        size_t n = n_in;
        std::vector<u32> compact, map;
        if (drop_infinity && !src_on_device) {
            size_t kept = 0;
            for (size_t i = 0; i < n_in; ++i) {
                const u32 *q = pts_in + i * AW_IO;
                u32 x = 0;
                for (int k = 0; k < AW_IO; ++k) x |= q[k];
                kept += x != 0;
            }
            if (kept < n_in) {
                if (kept == 0) kept = 1; // keep one infinity entry so that the set is never empty
                compact.resize(kept * AW_IO, 0u);
                map.resize(kept, 0u);
                size_t o = 0;
                for (size_t i = 0; i < n_in && o < kept; ++i) {
                    const u32 *q = pts_in + i * AW_IO;
                    u32 x = 0;
                    for (int k = 0; k < AW_IO; ++k) x |= q[k];
                    if (x != 0) {
                        std::memcpy(&compact[o * AW_IO], q, AW_IO * 4);
                        map[o++] = (u32)i;
                    }
                }
                pts = compact.data();
                n = kept;
            }
        }
        prime_occupancy();
        BaseSet *bs = new BaseSet();
        bs->curve = CURVE_ID;
        bs->group = GROUP;
        bs->device = current_device();
        bs->n = n;
        bs->n_orig = n_in;
        bs->n_sets = n_sets;
        bs->set_len = n_in / n_sets;
        if (n_sets > 1 && n_sets <= BaseSet::MAX_SETS) { // where every query starts among the stored points
            for (u32 q = 0; q <= n_sets; ++q) {
                const size_t first = (size_t)q * bs->set_len; // original index
                bs->set_first[q] = map.empty() ? (u32)(first < n ? first : n)
                                               : (u32)(std::lower_bound(map.begin(), map.end(), (u32)first) - map.begin());
            }
            bs->set_first[n_sets] = (u32)n;
        }
        if (!map.empty()) {
            if (hipMalloc((void **)&bs->d_map, map.size() * 4) != hipSuccess ||
                hipMemcpy(bs->d_map, map.data(), map.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
                bases_destroy(bs);
                return MG_ERR_OOM;
            }
        }
        // pre_c < 0: FULL tables of window width -pre_c -- besides 2^(c w) P every multiple m 2^(c w) P, m = 1 .. 2^(c-1), so that
        // a signed digit addresses its summand directly and the MSM is one plain sum: no buckets, no sort, no bucket reduce
        const bool full = pre_c < 0;
        if (full) pre_c = -pre_c;
        int W = 1;
        if (pre_c > 0) {
            W = (FrC::BITS + pre_c - 1) / pre_c; // digits_kernel: |k| < 2^(BITS - 1)
            bs->pre_c = pre_c;
            bs->pre_W = W;
            bs->full = full;
        }
        const u32 FB = full ? 1u << (pre_c - 1) : 1u; // table entries per (window, base)
        if (full && (pre_c < 2 || pre_c > 12 || (size_t)W * n * FB >= ((size_t)1 << 31))) {
            bases_destroy(bs);
            return MG_ERR_ARG;
        }
        bs->bytes = (size_t)W * n * FB * AWS * 4;
        hipError_t e = hipMalloc((void **)&bs->d_pts, bs->bytes);
        u32 *win_pts = nullptr; // full: the window tables are an intermediate, freed below
        if (e == hipSuccess && full) e = hipMalloc((void **)&win_pts, (size_t)W * n * AWS * 4);
        if (e != hipSuccess) {
            bases_destroy(bs);
            set_last_hip_error(e, "hipMalloc(bases)", __FILE__, __LINE__);
            return MG_ERR_OOM;
        }
        struct FreeWin {
            u32 *&p;
            ~FreeWin() {
                if (p) hipFree(p);
            }
        } free_win{win_pts};
        u32 *const dst = full ? win_pts : bs->d_pts;
        if (SAME) {
            e = hipMemcpy(dst, pts, n * AW * 4, src_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice);
        } else { // convert arkworks limbs -> internal representation on the device
            u32 *stage = nullptr;
            const u32 *src = pts;
            e = hipSuccess;
            if (!src_on_device) {
                e = hipMalloc((void **)&stage, n * AW_IO * 4);
                if (e == hipSuccess) e = hipMemcpy(stage, pts, n * AW_IO * 4, hipMemcpyHostToDevice);
                src = stage;
            }
            if (e == hipSuccess) {
                hipLaunchKernelGGL((bases_to_internal<F>), dim3(cdiv(n, 256)), dim3(256), 0, 0, src, n, dst, (u32)AWS);
                e = hipDeviceSynchronize();
            }
            if (stage) hipFree(stage);
        }
        if (e != hipSuccess) {
            bases_destroy(bs);
            set_last_hip_error(e, "upload/convert bases", __FILE__, __LINE__);
            return MG_ERR_HIP;
        }
        if (W > 1) {
            u32 *tmp = nullptr;
            const size_t cnt = (size_t)(W - 1) * n;
            e = hipMalloc((void **)&tmp, cnt * XW * 4);
            if (e != hipSuccess) {
                bases_destroy(bs);
                set_last_hip_error(e, "hipMalloc(precompute tmp)", __FILE__, __LINE__);
                return MG_ERR_OOM;
            }
            hipLaunchKernelGGL((precompute_chain<F>), dim3(cdiv(n, 256)), dim3(256), 0, 0, dst, (u32)AWS, (u32)n,
                               pre_c, W, tmp);
            constexpr int KB = 16;
            hipLaunchKernelGGL((xyzz_to_affine_batch<F, KB>), dim3(cdiv(cdiv(cnt, KB), 256)), dim3(256), 0, 0, tmp,
                               cnt, dst + n * AWS, (u32)AWS);
            e = hipDeviceSynchronize();
            hipFree(tmp);
            if (e != hipSuccess) {
                bases_destroy(bs);
                set_last_hip_error(e, "precompute kernels", __FILE__, __LINE__);
                return MG_ERR_HIP;
            }
        }
        if (full) { // expand the window tables, a slice of (window, base) pairs at a time (<= 512 MB of XYZZ points in flight)
            u32 *const final_pts = bs->d_pts;
            const size_t pairs = (size_t)W * n;
            size_t slice = ((size_t)512 << 20) / ((size_t)FB * XW * 4);
            if (slice < 256) slice = 256;
            if (slice > pairs) slice = pairs;
            u32 *tmp = nullptr;
            e = hipMalloc((void **)&tmp, slice * FB * XW * 4);
            constexpr int KBF = 64; // one Fermat inversion per 64 points
            for (size_t j0 = 0; e == hipSuccess && j0 < pairs; j0 += slice) {
                const size_t cntp = pairs - j0 < slice ? pairs - j0 : slice;
                hipLaunchKernelGGL((full_table_chain<F>), dim3(cdiv(cntp, 256)), dim3(256), 0, 0, win_pts, (u32)AWS, j0, (u32)cntp, FB,
                                   tmp);
                hipLaunchKernelGGL((xyzz_to_affine_batch<F, KBF>), dim3(cdiv(cdiv(cntp * FB, KBF), 256)), dim3(256), 0, 0, tmp,
                                   cntp * FB, final_pts + j0 * FB * AWS, (u32)AWS);
                e = hipGetLastError();
            }
            if (e == hipSuccess) e = hipDeviceSynchronize();
            else (void)hipDeviceSynchronize();
            if (tmp) hipFree(tmp);
            if (e != hipSuccess) {
                bases_destroy(bs);
                set_last_hip_error(e, "full-table kernels", __FILE__, __LINE__);
                return e == hipErrorOutOfMemory ? MG_ERR_OOM : MG_ERR_HIP;
            }
        }
        *out = bs;
        return MG_OK;
    }
    void bases_destroy(BaseSet *bs) override {
        if (!bs) return;
        if (bs->d_map) hipFree(bs->d_map);
        if (bs->d_pts) hipFree(bs->d_pts);
        delete bs;
    }

    // ---------------------------------------------------------------- plan
    MsmPlan plan_for(const BaseSet *bs, size_t n, int c_override, u32 batch = 1) const override {
        MsmPlan p;
        if (bs->pre_c > 0) {
            p.c = bs->pre_c;
            p.W = bs->pre_W;
            p.precomp = true;
            p.full = bs->full;
            p.Wb = 1;
        } else {
            int lg = 0;
            while (((size_t)1 << lg) < n) ++lg;
            // (2^20 plain bases, one MSM at a time: c = 14 / 15 / 16 / 17 -> 4.98 / 4.77 / 4.48 / 5.02 ms with the front levels of the
            // bucket reduce, which 16 windows of 32 768 buckets need: profiles/r03_plain_bases_sweep.txt)
            int c = c_override > 0 ? c_override : (lg <= 8 ? 5 : lg <= 12 ? 8 : lg <= 15 ? 10 : lg <= 18 ? 12 : lg <= 19 ? 14 : 16);
            p.c = c;
            p.W = (FrC::BITS + c - 1) / c;
            p.Wb = p.W;
        }
        p.B = 1u << (p.c - 1);
        // entries per lane. Large MSMs: the grid is a whole number of rounds of 2 wavefronts per SIMD (256 CUs x
        // 4 SIMDs x 2 x 64 = 131 072 lanes) -- the accumulate kernel holds two waves per SIMD, so 1.5 rounds leave
        // half the SIMDs idle for a third of the kernel (measured at 2^20, stand-alone kernel: L = 128 -> 2.94 ms,
        // L = 170 (1536 waves) -> 3.71 ms, L = 192 -> 4.14 ms; 328 / 322 / 327 Mscalar/s pipelined, 248 / 212 / 194
        // one MSM at a time). Longer chunks mean fewer partials for the merge levels, hence as few rounds as keep
        // L <= 192. Proof-sized MSMs are latency chains -- L mixed additions, then the merge levels -- and shorter
        // chunks shorten the chain (PrivateTransfer: L = 4 / 6 / 8 / 11 / 16 -> 581 / 610 / 595 / 573 / 564
        // proofs/s), so they get twice the lanes, never fewer than 6 entries each. Batched proofs: most digit
        // entries are invalid (sorted last), so the lanes are kept plentiful (L <= 96).
        const size_t M = n * (size_t)p.W * batch;
        size_t L;
        if (M < ((size_t)8 << 20)) {
            L = M / (192 * 1024);
            if (L < 6) L = 6;
            // (full tables: no sort and no bucket reduce behind the merge levels any more, and the balance moves to short chunks for
            // all five MSMs of a proof -- PrivateTransfer, sequential proof, 300 proofs per run, same box: L = 1 / 2 / 3 / 4 / 5 / 6 ->
            // 0.98-1.02 / 0.93-0.98 / 0.87-0.89 / 0.89-0.93 / 0.90-0.93 / 0.91-0.92 ms)
            if (p.full) L = 3;
        } else {
            const size_t round = 128 * 1024, lmax = batch > 1 ? 96 : 192;
            const size_t rounds = (M + round * lmax - 1) / (round * lmax);
            L = (M + round * rounds - 1) / (round * rounds);
        }
        if (const int l = ab_knob("MANTA_MSM_L", 0); l > 0) L = (size_t)l;
        p.L = (u32)L;
        return p;
    }

    // lanes of one full round of the accumulate kernel: what the device holds at the kernel's own occupancy (single MSMs: the
    // shortest chain) or at two wavefronts per SIMD (batched passes: that saturates the integer pipe, and fewer lanes mean fewer
    // partials to merge). MANTA_ACC_ROUND_WAVES = wavefronts per SIMD, 0 = off (host-side chunk length only).
    u32 acc_round_lanes(u32 batch, bool single = false) {
        static const int knob = [] {
            return ab_knob("MANTA_ACC_ROUND_WAVES", -1);
        }();
        if (knob == 0) return 0;
        const int dev = current_device();
        if (dev < 0 || dev >= 64 || !occ_[dev].cus.load(std::memory_order_acquire)) return 0; // (primed by bases_create)
        u32 w = single && occ_[dev].blocks_single ? occ_[dev].blocks_single : occ_[dev].blocks; // 256-thread blocks per CU = wavefronts per SIMD
        if (knob > 0) w = (u32)knob < w ? (u32)knob : w;
        else if (batch > 1 && w > 2) w = 2;
        return w * 256u * occ_[dev].cus.load(std::memory_order_relaxed);
    }
    struct Occ {
        u32 blocks = 0, blocks_single = 0; // accumulate_chunks / accumulate_single (more registers, LDS: its own round size)
        std::atomic<u32> cus{0};
    } occ_[64];
    // (asked once per device outside any stream capture: bases_create runs before the first MSM on its device)
    void prime_occupancy() {
        const int dev = current_device();
        if (dev < 0 || dev >= 64 || occ_[dev].cus.load(std::memory_order_acquire)) return;
        int nb = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, accumulate_chunks<F, false>, 256, 0) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || nb < 1 || cus < 1) {
            (void)hipGetLastError();
            return;
        }
        int nbs = 0;
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&accumulate_single<F>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)AccSingle<F>::LDS_BYTES) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&nbs, accumulate_single<F>, 256, AccSingle<F>::LDS_BYTES) != hipSuccess || nbs < 1) {
            (void)hipGetLastError();
            nbs = 0;
        }
        std::lock_guard<std::mutex> g(side_mu_);
        occ_[dev].blocks_single = (u32)nbs;
        occ_[dev].blocks = (u32)nb;
        occ_[dev].cus.store((u32)cus, std::memory_order_release);
    }

    // few tiles = a pure latency chain: spread each addition over the workgroup's four wavefronts
    static bool coop_tiles(u32 tiles) {
        static const int lim = [] {
            return ab_knob("MANTA_COOP_TILES", 64);
        }();
        return (int)tiles <= lim;
    }
    static u32 coop_waves() { // merge levels with at most this many 64-entry waves use the cooperative kernel
        static const u32 lim = [] {
            return (u32)ab_knob("MANTA_COOP_WAVES", 512);
        }();
        return lim;
    }
    // entries folded serially per lane in the first merge level. Large MSMs: 4 (throughput). Proof-sized MSMs: 16 --
    // the level then has few enough logical waves (<= coop_waves()) for the cooperative kernel, whose additions
    // cost a third: 15 cooperative serial steps + the scan beat 3 plain steps + the scan and shrink the next level
    // (PrivateTransfer: G = 4 / 8 / 16 / 32 -> 865 / 927 / 955 / 832 proofs/s).
    static u32 merge_g1(size_t M) {
        static const u32 g = [] {
            const int v = ab_knob("MANTA_MERGE_G", 0);
            return (u32)(v >= 1 && v <= 64 ? v : 0);
        }();
        if (g) return g;
        return M < ((size_t)8 << 20) ? 16u : 4u;
    }

    // front levels of the bucket reduce (serial_reduce): 2^lgS0 items per lane while a level has >= 2^18 items, 2^lgS below
    // (MANTA_RED_S0 / MANTA_RED_S; MANTA_RED_S=0: scan kernels only; unset = 3), applied while a window segment has at
    // least min_items items (MANTA_RED_MIN); 2^lgSP items per lane in the plain sums of the Sx arrays (MANTA_RED_SP), which
    // run on a side stream next to the weighted chain unless MANTA_RED_SIDE=0.
    // History (profiles/r03_window_and_tail_study.txt): the first versions -- serial chains for the plain sums, a side stream per
    // workspace -- lost 6-9 % of the pipelined rate and were off by default; c = 20 tables (accumulate kernel 19 % shorter) still do
    // not pay: the 2^19-bucket reduce is eight more dependent launches and a third sort pass.
    struct RedKnobs {
        int lgS0, lgS, lgSP;
        u32 min_items;
        bool side;
    };
    static const RedKnobs &red_knobs() {
        static const RedKnobs k = [] {
            RedKnobs r{2, -1, 3, 16384u, true}; // lgS = -1: automatic (below)
            auto env = [](const char *n, int lo, int hi, int dflt) {
                const int v = ab_knob(n, dflt);
                return v < lo ? lo : (v > hi ? hi : v);
            };
            r.lgS0 = env("MANTA_RED_S0", 1, 8, r.lgS0);
            r.lgS = env("MANTA_RED_S", -1, 8, r.lgS);
            r.lgSP = env("MANTA_RED_SP", 1, 8, r.lgSP);
            r.min_items = (u32)env("MANTA_RED_MIN", 128, 1 << 30, (int)r.min_items);
            r.side = env("MANTA_RED_SIDE", 0, 1, 1) != 0;
            return r;
        }();
        return k;
    }

    hipStream_t engine_side_stream() {
        std::lock_guard<std::mutex> g(side_mu_);
        if (!side_stream_) {
            int lo = 0, hi = 0;
            if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess ||
                hipStreamCreateWithPriority(&side_stream_, hipStreamNonBlocking, hi) != hipSuccess)
                side_stream_ = nullptr;
        }
        return side_stream_;
    }
    std::mutex side_mu_;
    hipStream_t side_stream_ = nullptr; // process lifetime

    // ---------------------------------------------------------------- launch
    int msm_launch(const BaseSet *bs, const u32 *d_scalars, size_t n, int scalar_mode, int c_override,
                   MsmWorkspace *ws, u32 batch = 1, size_t scalar_stride_words = 0, bool sparse = false) override {
        if (!bs || !d_scalars || !ws || n == 0 || n > bs->n_orig || batch == 0 || batch > 65535) return MG_ERR_ARG;
        if (bs->curve != CURVE_ID || bs->group != GROUP) return MG_ERR_ARG;
        const u32 nsets = bs->n_sets; // concatenated queries over one scalar vector: nsets results per vector
        if (nsets > 1 && n > bs->set_len) return MG_ERR_ARG;
        const size_t n_scalars = n;        // scalars supplied by the caller (indexed by original position)
        if (bs->d_map || n > bs->n || nsets > 1) n = bs->n; // entries = stored points; the kernel zips to the shorter side
        const MsmPlan pl = plan_for(bs, n, c_override, batch);
        hipStream_t s = msm_stream_of(ws);
        const size_t M = n * (size_t)pl.W * batch;
        // full tables: a digit addresses its summand, every pair of a scalar vector carries the same key and the "bucket" is the result
        const u32 KB = pl.full ? 1u : pl.B; // bucket keys per bucket window
        if (M >= (1ull << 31) || (size_t)batch * nsets * pl.Wb * KB >= (1ull << 24)) return MG_ERR_ARG;
        if (pl.full) sparse = true; // compacting digit kernel: no invalid keys, so a single MSM needs no sort at all
        const u32 seg_keys = (u32)pl.Wb * KB; // bucket keys per (scalar vector, query)
        const u32 nb = batch * nsets * seg_keys; // real buckets; key nb = INVALID
        const u32 invalid = nb;
        int rc;
        if ((rc = ws->keys_in.reserve(M * 4)) || (rc = ws->keys_out.reserve(M * 4)) ||
            (rc = ws->vals_in.reserve(M * 4)) || (rc = ws->vals_out.reserve(M * 4)))
            return rc;
        const size_t tmpb = sort_pairs_temp_bytes(M);
        if ((rc = ws->sort_tmp.reserve(tmpb))) return rc;
        if ((rc = ws->buckets.reserve((size_t)(nb + 1) * XW * 4))) return rc;
        const u32 T = cdiv(M, pl.L);
        if ((rc = ws->pkeys[0].reserve((size_t)2 * T * 4)) || (rc = ws->ppts[0].reserve((size_t)2 * T * XW * 4)))
            return rc;
        const u32 waves1 = cdiv((size_t)2 * T, 64);
        if ((rc = ws->pkeys[1].reserve((size_t)2 * waves1 * 4)) ||
            (rc = ws->ppts[1].reserve((size_t)2 * waves1 * XW * 4)))
            return rc;

        // with precomputed tables the base index is w*stride + i: table w starts bs->n points after w-1
        if ((size_t)pl.W * bs->n * (pl.full ? pl.B : 1u) >= (1ull << 31)) return MG_ERR_ARG;
        int end_bit = 1;
        while ((1u << end_bit) <= invalid) ++end_bit;
        // the fixed layout marks a zero digit with the key `invalid` = one past the last bucket; where that key alone would cost
        // the sort another 8-bit pass (2^16 buckets: c = 17 tables) the compacting digit kernel is used instead -- its second walk
        // over the digits is a fifth of a radix pass
        int end_bit_real = 1;
        while (nb > 1 && (1u << end_bit_real) <= nb - 1) ++end_bit_real;
        if ((end_bit + 7) / 8 > (end_bit_real + 7) / 8) sparse = true;
        if (sparse) end_bit = end_bit_real; // no pair carries the invalid key there
        // Several scalar vectors in the fixed layout (the dense h MSM of a batched pass): the digit kernel writes vector q's pairs
        // behind vector q - 1's, and key = q * seg_keys + bucket with seg_keys a power of two -- a stable sort by the BUCKET bits
        // (+ one value for the invalid key) keeps every (q, bucket) run contiguous and needs bits(seg_keys) + 1 bits instead of
        // bits(batch * seg_keys) + 1: 14 instead of 19 for 32 proofs at c = 14, two radix passes over 40 M pairs instead of three
        // (sort.hip sort_key). MANTA_SORT_LOW=0: the full key (A/B).
        u32 sort_mask = 0xffffffffu, sort_inv = 0xffffffffu;
        static const bool sort_low = [] {
            return ab_knob("MANTA_SORT_LOW", 1) != 0;
        }();
        if (sort_low && !sparse && batch > 1 && nsets == 1 && (seg_keys & (seg_keys - 1)) == 0) {
            int eb = 1;
            while ((1u << eb) <= seg_keys) ++eb; // keys 0 .. seg_keys - 1, and seg_keys for the invalid ones
            if ((eb + 7) / 8 < (end_bit + 7) / 8) sort_mask = seg_keys - 1, sort_inv = invalid, end_bit = eb;
        }
        // zero digits are compacted away by the digit kernel; how many pairs remain is known on the device only
        u32 *d_count = nullptr;
        if (sparse && sort_pairs_takes_device_count(end_bit)) {
            if ((rc = ws->count.reserve(256))) return rc;
            d_count = ws->count.as<u32>();
        }
        // one key in all (full tables, one scalar vector): the run the last merge level closes IS the result -- it is stored in the
        // host's format straight away (no bucket array, no reduce launch: one node fewer on the latency chain of a proof's MSM)
#ifdef MG_NO_DIRECT // A/B builds (tools/build_variant.sh)
        const bool direct = false;
#else
        const bool direct = nb == 1;
#endif
        constexpr int XWM0 = XW > XW_IO ? XW : XW_IO;
        if (direct && ((rc = ws->redA.reserve((size_t)XWM0 * 4)) || (rc = ws->redS.reserve((size_t)XWM0 * 4)))) return rc;
        ws->timed = kernel_timing() && !ws->capturing;
        if (ws->timed && !ws->h_clk) MG_HIP(hipHostMalloc((void **)&ws->h_clk, 64, hipHostMallocDefault));
        { // every zero-fill of this launch, up front (none of the targets is touched by the digit kernel or the sort)
            ZeroRanges zr{};
            zr.p[0] = d_count, zr.n[0] = d_count ? 1u : 0u;
            // direct: no pair at all means the sum is the point at infinity; else the buckets (+ the slot of the invalid key)
            zr.p[1] = direct ? ws->redS.as<u32>() : ws->buckets.as<u32>();
            zr.n[1] = direct ? (u32)XWM0 : (u32)((size_t)(nb + 1) * XW);
            zr.p[2] = ws->timed ? (u32 *)ws->h_clk : nullptr, zr.n[2] = ws->timed ? 4u : 0u;
            const u32 most = zr.n[1] > 4u ? zr.n[1] : 4u;
            hipLaunchKernelGGL((zero_ranges<F>), dim3(most > 256u * 1024u ? 1024u : cdiv(most, 256)), dim3(256), 0, s, zr);
        }
        // Compacted pairs (witness MSMs: two thirds of the digits are zero): the host sized T for all n W digits, so the pairs
        // that remain fill an arbitrary part of it -- 1.35 rounds of wavefronts for the G2 MSM of a PrivateTransfer proof, i.e. two
        // rounds of 6 dependent additions where one round of 9 does, and 1.4 wavefronts per SIMD for a batched pass where two
        // balanced ones do. Launch one round of lanes and let the kernel derive the chunk length from the pair count.
        u32 Tl = T, adapt = 0;
        u32 Lk = pl.L; // the chunk length the kernel starts from
        // single-key MSMs sum inside the workgroup: one partial per workgroup (MANTA_ACC_SINGLE=0: the general kernel, A/B)
        // MANTA_ACC_SINGLE: bit 0 = G1, bit 1 = G2. Default G1 only (sequential PrivateTransfer proofs, sparse / W / dense, two
        // alternations on one box: off 0.770 / 0.859 / 1.258 ms, G1 0.755 / 0.852 / 1.270, G2 0.749 / 0.853 / 1.286, both 0.739 /
        // 0.863 / 1.314 -- over Fp2 the cooperative additions are ~20 us each and the dense G2 chain gets longer)
        static const bool acc_single_on = [] {
            const int v = ab_knob("MANTA_ACC_SINGLE", 1);
            return ((v >> (GROUP - 1)) & 1) != 0;
        }();
        const int dev_now = current_device();
        const bool acc_single = nb == 1 && d_count && acc_single_on && !(kernel_timing() && !ws->capturing) && dev_now >= 0 && dev_now < 64 &&
                                occ_[dev_now].cus.load(std::memory_order_acquire) && occ_[dev_now].blocks_single;
        if (d_count) {
            const u32 tgt = acc_round_lanes(batch, acc_single);
            if (tgt && Tl > tgt) Tl = tgt, adapt = 1;
            // one LARGE scalar vector (host chunk length above 6: 2^20 scalars): whatever the lane count came to, the pair count
            // decides (a batched pass that fits one round keeps its host-side chunk length: measured, -12 % otherwise)
            else if (tgt && batch == 1 && pl.L > 6) adapt = 1;
            // The kernel takes max(Lk, ceil(pairs / lanes)). The host's L is sized for ALL n W digits (2^20 scalars: 120 entries per
            // lane): on a witness of which a tenth survives the compaction it left nine SIMDs in ten idle and the others walking 120
            // dependent additions -- the 2^20 BLS12-381 G2 accumulate of BASELINE configs[2] took 7.6 ms for 0.9 M pairs
            // (profiles/r04_config2_timeline.txt). With the round of lanes fixed the pair count alone decides the chunk length.
            if (adapt && Lk > 6) Lk = 6;
        }
        static const u32 dthreads_sparse = [] {
            const int v = ab_knob("MANTA_DIGITS_THREADS", 0);
            return (u32)(v == 256 || v == 512 || v == 1024 ? v : 256); // measured: 256 beats 512 and 1024 on the same box
        }();
        const u32 dthreads = d_count ? dthreads_sparse : 256u; // compacting path: fewer, larger workgroups = fewer atomics on the counter
        // Concatenated queries on full tables, ONE scalar vector (the a | b_g1 | l MSM of a single proof): every pair's key is its
        // query. One digit launch per query, in stream order, appends query 0's pairs, then query 1's, ... -- the pairs ARE sorted
        // and the radix pass over them (histogram, two scans, scatter: 135-150 us on the chain that ends a W or dense proof) is
        // not run. MANTA_Z3_SORT=1 restores the single launch + sort (A/B).
        static const bool z3_sort = [] {
            return ab_knob("MANTA_Z3_SORT", 0) != 0;
        }();
        const bool per_query = pl.full && nsets > 1 && nsets <= BaseSet::MAX_SETS && batch == 1 && d_count && !z3_sort &&
                               bs->set_first[nsets] == (u32)bs->n;
        if (per_query) {
            for (u32 q = 0; q < nsets; ++q) {
                const u32 lo = bs->set_first[q], hi = bs->set_first[q + 1];
                if (hi <= lo) continue;
                hipLaunchKernelGGL((digits_kernel<FrC>), dim3(cdiv(hi - lo, dthreads), 1), dim3(dthreads), 0, s, d_scalars, hi, pl.c, pl.W,
                                   pl.B, 2, (u32)bs->n, scalar_mode, invalid, ws->keys_in.as<u32>(), ws->vals_in.as<u32>(),
                                   (const u32 *)bs->d_map, (u32)n_scalars, scalar_stride_words, seg_keys, d_count, nsets,
                                   (u32)bs->set_len, lo);
            }
        } else
        hipLaunchKernelGGL((digits_kernel<FrC>), dim3(cdiv(n, dthreads), batch), dim3(dthreads), 0, s, d_scalars, (u32)n, pl.c, pl.W,
                           pl.B, pl.full ? 2 : (pl.precomp ? 1 : 0), (u32)bs->n, scalar_mode, invalid,
                           ws->keys_in.as<u32>(), ws->vals_in.as<u32>(), (const u32 *)bs->d_map, (u32)n_scalars,
                           scalar_stride_words, seg_keys, d_count, nsets, (u32)bs->set_len);
        batch *= nsets; // from here on every (vector, query) pair is a vector of its own: its keys, its window sums, its result
        // one key in all (a single MSM on full tables, pairs compacted): any order is sorted; one digit launch per query: sorted
        const bool no_sort = (nb == 1 || per_query) && d_count;
        const u32 *skeys = no_sort ? ws->keys_in.as<u32>() : ws->keys_out.as<u32>();
        const u32 *svals = no_sort ? ws->vals_in.as<u32>() : ws->vals_out.as<u32>();
        if (!no_sort && (rc = sort_pairs(ws->keys_in.as<u32>(), ws->keys_out.as<u32>(), ws->vals_in.as<u32>(),
                                         ws->vals_out.as<u32>(), M, end_bit, ws->sort_tmp.p, tmpb, s, d_count, sort_mask, sort_inv)))
            return rc;
        u32 *const std_final = direct ? ws->redS.as<u32>() : (u32 *)nullptr;
        if (ws->timed) MG_HIP(hipEventRecord(ws->t0, s));
#ifdef MG_CALIBRATION
        static const bool gather_only = std::getenv("MANTA_ACC_GATHER_ONLY") != nullptr; // -DMG_CALIBRATION build only (wrong results)
        if (gather_only)
            hipLaunchKernelGGL((gather_only_chunks<F>), dim3(cdiv(T, 256)), dim3(256), 0, s, ws->keys_out.as<u32>(),
                               ws->vals_out.as<u32>(), (u32)M, pl.L, invalid, bs->d_pts, (u32)AWS, ws->pkeys[0].as<u32>(), T,
                               (const u32 *)d_count);
        else
#endif
        if (acc_single)
            hipLaunchKernelGGL((accumulate_single<F>), dim3(cdiv(Tl, 256)), dim3(256), AccSingle<F>::LDS_BYTES, s, svals, (u32)M, Lk, bs->d_pts, (u32)AWS,
                               ws->pkeys[0].as<u32>(), ws->ppts[0].as<u32>(), Tl, (const u32 *)d_count, adapt, invalid);
        else
        if (ws->timed)
            hipLaunchKernelGGL((accumulate_chunks<F, true>), dim3(cdiv(Tl, 256)), dim3(256), 0, s, skeys,
                               svals, (u32)M, Lk, invalid, bs->d_pts, (u32)AWS, ws->buckets.as<u32>(),
                               ws->pkeys[0].as<u32>(), ws->ppts[0].as<u32>(), Tl, (const u32 *)d_count, ws->h_clk, adapt);
        else
            hipLaunchKernelGGL((accumulate_chunks<F, false>), dim3(cdiv(Tl, 256)), dim3(256), 0, s, skeys,
                               svals, (u32)M, Lk, invalid, bs->d_pts, (u32)AWS, ws->buckets.as<u32>(),
                               ws->pkeys[0].as<u32>(), ws->ppts[0].as<u32>(), Tl, (const u32 *)d_count, (unsigned long long *)nullptr, adapt);
        if (ws->timed) MG_HIP(hipEventRecord(ws->t1, s));
        u32 cnt = acc_single ? cdiv(Tl, 256) : 2 * Tl;
        int src = 0;
        for (int level = 0;; ++level) {
            // entries folded serially per lane: the first level is throughput-bound (as many entries as
            // accumulate lanes x 2), later ones are pure latency; <= 512 entries finish in one wave
            u32 G = level == 0 && !acc_single ? merge_g1(M) : 2;
            if (cnt <= 512) G = cnt <= 64 ? 1 : cdiv(cnt, 64);
            const u32 waves = cdiv(cdiv(cnt, G), 64);
            const int fin = waves == 1;
            if (waves <= coop_waves())
                hipLaunchKernelGGL((merge_partials_coop<F>), dim3(waves), dim3(256), 0, s, ws->pkeys[src].as<u32>(),
                                   ws->ppts[src].as<u32>(), cnt, G, invalid, fin, ws->buckets.as<u32>(),
                                   ws->pkeys[1 - src].as<u32>(), ws->ppts[1 - src].as<u32>(), std_final);
            else
                hipLaunchKernelGGL((merge_partials<F>), dim3(cdiv(waves, 4)), dim3(256), 0, s, ws->pkeys[src].as<u32>(),
                                   ws->ppts[src].as<u32>(), cnt, G, invalid, fin, ws->buckets.as<u32>(),
                                   ws->pkeys[1 - src].as<u32>(), ws->ppts[1 - src].as<u32>(), waves, std_final);
            if (fin) break;
            cnt = 2 * waves;
            src ^= 1;
        }
        // ---- bucket reduce
        const u32 segs = batch * (u32)pl.Wb;
        // what the scan kernels below reduce: (array, points per segment, first item, items); the front levels replace
        // the bucket array by their A arrays
        const u32 *rin = ws->buckets.as<u32>();
        u32 rstride = KB, roff = 0, rn = KB, tail_shift = 0, n_extra = 0;
        u32 extra_shift[MsmWorkspace::MAX_EXTRA] = {};
        hipStream_t side = nullptr; // plain sums of the front levels run beside the weighted chain (stand-alone MSMs)
        {
            const RedKnobs &rk = red_knobs();
            // (never for the MSMs of a proof slot -- ws->in_graph_slot: their launches are captured into hipGraphs, and a pass captured
            // with the front levels in it made hipGraphLaunch segfault on ROCm 7.0, the multi-branch-graph defect described in
            // runtime.cpp; eagerly launched, the batched prover gains 2-3 % from them: profiles/r03_batched_front_levels.txt)
            // On by default (MANTA_RED_S unset = 8 buckets per lane, 16 from 2^16 buckets on) wherever a window segment has >= min_items buckets: same box, three runs each,
            // 2^20 BLS12-381 G1, c = 16 tables -- scan kernels only 364-367 Mscalar/s three in flight / 3.62-3.66 ms one at a time,
            // with one front level 364-376 / 3.42-3.51; plain bases (16 windows x 32 768 buckets) 4.98 -> 4.36 ms
            // (profiles/r03_front_levels_ab.txt). The plain sums ride on ONE high-priority side stream per engine: a side stream per
            // workspace aliased the runtime's four normal-priority hardware queues and cost the pipelined rate 10-15 % by itself.
            const int lgS_eff = rk.lgS >= 0 ? rk.lgS : 3;
            // (MANTA_FRONT_IN_GRAPH, diagnosis builds only: the front levels inside a proof slot's captures -- DESIGN section 6)
            static const bool front_in_graph = ab_knob("MANTA_FRONT_IN_GRAPH", 0) != 0;
            if (lgS_eff > 0 && rn >= rk.min_items && (!ws->in_graph_slot || front_in_graph)) {
                // The side stream is for STAND-ALONE launches only, and never for a stream that is being captured. Round 6 root cause
                // (profiles/r06_front_levels_in_graph.txt): inside the forked capture of a proof slot the four G1 MSMs are four
                // branches, and the ONE side stream of the engine was forked from and joined into each of them in turn -- the
                // runtime's per-stream lists of "parallel capture streams" became cyclic (branch a <-> side <-> branch b) and
                // hipStreamEndCapture recursed over them until the stack was gone (SIGSEGV in hip::Stream::EndCapture, 25+ frames
                // of itself). That was the "pass fails" of profiles/r05_batched_ab.txt (3) and the reason behind in_graph_slot.
                hipStreamCaptureStatus cst = hipStreamCaptureStatusNone;
                const bool being_captured = hipStreamIsCapturing(s, &cst) != hipSuccess || cst != hipStreamCaptureStatusNone;
                if (!ws->capturing && !ws->run_on && !ws->in_graph_slot && !being_captured && rk.side) {
                    // ONE side stream per engine, high priority (= the runtime's other pool of hardware queues): a stream per
                    // workspace put six streams on the four normal-priority queues and cost the pipelined rate 15 % through
                    // aliasing alone, whether or not the side stream was used (measured: 308 against 365 Mscalar/s)
                    if (!(side = engine_side_stream())) return MG_ERR_HIP;
                    if (!ws->side_fork) {
                        MG_HIP(hipEventCreateWithFlags(&ws->side_fork, hipEventDisableTiming));
                        MG_HIP(hipEventCreateWithFlags(&ws->side_join, hipEventDisableTiming));
                    }
                    ws->side_stream = side; // (for the abandon paths: they drain it; not owned by the workspace)
                }
                // one level: lanes of 2^lg items; cooperative additions when the level has few lanes
                auto level = [&](hipStream_t st, const u32 *in, u32 stride, u32 off, u32 n, int lg, u32 lanes, u32 *A, u32 *Sx) {
                    const size_t nl = (size_t)segs * lanes;
                    if (cdiv(nl, 64) <= coop_waves())
                        hipLaunchKernelGGL((serial_reduce_coop<F>), dim3(cdiv(nl, 64)), dim3(256), 0, st, in, stride, off, n, 1u << lg,
                                           lanes, (u32)nl, A, Sx);
                    else
                        hipLaunchKernelGGL((serial_reduce<F>), dim3(cdiv(nl, 256)), dim3(256), 0, st, in, stride, off, n, 1u << lg,
                                           lanes, (u32)nl, A, Sx);
                };
                // two passes over the same loop: sizes first (one reservation), then the launches
                for (int pass = 0; pass < 2; ++pass) {
                    size_t used = 0; // points
                    auto take = [&](size_t pts) {
                        u32 *p = pass ? ws->front.as<u32>() + used * XW : nullptr;
                        used += pts;
                        return p;
                    };
                    const u32 *in = ws->buckets.as<u32>();
                    u32 stride = pl.B, off = 0, n = pl.B, shift = 0, ne = 0;
                    while (n >= rk.min_items && ne < (u32)MsmWorkspace::MAX_EXTRA) {
                        // the big first levels are throughput-bound: short stretches = enough lanes for two wavefronts per SIMD;
                        // below that a level is a latency chain either way and longer stretches save a level
                        // (2^16 buckets -- c = 17 tables --: 16 per lane leaves the scan kernels the 4 096 items they take at c = 16;
                        // 8 per lane left 8 192 and a non-cooperative tile kernel of 0.36 ms: 3.65-3.79 ms one MSM at a time against
                        // 3.44-3.50, 362-365 Mscalar/s three in flight against 371; 32: 357-368)
                        const int lg = (size_t)segs * n >= ((size_t)1 << 18) ? rk.lgS0 : (rk.lgS < 0 && n >= (1u << 16) ? 4 : lgS_eff);
                        const u32 lanes = cdiv(n, 1u << lg);
                        u32 *A = take((size_t)segs * lanes), *Sx = take((size_t)segs * lanes);
                        if (pass) level(s, in, stride, off, n, lg, lanes, A, Sx);
                        // plain sum of the Sx_t: serial partial sums until one tile per segment is left, then one wavefront
                        hipStream_t ps = side ? side : s;
                        if (pass && side) {
                            MG_HIP(hipEventRecord(ws->side_fork, s));
                            MG_HIP(hipStreamWaitEvent(side, ws->side_fork, 0));
                        }
                        const u32 *pin = Sx;
                        u32 pcnt = lanes;
                        while (pcnt > 64) {
                            int plg = rk.lgSP;
                            while (plg > 1 && (pcnt >> plg) < 32 && pcnt > 64u << 1) --plg; // do not shrink below a tile
                            const u32 pl2 = cdiv(pcnt, 1u << plg);
                            u32 *t = take((size_t)segs * pl2);
                            if (pass) level(ps, pin, pcnt, 0u, pcnt, plg, pl2, t, (u32 *)nullptr);
                            pin = t;
                            pcnt = pl2;
                        }
                        if (pass) {
                            u32 *dst = ws->extra.as<u32>() + (size_t)ne * segs * XW_IO;
                            if (coop_tiles(segs))
                                hipLaunchKernelGGL((tile_reduce_coop<F>), dim3(segs), dim3(256), 0, ps, pin, pcnt, 0u, pcnt, 1u, dst,
                                                   (u32 *)nullptr, 1);
                            else
                                hipLaunchKernelGGL((tile_reduce<F>), dim3(cdiv((size_t)segs, 4)), dim3(256), 0, ps, pin, pcnt, 0u, pcnt,
                                                   1u, segs, dst, (u32 *)nullptr, 1);
                        }
                        extra_shift[ne++] = shift;
                        shift += lg;
                        in = A, stride = lanes, off = 1, n = lanes - 1;
                    }
                    if (!pass) {
                        if ((rc = ws->front.reserve(used * XW * 4)) ||
                            (rc = ws->extra.reserve((size_t)MsmWorkspace::MAX_EXTRA * segs * XW_IO * 4)))
                            return rc;
                    } else {
                        rin = in, rstride = stride, roff = off, rn = n, tail_shift = shift, n_extra = ne;
                    }
                }
            }
        }
        const u32 T0 = cdiv(rn, 64);
        u32 T1 = 0, nP = 0;
        size_t stage_pts;
        constexpr int XWM = XW > XW_IO ? XW : XW_IO;
        // small results (every MSM of a proof) leave through stage_and_notify_kernel; large ones keep the runtime's copy
        auto own_stage = [&](size_t pts) { return ws->notify || (pts + (size_t)n_extra * segs) * XW_IO <= 16384; };
        if (direct) { // the last merge level left the result in redS
            stage_pts = segs;
            if ((rc = stage_reserve(ws, stage_pts * XW_IO * 4))) return rc;
            ws->d_tail = ws->redS.as<u32>();
            if (!own_stage(stage_pts)) MG_HIP(hipMemcpyAsync(ws->h_stage, ws->redS.p, stage_pts * XW_IO * 4, hipMemcpyDeviceToHost, s));
        } else if (T0 == 1) { // a single tile per window: its S is the window sum
            if ((rc = ws->redA.reserve((size_t)segs * XWM * 4)) || (rc = ws->redS.reserve((size_t)segs * XWM * 4))) return rc;
            if (coop_tiles(segs) && rn > 1) // (rn = 1, full tables: the scan kernel has no addition to make, it converts the point)
                hipLaunchKernelGGL((tile_reduce_coop<F>), dim3(segs), dim3(256), 0, s, rin, rstride, roff, rn, 1u,
                                   ws->redA.as<u32>(), ws->redS.as<u32>(), 1);
            else
                hipLaunchKernelGGL((tile_reduce<F>), dim3(cdiv((size_t)segs, 4)), dim3(256), 0, s, rin, rstride, roff, rn, 1u, segs, ws->redA.as<u32>(), ws->redS.as<u32>(), 1);
            stage_pts = segs;
            if ((rc = stage_reserve(ws, (stage_pts + (size_t)n_extra * segs) * XW_IO * 4))) return rc;
            ws->d_tail = ws->redS.as<u32>();
            if (!own_stage(stage_pts)) MG_HIP(hipMemcpyAsync(ws->h_stage, ws->redS.p, stage_pts * XW_IO * 4, hipMemcpyDeviceToHost, s));
        } else if (T0 <= 64) { // two launches: tiles, then (X, sumS) per window
            if ((rc = ws->redA.reserve((size_t)segs * T0 * XW * 4)) || (rc = ws->redS.reserve((size_t)segs * T0 * XW * 4)) ||
                (rc = ws->misc.reserve((size_t)segs * 2 * XW_IO * 4)))
                return rc;
            if (coop_tiles(segs * T0))
                hipLaunchKernelGGL((tile_reduce_coop<F>), dim3(segs * T0), dim3(256), 0, s, rin, rstride, roff, rn,
                                   T0, ws->redA.as<u32>(), ws->redS.as<u32>(), 0);
            else
                hipLaunchKernelGGL((tile_reduce<F>), dim3(cdiv((size_t)segs * T0, 4)), dim3(256), 0, s, rin, rstride, roff, rn, T0, segs * T0, ws->redA.as<u32>(), ws->redS.as<u32>(), 0);
            if (coop_tiles(segs * 2))
                hipLaunchKernelGGL((reduce_level1_coop<F>), dim3(segs * 2), dim3(256), 0, s, ws->redA.as<u32>(), ws->redS.as<u32>(),
                                   T0, ws->misc.as<u32>());
            else
                hipLaunchKernelGGL((reduce_level1<F>), dim3(segs), dim3(128), 0, s, ws->redA.as<u32>(), ws->redS.as<u32>(), T0,
                                   ws->misc.as<u32>());
            stage_pts = (size_t)segs * 2;
            if ((rc = stage_reserve(ws, (stage_pts + (size_t)n_extra * segs) * XW_IO * 4))) return rc;
            ws->d_tail = ws->misc.as<u32>();
            if (!own_stage(stage_pts)) MG_HIP(hipMemcpyAsync(ws->h_stage, ws->misc.p, stage_pts * XW_IO * 4, hipMemcpyDeviceToHost, s));
            T1 = 0xffffffffu; // marks the (X, sumS) layout for msm_finish
        } else {
            if ((rc = ws->redA.reserve((size_t)segs * T0 * XWM * 4)) || (rc = ws->redS.reserve((size_t)segs * T0 * XWM * 4)))
                return rc;
            hipLaunchKernelGGL((tile_reduce<F>), dim3(cdiv((size_t)segs * T0, 4)), dim3(256), 0, s, rin, rstride, roff, rn, T0, segs * T0, ws->redA.as<u32>(), ws->redS.as<u32>(), 0);
            T1 = cdiv(T0 - 1, 64); // level 1 over A0[1..T0-1]
            nP = cdiv(T0, 64);     // plain sums of S0[0..T0-1]
            if ((rc = ws->misc.reserve((size_t)segs * (2 * T1 + nP) * XW_IO * 4))) return rc;
            u32 *A1 = ws->misc.as<u32>();
            u32 *S1 = A1 + (size_t)segs * T1 * XW_IO;
            u32 *P0 = S1 + (size_t)segs * T1 * XW_IO;
            if (coop_tiles(segs * T1))
                hipLaunchKernelGGL((tile_reduce_coop<F>), dim3(segs * T1), dim3(256), 0, s, ws->redA.as<u32>(), T0, 1u, T0 - 1, T1,
                                   A1, S1, 1);
            else
                hipLaunchKernelGGL((tile_reduce<F>), dim3(cdiv((size_t)segs * T1, 4)), dim3(256), 0, s,
                                   ws->redA.as<u32>(), T0, 1u, T0 - 1, T1, segs * T1, A1, S1, 1);
            if (coop_tiles(segs * nP))
                hipLaunchKernelGGL((tile_reduce_coop<F>), dim3(segs * nP), dim3(256), 0, s, ws->redS.as<u32>(), T0, 0u, T0, nP, P0,
                                   (u32 *)nullptr, 1);
            else
                hipLaunchKernelGGL((tile_reduce<F>), dim3(cdiv((size_t)segs * nP, 4)), dim3(256), 0, s,
                                   ws->redS.as<u32>(), T0, 0u, T0, nP, segs * nP, P0, (u32 *)nullptr, 1);
            stage_pts = (size_t)segs * (2 * T1 + nP);
            if ((rc = stage_reserve(ws, (stage_pts + (size_t)n_extra * segs) * XW_IO * 4))) return rc;
            ws->d_tail = ws->misc.as<u32>();
            if (!own_stage(stage_pts)) MG_HIP(hipMemcpyAsync(ws->h_stage, ws->misc.p, stage_pts * XW_IO * 4, hipMemcpyDeviceToHost, s));
        }
        if (n_extra && side) { // the plain sums ran on the side stream: join
            MG_HIP(hipEventRecord(ws->side_join, side));
            MG_HIP(hipStreamWaitEvent(s, ws->side_join, 0));
        }
        if (n_extra && !own_stage(stage_pts))
            MG_HIP(hipMemcpyAsync((u32 *)ws->h_stage + stage_pts * XW_IO, ws->extra.p, (size_t)n_extra * segs * XW_IO * 4,
                                  hipMemcpyDeviceToHost, s));
        ws->tail_shift = tail_shift;
        ws->n_extra = n_extra;
        for (u32 e = 0; e < n_extra; ++e) ws->extra_shift[e] = extra_shift[e];
        ws->extra_off_pts = stage_pts;
        if (own_stage(stage_pts)) {
            // The staged points leave through a kernel of ours (no copy node of the runtime's in a captured graph), and where the host
            // polls for the end of this chain, the same kernel raises the token. The host polls *h_flag to learn that THIS chain has ended (prover.cpp finish_pass_body). Rounds 4-5 wrote the staged
            // points with one D2H copy and the token with a second one behind it in the same stream: stream order says when each
            // copy may START, not in which order two different dispatches' writes become visible to a host that polls memory -- the
            // soak (tools/soak.py, distinct assignments) caught one single proof in ~10^5 whose a / l sum was read before it had
            // arrived (A and C, or C alone, wrong; status 0). One kernel now writes the staged points to pinned memory, fences at
            // system scope, and only then writes the token.
            if (ws->notify && !ws->h_flag) {
                MG_HIP(hipHostMalloc((void **)&ws->h_flag, 64, hipHostMallocDefault));
                *ws->h_flag = 0;
            }
            hipLaunchKernelGGL(stage_and_notify_kernel, dim3(1), dim3(256), 0, s, ws->d_tail, (u32 *)ws->h_stage, (u32)(stage_pts * XW_IO),
                               (const u32 *)ws->extra.p, (u32 *)ws->h_stage + stage_pts * XW_IO, (u32)((size_t)n_extra * segs * XW_IO),
                               ws->notify ? ws->h_flag : (u32 *)nullptr);
        }
        if (!ws->capturing) MG_HIP(hipEventRecord(ws->done, s));
        MG_HIP(hipGetLastError());
        ws->plan = pl;
        ws->T1 = T1;
        ws->nP = nP;
        ws->batch = batch;
        ws->pending = 1;
        return MG_OK;
    }

    static int stage_reserve(MsmWorkspace *ws, size_t bytes) {
        if (ws->h_stage_cap >= bytes) return MG_OK;
        if (ws->h_stage) hipHostFree(ws->h_stage);
        ws->h_stage = nullptr;
        ws->h_stage_cap = 0;
        size_t cap = bytes < 65536 ? 65536 : bytes;
        MG_HIP(hipHostMalloc(&ws->h_stage, cap, hipHostMallocDefault));
        ws->h_stage_cap = cap;
        return MG_OK;
    }

    // ---------------------------------------------------------------- finish (host fold)
    int msm_finish(MsmWorkspace *ws, HostPoint *out, bool already_synced = false) override {
        if (!ws || !ws->pending) return MG_ERR_STATE;
        if (!already_synced) MG_HIP(hipEventSynchronize(ws->done));
        ws->pending = 0;
        if (ws->timed && !already_synced) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, ws->t0, ws->t1) == hipSuccess) set_last_accumulate_ms(ms);
            const unsigned long long ck[2] = {((volatile unsigned long long *)ws->h_clk)[0], ((volatile unsigned long long *)ws->h_clk)[1]};
            int khz = 0, dev = 0;
            if (ck[1] && hipGetDevice(&dev) == hipSuccess &&
                hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) == hipSuccess)
                set_last_accumulate_mhz((float)((double)ck[0] / (double)ck[1] * (double)khz / 1e3));
        }
        const MsmPlan &pl = ws->plan;
        const u32 Wb = (u32)pl.Wb, segs = ws->batch * Wb, T1 = ws->T1, nP = ws->nP;
        const u32 *st = (const u32 *)ws->h_stage;
        for (u32 q = 0; q < ws->batch; ++q) {
        HP total = HP::inf();
        for (int w = (int)((q + 1) * Wb) - 1; w >= (int)(q * Wb); --w) {
            HP win;
            if (T1 == 0xffffffffu) { // fused reduce: (X, sumS) per window, window = sumS + 64 X
                const HP X = HP::from_xyzz_words(st + ((size_t)w * 2 + 0) * XW_IO);
                const HP sumS = HP::from_xyzz_words(st + ((size_t)w * 2 + 1) * XW_IO);
                win = HP::add(sumS, HP::mul_pow2(X, 6));
            } else if (T1 == 0) { // one tile per window: the staged point is the window sum
                win = HP::from_xyzz_words(st + (size_t)w * XW_IO);
            } else {
                const u32 *A1 = st + ((size_t)w * T1) * XW_IO;
                const u32 *S1 = st + ((size_t)segs * T1 + (size_t)w * T1) * XW_IO;
                const u32 *P0 = st + ((size_t)segs * 2 * T1 + (size_t)w * nP) * XW_IO;
                // X = sum_{t>=1} t*A0[t] = sum_u ( S1[u] + 64*u*A1[u] )
                HP sumS = HP::inf(), run = HP::inf(), uA = HP::inf();
                for (int u = (int)T1 - 1; u >= 0; --u) {
                    sumS = HP::add(sumS, HP::from_xyzz_words(S1 + (size_t)u * XW_IO));
                    if (u >= 1) {
                        run = HP::add(run, HP::from_xyzz_words(A1 + (size_t)u * XW_IO));
                        uA = HP::add(uA, run); // sum_u u*A1[u]
                    }
                }
                HP X = HP::add(sumS, HP::mul_pow2(uA, 6));
                HP sumP = HP::inf();
                for (u32 u = 0; u < nP; ++u) sumP = HP::add(sumP, HP::from_xyzz_words(P0 + (size_t)u * XW_IO));
                win = HP::add(sumP, HP::mul_pow2(X, 6));
            }
            if (ws->tail_shift) win = HP::mul_pow2(win, ws->tail_shift); // front levels: window = 2^shift * tail + extras
            for (u32 e = 0; e < ws->n_extra; ++e) {
                const HP x = HP::from_xyzz_words(st + (ws->extra_off_pts + (size_t)e * segs + (size_t)w) * XW_IO);
                win = HP::add(win, ws->extra_shift[e] ? HP::mul_pow2(x, ws->extra_shift[e]) : x);
            }
            if (w != (int)((q + 1) * Wb) - 1) total = HP::mul_pow2(total, (unsigned)pl.c);
            total = HP::add(total, win);
        }
        hp(out + q) = total;
        }
        return MG_OK;
    }

    // ---------------------------------------------------------------- finish on the device
    int msm_fold_device(MsmWorkspace *ws, u32 *d_out, size_t out_stride_words, hipStream_t on = nullptr) override {
        if (!ws || !ws->pending || !d_out || !ws->d_tail) return MG_ERR_STATE;
        const MsmPlan &pl = ws->plan;
        if (pl.Wb != 1) return MG_ERR_STATE; // plain bases keep the host fold (up to 255 Horner doublings: a host job)
        FoldDesc d{};
        d.tail = ws->d_tail;
        d.extra = ws->extra.as<u32>();
        d.kind = ws->T1 == 0xffffffffu ? 1u : (ws->T1 == 0 ? 0u : 2u);
        d.T1 = ws->T1, d.nP = ws->nP, d.segs = ws->batch, d.n_extra = ws->n_extra, d.tail_shift = ws->tail_shift;
        for (u32 e = 0; e < ws->n_extra; ++e) d.extra_shift[e] = ws->extra_shift[e];
        hipStream_t s = on ? on : (msm_stream_of(ws));
        hipLaunchKernelGGL((fold_windows<F>), dim3(ws->batch), dim3(64), 0, s, d, d_out, out_stride_words);
        MG_HIP(hipGetLastError());
        return MG_OK;
    }
    int msm_discard(MsmWorkspace *ws) override {
        if (!ws) return MG_ERR_STATE;
        ws->pending = 0;
        MG_HIP(hipStreamSynchronize(msm_stream_of(ws)));
        return MG_OK;
    }

    // ---------------------------------------------------------------- fixed-base batch mul
    int fixed_base_mul(const u32 *base_affine_host, const u32 *d_scalars, size_t n, u32 *d_out_affine,
                       hipStream_t s) override {
        u32 *d_base = nullptr, *tmp = nullptr;
        MG_HIP(hipMalloc((void **)&d_base, AW_IO * 4));
        hipError_t e = hipMalloc((void **)&tmp, n * XW_IO * 4);
        if (e != hipSuccess) {
            hipFree(d_base);
            set_last_hip_error(e, "hipMalloc(fixed_base tmp)", __FILE__, __LINE__);
            return MG_ERR_OOM;
        }
        hipMemcpyAsync(d_base, base_affine_host, AW_IO * 4, hipMemcpyHostToDevice, s);
        constexpr int KB = 16;
        static const size_t table_min = [] {
            return (size_t)ab_knob("MANTA_FIXED_BASE_TABLE_MIN", 16384);
        }();
        u32 *t_xyzz = nullptr, *t_aff = nullptr;
        if (n >= table_min) { // many multiples of one base: 32 table additions each instead of ~380 group operations
            constexpr size_t TN = 32 * 255;
            if (hipMalloc((void **)&t_xyzz, TN * XW_IO * 4) == hipSuccess && hipMalloc((void **)&t_aff, TN * AW_IO * 4) == hipSuccess) {
                hipLaunchKernelGGL((fixed_base_table_kernel<FIO>), dim3(cdiv(TN, 256)), dim3(256), 0, s, d_base, t_xyzz);
                hipLaunchKernelGGL((xyzz_to_affine_batch<FIO, KB>), dim3(cdiv(cdiv(TN, KB), 256)), dim3(256), 0, s, t_xyzz, TN, t_aff,
                                   (u32)AW_IO);
                hipLaunchKernelGGL((fixed_base_mul_table_kernel<FIO>), dim3(cdiv(n, 256)), dim3(256), 0, s, t_aff, d_scalars, n, tmp);
            } else {
                (void)hipGetLastError();
                if (t_xyzz) hipFree(t_xyzz);
                t_xyzz = nullptr;
            }
        }
        if (!t_xyzz)
        hipLaunchKernelGGL((fixed_base_mul_kernel<FIO>), dim3(cdiv(n, 256)), dim3(256), 0, s, d_base, d_scalars, n, tmp);
        hipLaunchKernelGGL((xyzz_to_affine_batch<FIO, KB>), dim3(cdiv(cdiv(n, KB), 256)), dim3(256), 0, s, tmp, n,
                           d_out_affine, (u32)AW_IO);
        e = hipStreamSynchronize(s);
        hipFree(d_base);
        hipFree(tmp);
        if (t_xyzz) hipFree(t_xyzz);
        if (t_aff) hipFree(t_aff);
        if (e != hipSuccess) {
            set_last_hip_error(e, "fixed_base_mul", __FILE__, __LINE__);
            return MG_ERR_HIP;
        }
        return MG_OK;
    }

    int ec_elementwise(int op, const u32 *a_host, const u32 *b_host, size_t n, u32 *out_affine_host) override {
        return ec_elementwise_impl(op, a_host, b_host, n, out_affine_host, false);
    }
    // the same results as XYZZ points (XW_IO words each): no inversion on the device -- xyzz_batch_to_affine() turns them
    // into affine points on the host with one inversion for all of them
    int ec_elementwise_xyzz(int op, const u32 *a_host, const u32 *b_host, size_t n, u32 *out_xyzz_host) override {
        return ec_elementwise_impl(op, a_host, b_host, n, out_xyzz_host, true);
    }
    // the multiplication k_i P_i (op MG_EC_MUL) as two calls around other work: begin() uploads into the workspace's grow-only
    // scratch buffer and launches on the workspace's stream (no hipMalloc / hipFree / stream 0: nothing else on the device
    // waits for it and it waits for nothing), finish() waits and fetches the XYZZ results
    int ec_mul_xyzz_begin(const u32 *a_host, const u32 *k_host, size_t n, MsmWorkspace *ws, const u32 *glv_beta_std) override {
        if (!a_host || !k_host || !n || !ws) return MG_ERR_ARG;
        const size_t ab = n * AW_IO * 4, bb = n * 32 + (glv_beta_std ? (size_t)AW_IO / 2 * 4 : 0), tb = n * XW_IO * 4;
        int rc = ws->scratch.reserve(ab + bb + tb);
        if (rc) return rc;
        unsigned char *d = (unsigned char *)ws->scratch.p;
        hipError_t e = hipMemcpyAsync(d, a_host, ab, hipMemcpyHostToDevice, ws->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(d + ab, k_host, n * 32, hipMemcpyHostToDevice, ws->stream);
        if (e == hipSuccess && glv_beta_std)
            e = hipMemcpyAsync(d + ab + n * 32, glv_beta_std, (size_t)AW_IO / 2 * 4, hipMemcpyHostToDevice, ws->stream);
        if (e == hipSuccess) {
            hipLaunchKernelGGL((ec_elementwise_kernel<F>), dim3(cdiv(n, 256)), dim3(256), 0, ws->stream, glv_beta_std ? 6 : 3, (const u32 *)d,
                               (const u32 *)(d + ab), n, (u32 *)(d + ab + bb));
            e = hipGetLastError();
        }
        if (e != hipSuccess) {
            hipStreamSynchronize(ws->stream);
            set_last_hip_error(e, "ec_mul_xyzz", __FILE__, __LINE__);
            return MG_ERR_HIP;
        }
        return MG_OK;
    }
    const u32 *ec_mul_xyzz_device(MsmWorkspace *ws, size_t n, bool glv) const override { // where begin()'s kernel leaves the n results
        const size_t ab = n * AW_IO * 4, bb = n * 32 + (glv ? (size_t)AW_IO / 2 * 4 : 0);
        return ws && ws->scratch.p ? (const u32 *)((unsigned char *)ws->scratch.p + ab + bb) : nullptr;
    }
    int ec_mul_xyzz_finish(MsmWorkspace *ws, size_t n, u32 *out_xyzz_host, bool glv) override {
        if (!ws || !n || !out_xyzz_host) return MG_ERR_ARG;
        const size_t ab = n * AW_IO * 4, bb = n * 32 + (glv ? (size_t)AW_IO / 2 * 4 : 0), tb = n * XW_IO * 4;
        hipError_t e = hipMemcpyAsync(out_xyzz_host, (unsigned char *)ws->scratch.p + ab + bb, tb, hipMemcpyDeviceToHost, ws->stream);
        const hipError_t e2 = hipStreamSynchronize(ws->stream);
        if (e == hipSuccess) e = e2;
        if (e != hipSuccess) {
            set_last_hip_error(e, "ec_mul_xyzz", __FILE__, __LINE__);
            return MG_ERR_HIP;
        }
        return MG_OK;
    }
    void xyzz_batch_to_affine(const u32 *xyzz_host, size_t n, u32 *out_affine_host) const override {
        typedef decltype(HP{}.x) HF;
        std::vector<HF> den(n), pre(n);
        HF acc = HF::one();
        for (size_t i = 0; i < n; ++i) { // Montgomery's trick: prefix products of the denominators ZZ ZZZ (1 for infinity)
            const HP q = HP::from_xyzz_words(xyzz_host + i * XW_IO);
            den[i] = q.is_inf() ? HF::one() : HF::mul(q.zz, q.zzz);
            pre[i] = acc;
            acc = HF::mul(acc, den[i]);
        }
        HF inv = HF::inv(acc);
        for (size_t i = n; i-- > 0;) {
            const HP q = HP::from_xyzz_words(xyzz_host + i * XW_IO);
            const HF t = HF::mul(inv, pre[i]); // 1 / (ZZ ZZZ) of point i
            inv = HF::mul(inv, den[i]);
            u32 *o = out_affine_host + i * AW_IO;
            if (q.is_inf()) {
                std::memset(o, 0, AW_IO * 4);
                continue;
            }
            HF::mul(q.x, HF::mul(t, q.zzz)).store_words(o);
            HF::mul(q.y, HF::mul(t, q.zz)).store_words(o + HF::WORDS);
        }
    }
    int ec_elementwise_impl(int op, const u32 *a_host, const u32 *b_host, size_t n, u32 *out_host, bool xyzz) {
        if (op < 0 || op > 5 || !a_host || !out_host || n == 0 || (op != 2 && !b_host)) return MG_ERR_ARG;
        const size_t ab = n * AW_IO * 4, bb = op == 3 ? n * 32 : (op == 5 ? 32 : ab);
        u32 *da = nullptr, *db = nullptr, *tmp = nullptr, *dout = nullptr;
        // a stream of its own (not stream 0: a synchronous copy anywhere else in the process -- another thread creating a base
        // set, say -- would wait for this kernel, a millisecond of one-lane latency for 128-bit multipliers)
        hipStream_t st = stream_pool_get_normal(); // (pooled: the library destroys no stream, runtime.cpp)
        hipError_t e = st ? hipSuccess : hipErrorOutOfMemory;
        if (e == hipSuccess) e = hipMalloc((void **)&da, ab);
        if (e == hipSuccess) e = hipMalloc((void **)&db, bb);
        if (e == hipSuccess) e = hipMalloc((void **)&tmp, n * XW_IO * 4);
        if (e == hipSuccess && !xyzz) e = hipMalloc((void **)&dout, ab);
        if (e == hipSuccess) e = hipMemcpyAsync(da, a_host, ab, hipMemcpyHostToDevice, st);
        if (e == hipSuccess && op != 2) e = hipMemcpyAsync(db, b_host, bb, hipMemcpyHostToDevice, st);
        if (e == hipSuccess) {
            hipLaunchKernelGGL((ec_elementwise_kernel<F>), dim3(cdiv(n, 256)), dim3(256), 0, st, op, da, db, n, tmp);
            if (xyzz) {
                e = hipMemcpyAsync(out_host, tmp, n * XW_IO * 4, hipMemcpyDeviceToHost, st);
            } else {
                constexpr int KB = 16;
                hipLaunchKernelGGL((xyzz_to_affine_batch<FIO, KB>), dim3(cdiv(cdiv(n, KB), 256)), dim3(256), 0, st, tmp, n, dout,
                                   (u32)AW_IO);
                e = hipMemcpyAsync(out_host, dout, ab, hipMemcpyDeviceToHost, st);
            }
        }
        if (st) {
            const hipError_t e2 = hipStreamSynchronize(st);
            if (e == hipSuccess) e = e2;
            stream_pool_put_normal(st);
        }
        hipFree(da);
        hipFree(db);
        hipFree(tmp);
        if (dout) hipFree(dout);
        if (e != hipSuccess) {
            set_last_hip_error(e, "ec_elementwise", __FILE__, __LINE__);
            return e == hipErrorOutOfMemory ? MG_ERR_OOM : MG_ERR_HIP;
        }
        return MG_OK;
    }

    // NTT over group elements: host affine in, host affine out (natural order both); tw = the Fr domain's device twiddle
    // table (omega^k, k < n/2, Montgomery), n_inv_canonical = n^-1 for the inverse transform (nullptr: forward)
    int group_ntt(const u32 *in_affine_host, unsigned lg, const u32 *d_twiddles_mont, const u32 *n_inv_canonical,
                  u32 *out_affine_host) override {
        if (!in_affine_host || !out_affine_host || lg > 26 || (lg > 0 && !d_twiddles_mont)) return MG_ERR_ARG;
        const size_t n = (size_t)1 << lg, ab = n * AW_IO * 4;
        u32 *d_in = nullptr, *d_pts = nullptr, *d_std = nullptr, *d_out = nullptr, *d_sc = nullptr;
        hipError_t e = hipMalloc((void **)&d_in, ab);
        if (e == hipSuccess) e = hipMalloc((void **)&d_pts, n * XW * 4);
        if (e == hipSuccess) e = hipMalloc((void **)&d_std, n * XW_IO * 4);
        if (e == hipSuccess) e = hipMalloc((void **)&d_out, ab);
        if (e == hipSuccess && n_inv_canonical) e = hipMalloc((void **)&d_sc, 32);
        if (e == hipSuccess) e = hipMemcpy(d_in, in_affine_host, ab, hipMemcpyHostToDevice);
        if (e == hipSuccess && n_inv_canonical) e = hipMemcpy(d_sc, n_inv_canonical, 32, hipMemcpyHostToDevice);
        if (e == hipSuccess) {
            hipLaunchKernelGGL((group_ntt_load_kernel<F>), dim3(cdiv(n, 256)), dim3(256), 0, 0, d_in, lg, d_pts);
            for (unsigned s = 1; s <= lg; ++s)
                hipLaunchKernelGGL((group_ntt_stage_kernel<F, FrC>), dim3(cdiv(n / 2, 256)), dim3(256), 0, 0, d_pts, d_twiddles_mont, lg, s);
            hipLaunchKernelGGL((group_scale_store_kernel<F>), dim3(cdiv(n, 256)), dim3(256), 0, 0, d_pts, (const u32 *)d_sc, n, d_std);
            constexpr int KB = 16;
            hipLaunchKernelGGL((xyzz_to_affine_batch<FIO, KB>), dim3(cdiv(cdiv(n, KB), 256)), dim3(256), 0, 0, d_std, n, d_out, (u32)AW_IO);
            e = hipMemcpy(out_affine_host, d_out, ab, hipMemcpyDeviceToHost);
        }
        hipFree(d_in), hipFree(d_pts), hipFree(d_std), hipFree(d_out), hipFree(d_sc);
        if (e != hipSuccess) {
            set_last_hip_error(e, "group_ntt", __FILE__, __LINE__);
            return e == hipErrorOutOfMemory ? MG_ERR_OOM : MG_ERR_HIP;
        }
        return MG_OK;
    }

    int sum_affine(const u32 *d_pts, size_t n, HostPoint *out) override {
        const u32 T = n < 4096 ? (u32)(n ? n : 1) : 4096;
        u32 *tmp = nullptr;
        MG_HIP(hipMalloc((void **)&tmp, (size_t)T * XW_IO * 4));
        hipLaunchKernelGGL((sum_affine_kernel<FIO>), dim3(cdiv(T, 256)), dim3(256), 0, 0, d_pts, n, T, tmp);
        std::vector<u32> h((size_t)T * XW_IO);
        hipError_t e = hipMemcpy(h.data(), tmp, h.size() * 4, hipMemcpyDeviceToHost);
        hipFree(tmp);
        if (e != hipSuccess) {
            set_last_hip_error(e, "sum_affine", __FILE__, __LINE__);
            return MG_ERR_HIP;
        }
        HP acc = HP::inf();
        for (u32 t = 0; t < T; ++t) acc = HP::add(acc, HP::from_xyzz_words(h.data() + (size_t)t * XW_IO));
        hp(out) = acc;
        return MG_OK;
    }
};

} // namespace mg
