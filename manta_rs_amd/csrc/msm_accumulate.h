// Pippenger MSM, stage K7a: the bucket-accumulate kernels (the dominant kernel of the path). Part of msm_impl.h.
#pragma once
#include "msm_common.h"

namespace mg {

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

} // namespace mg
