// Pippenger MSM, stages K7b / K8 / K9: merge of partial runs, bucket reduce (scan tiles, work-efficient front levels, cooperative
// twins), device-side fold of the window sums. Part of msm_impl.h.
#pragma once
#include "msm_common.h"

namespace mg {

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

} // namespace mg
