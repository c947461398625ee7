// K6: sort of the (bucket key, base index) pairs by key -- a hand-written LSD radix sort for gfx950.
//
// The general-purpose device sort (hipCUB / rocPRIM) takes 0.8 ms for the 16.7 M pairs of a 2^20 MSM. This one
// is specialised to what the MSM needs -- 32-bit pairs, keys of at most 24 significant bits (usually 16 -> two
// 8-bit passes), stable, no temporary beyond a per-tile histogram -- and needs ~0.45 ms of kernel time; in the
// pipelined bench (three MSMs in flight) the two are equal within noise because the step is bounded by the
// accumulate kernel. Per pass:
//   radix_hist     per 4096-element tile, digit counts in LDS (ds_add) -> hist[digit][tile]
//   radix_scan_*   exclusive scan over hist (digit-major): one workgroup per digit row, then the 256 row totals
//   radix_scatter  every wavefront ranks its 1024 elements 64 at a time with ballots: the lanes holding
//                  the same digit are found by intersecting eight v_cmp/ballot masks (wave64 "match"),
//                  their rank is a popcount of the lanes before them plus the wave's running count for the
//                  digit; no atomics, stable by construction. The tile is then laid out digit-sorted in LDS so
//                  that consecutive lanes store to consecutive addresses within each digit run.
// 20 B of HBM traffic per element per pass. The only sort on the product path: keys wider than 32 bits' worth of
// 8-bit passes do not occur (msm_launch caps the bucket-key space at 2^24), and no library sort is linked.
#include "engine.h"
#include <cstdlib>

namespace mg {

static constexpr int SORT_TILE = 4096; // elements per workgroup: 4 waves x 16 items x 64 lanes
static constexpr int SORT_ITEMS = 16;

// The key a pass sorts by. Normally the key itself. Batched MSMs in the fixed layout arrive proof-major (vector q's pairs, then
// vector q + 1's) with key = q * seg + bucket, seg a power of two: a STABLE sort by the bucket bits alone leaves every (q, bucket)
// run contiguous -- order (bucket, q) instead of (q, bucket), which the run-detecting consumers do not care about -- and saves the
// passes over the vector bits (the dense h MSM of 32 proofs: 14 bits = two passes instead of 19 = three over 40 M pairs). Keys
// from `inv_from` up (the invalid key of zero digits) sort behind every bucket. lowmask = inv_from = ~0: the identity.
__device__ __forceinline__ u32 sort_key(u32 k, u32 lowmask, u32 inv_from) { return k >= inv_from ? lowmask + 1u : (k & lowmask); }

// (`count`: when non-null the number of elements is read from the device -- the MSM's digit kernel compacts
// away zero digits and only it knows how many pairs are left; the grid is sized for the maximum and tiles
// beyond the count contribute zero histograms and no elements)
__global__ __launch_bounds__(256) void radix_hist(const u32 *__restrict__ keys, u32 M, int shift, u32 ntiles,
                                                  u32 *__restrict__ hist, const u32 *__restrict__ count, u32 lowmask, u32 inv_from) {
    MG_PRIO_HIGH();
    if (count) M = *count;
    if ((size_t)blockIdx.x * SORT_TILE >= M) { // a tile past the last element: an all-zero histogram column
        hist[(size_t)threadIdx.x * ntiles + blockIdx.x] = 0;
        return;
    }
    __shared__ u32 cnt[256];
    cnt[threadIdx.x] = 0;
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * SORT_TILE;
#pragma unroll
    for (int j = 0; j < SORT_ITEMS; ++j) {
        const size_t e = base + (size_t)j * 256 + threadIdx.x;
        if (e < M) atomicAdd(&cnt[(sort_key(keys[e], lowmask, inv_from) >> shift) & 255u], 1u);
    }
    __syncthreads();
    hist[(size_t)threadIdx.x * ntiles + blockIdx.x] = cnt[threadIdx.x];
}

// exclusive scan of every digit row hist[d][0..ntiles) in place (one workgroup per digit) + the row total
__global__ __launch_bounds__(256) void radix_scan_rows(u32 *__restrict__ hist, u32 ntiles, u32 *__restrict__ totals) {
    MG_PRIO_HIGH();
    __shared__ u32 part[256];
    u32 *row = hist + (size_t)blockIdx.x * ntiles;
    const u32 per = (ntiles + 255) / 256;
    const u32 lo = threadIdx.x * per, hi = lo + per < ntiles ? lo + per : ntiles;
    u32 s = 0;
    for (u32 i = lo; i < hi; ++i) s += row[i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        const u32 v = threadIdx.x >= (u32)d ? part[threadIdx.x - d] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    u32 run = threadIdx.x ? part[threadIdx.x - 1] : 0;
    for (u32 i = lo; i < hi; ++i) {
        const u32 c = row[i];
        row[i] = run;
        run += c;
    }
    if (threadIdx.x == 255) totals[blockIdx.x] = part[255];
}
// exclusive scan of the 256 row totals in place
__global__ __launch_bounds__(256) void radix_scan_totals(u32 *__restrict__ totals) {
    MG_PRIO_HIGH();
    __shared__ u32 part[256];
    const u32 mine = totals[threadIdx.x];
    part[threadIdx.x] = mine;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        const u32 v = threadIdx.x >= (u32)d ? part[threadIdx.x - d] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    totals[threadIdx.x] = part[threadIdx.x] - mine;
}

__global__ __launch_bounds__(256) void radix_scatter(const u32 *__restrict__ keys_in, const u32 *__restrict__ vals_in,
                                                     u32 *__restrict__ keys_out, u32 *__restrict__ vals_out, u32 M,
                                                     int shift, u32 ntiles, const u32 *__restrict__ hist,
                                                     const u32 *__restrict__ totals, const u32 *__restrict__ count, u32 lowmask,
                                                     u32 inv_from) {
    MG_PRIO_HIGH();
    if (count) M = *count;
    if ((size_t)blockIdx.x * SORT_TILE >= M) return; // a tile past the last element
    __shared__ u32 wcount[4][256]; // per-wave running digit counts, then per-wave tile-local offsets
    __shared__ u32 gbase[256];     // global position of the tile's first element of each digit, minus its local offset
    __shared__ u32 sk[SORT_TILE], sv[SORT_TILE]; // the tile, locally sorted by digit (stable)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int w = 0; w < 4; ++w) wcount[w][threadIdx.x] = 0;
    __syncthreads();
    const size_t tile0 = (size_t)blockIdx.x * SORT_TILE;
    const size_t base = tile0 + (size_t)wave * (SORT_ITEMS * 64);
    u32 k[SORT_ITEMS], v[SORT_ITEMS];
    unsigned short rk[SORT_ITEMS];
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
#pragma unroll
    for (int j = 0; j < SORT_ITEMS; ++j) {
        const size_t e = base + (size_t)j * 64 + lane;
        const bool valid = e < M;
        k[j] = valid ? keys_in[e] : 0u;
        v[j] = valid ? vals_in[e] : 0u;
        const u32 d = (sort_key(k[j], lowmask, inv_from) >> shift) & 255u;
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bool bit = (d >> b) & 1u;
            const unsigned long long bb = __ballot(bit);
            peers &= bit ? bb : ~bb;
        }
        const u32 before = wcount[wave][d]; // every peer reads the count, then the first peer bumps it
        const u32 r = (u32)__popcll(peers & lt_mask);
        rk[j] = (unsigned short)(before + r);
        if (valid && r == 0) wcount[wave][d] = before + (u32)__popcll(peers);
    }
    __syncthreads();
    { // tile-local exclusive offsets: digit-major, wave-minor (stable); one thread per digit + a 256-wide scan
        const u32 d = threadIdx.x;
        const u32 c0 = wcount[0][d], c1 = wcount[1][d], c2 = wcount[2][d], c3 = wcount[3][d];
        const u32 tot = c0 + c1 + c2 + c3;
        gbase[d] = tot;
        __syncthreads();
        for (int s = 1; s < 256; s <<= 1) {
            const u32 x = d >= (u32)s ? gbase[d - s] : 0;
            __syncthreads();
            gbase[d] += x;
            __syncthreads();
        }
        const u32 loc = gbase[d] - tot; // exclusive local offset of digit d
        __syncthreads();
        wcount[0][d] = loc;
        wcount[1][d] = loc + c0;
        wcount[2][d] = loc + c0 + c1;
        wcount[3][d] = loc + c0 + c1 + c2;
        gbase[d] = hist[(size_t)d * ntiles + blockIdx.x] + totals[d] - loc;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < SORT_ITEMS; ++j) {
        const size_t e = base + (size_t)j * 64 + lane;
        if (e < M) {
            const u32 lp = wcount[wave][(sort_key(k[j], lowmask, inv_from) >> shift) & 255u] + rk[j];
            sk[lp] = k[j];
            sv[lp] = v[j];
        }
    }
    __syncthreads();
    const u32 cnt = tile0 >= M ? 0u : (u32)((tile0 + SORT_TILE <= M) ? SORT_TILE : (M - tile0));
    for (u32 t = threadIdx.x; t < cnt; t += 256) { // consecutive lanes -> consecutive addresses within a digit run
        const u32 kk = sk[t];
        const u32 pos = gbase[(sort_key(kk, lowmask, inv_from) >> shift) & 255u] + t;
        keys_out[pos] = kk;
        vals_out[pos] = sv[t];
    }
}

size_t sort_pairs_temp_bytes(size_t n) {
    const size_t ntiles = (n + SORT_TILE - 1) / SORT_TILE;
    return 2 * n * 4 /* ping-pong pair */ + 256 * ntiles * 4 + 256 * 4 + 256;
}

bool sort_pairs_takes_device_count(int end_bit) { return end_bit >= 1 && end_bit <= 32; }

int sort_pairs(const u32 *keys_in, u32 *keys_out, const u32 *vals_in, u32 *vals_out, size_t n, int end_bit,
               void *tmp, size_t tmp_bytes, hipStream_t s, const u32 *d_count, u32 lowmask, u32 inv_from) {
    if (n == 0) return MG_OK;
    if (end_bit < 1 || end_bit > 32 || n >= (1ull << 32) || tmp_bytes < sort_pairs_temp_bytes(n)) return MG_ERR_ARG;
    const u32 M = (u32)n;
    const u32 ntiles = (u32)((n + SORT_TILE - 1) / SORT_TILE);
    const int passes = (end_bit + 7) / 8;
    // ping-pong so that the LAST pass lands in (keys_out, vals_out): with an even number of passes the first
    // one writes to the scratch pair
    u32 *tk = (u32 *)tmp, *tv = tk + n, *hist = tv + n, *totals = hist + (size_t)256 * ntiles;
    const u32 *ik = keys_in, *iv = vals_in;
    for (int p = 0; p < passes; ++p) {
        const bool to_out = ((passes - 1 - p) % 2) == 0;
        u32 *ok = to_out ? keys_out : tk, *ov = to_out ? vals_out : tv;
        hipLaunchKernelGGL(radix_hist, dim3(ntiles), dim3(256), 0, s, ik, M, 8 * p, ntiles, hist, d_count, lowmask, inv_from);
        hipLaunchKernelGGL(radix_scan_rows, dim3(256), dim3(256), 0, s, hist, ntiles, totals);
        hipLaunchKernelGGL(radix_scan_totals, dim3(1), dim3(256), 0, s, totals);
        hipLaunchKernelGGL(radix_scatter, dim3(ntiles), dim3(256), 0, s, ik, iv, ok, ov, M, 8 * p, ntiles, hist, totals, d_count, lowmask, inv_from);
        ik = ok;
        iv = ov;
    }
    MG_HIP(hipGetLastError());
    return MG_OK;
}

} // namespace mg
