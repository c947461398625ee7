// Go / no-go probe: the m*p half of the 14 x 28-bit Montgomery product on the matrix pipe.
//
// In a*b every lane multiplies its own operands, but in the reduction half every lane multiplies its quotient m by the SAME
// constant p: across a wavefront that is a (64 elements x 56 digits) . (56 x 112 Toeplitz matrix of p's digits) integer
// matrix product. This probe does exactly that with v_mfma_i32_32x32x32_i8 on 7-bit digit slices and times it against the 196
// v_mad_u64_u32 it would replace, INCLUDING what the formulation costs around the matrix instruction:
//   split      m (14 limbs x 28 bits, one element per lane) -> 56 digits of 7 bits, packed four to a VGPR
//   exchange   the B operand of a 32x32x32 tile holds 16 digits of element (lane & 31) per lane, the upper lane half the second
//              16: v_permlane32_swap of the digit registers builds the operands of both 32-element blocks (8 swaps)
//   mfma       6 non-zero (row-tile, k-step) tiles per block of the Toeplitz matrix (constant A operands): 12 instructions
//   exchange   each lane ends up with half the 128 output rows of two elements: 64 v_permlane32_swap give it all rows of its own
//   recombine  112 column sums at 7-bit spacing -> the 28 column accumulators of the 28-bit representation (2 shift-adds and
//              2 multiply-adds per column)
// and checks the 28 recombined columns against the plain integer product. Not timed and not included: computing m itself
// (which in the column-wise Montgomery product depends on the running sum, digit by digit -- the matrix form needs the
// two-pass variant with a low-half product m = T_lo * p' first: another 105 multiply-adds).
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench_mfma_reduce.hip -o tools/ubench_mfma_reduce
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
typedef uint32_t u32;
typedef uint64_t u64;
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// BLS12-381 p in 14 limbs of 28 bits (low first)
__constant__ u32 PL[14];
static const u32 PL_H[14] = {0xfffaaab, 0xfefffff, 0x3ffffb9, 0xfffeb15, 0x6241eab, 0xa0f6b0f, 0xf6730d2, 0xf38512b, 0x4774b84, 0x4bacd76, 0xba7b643, 0xe69a4b1, 0x1ea397f, 0x1a011};

struct ATiles { // the six non-zero tiles of the Toeplitz matrix, as per-lane operand registers
    v4i t[6];
};

__device__ __forceinline__ void mad(u64 &acc, u32 a, u32 b) { asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b) : "vcc"); }

// ---- (A) what the MSM kernels do today: 14 x 14 multiply-adds into 28 column accumulators
template <bool CHECK> __device__ __forceinline__ void mp_valu(const u32 (&m)[14], u64 (&col)[28]) {
#pragma unroll
    for (int i = 0; i < 14; ++i)
#pragma unroll
        for (int j = 0; j < 14; ++j) mad(col[i + j], m[i], PL[j]);
}

// ---- (B) the matrix-pipe formulation
__device__ __forceinline__ void mp_mfma(const u32 (&m)[14], const v4i (&A)[6], u64 (&col)[28]) {
    // split: limb -> four 7-bit digits in the four bytes of a register
    u32 dg[16];
#pragma unroll
    for (int i = 0; i < 14; ++i) {
        const u32 x = m[i];
        u32 t = x & 0x7fu;
        t = ((x << 1) & 0x7f00u) | t;
        t = ((x << 2) & 0x7f0000u) | t;
        t = ((x << 3) & 0x7f000000u) | t;
        dg[i] = t;
    }
    dg[14] = dg[15] = 0;
    // operands of the two 32-element blocks for the two k-steps (digits 0..31, 32..63)
    v4i B0[2], B1[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        u32 lo[4], hi[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            lo[r] = dg[8 * s + r], hi[r] = dg[8 * s + 4 + r];
            // lanes 32..63 of `lo` <-> lanes 0..31 of `hi`
            const auto sw = __builtin_amdgcn_permlane32_swap(lo[r], hi[r], false, false);
            lo[r] = sw[0], hi[r] = sw[1];
        }
        B0[s] = v4i{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3]};
        B1[s] = v4i{(int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
    }
    // row tiles 0..3 (output columns 32 it .. 32 it + 31): tile list = (0,k0) (1,k0) (1,k1) (2,k0) (2,k1) (3,k1)
    v16i D0[4], D1[4];
    const v16i z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    D0[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[0], B0[0], z, 0, 0, 0);
    D1[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[0], B1[0], z, 0, 0, 0);
    D0[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[1], B0[0], z, 0, 0, 0);
    D1[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[1], B1[0], z, 0, 0, 0);
    D0[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[2], B0[1], D0[1], 0, 0, 0);
    D1[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[2], B1[1], D1[1], 0, 0, 0);
    D0[2] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[3], B0[0], z, 0, 0, 0);
    D1[2] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[3], B1[0], z, 0, 0, 0);
    D0[2] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[4], B0[1], D0[2], 0, 0, 0);
    D1[2] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[4], B1[1], D1[2], 0, 0, 0);
    D0[3] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[5], B0[1], z, 0, 0, 0);
    D1[3] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[5], B1[1], z, 0, 0, 0);
    // every lane gets all rows of ITS element: D0 reg <- rows of lane-half 0, D1 reg <- rows of lane-half 1
    u32 c[128]; // column sum of output position i: c[i]
#pragma unroll
    for (int it = 0; it < 4; ++it)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            u32 a = (u32)D0[it][j], b = (u32)D1[it][j];
            const auto sw = __builtin_amdgcn_permlane32_swap(a, b, false, false);
            a = sw[0], b = sw[1];
            // after the swap `a` holds the rows written by lane half 0 (row = (j&3) + 8 (j>>2)), `b` those of half 1 (+4)
            c[32 * it + (j & 3) + 8 * (j >> 2)] = a;
            c[32 * it + (j & 3) + 8 * (j >> 2) + 4] = b;
        }
    // recombine: 28-bit column J gets c[4J] + 2^7 c[4J+1] + 2^14 c[4J+2] + 2^21 c[4J+3]
#pragma unroll
    for (int J = 0; J < 28; ++J) {
        const u32 t = c[4 * J] + (c[4 * J + 1] << 7), u = c[4 * J + 2] + (c[4 * J + 3] << 7);
        mad(col[J], t, 1u);
        mad(col[J], u, 1u << 14);
    }
}

template <int MODE> __global__ __launch_bounds__(256) void k_bench(const ATiles *at, u64 *out, int iters, long long *cycles) {
    const int lane = threadIdx.x & 63;
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    u32 m[14];
    for (int i = 0; i < 14; ++i) m[i] = (t * 2654435761u + i * 40503u) & 0xfffffffu;
    v4i A[6];
    for (int q = 0; q < 6; ++q) A[q] = at[lane].t[q];
    u64 col[28];
    for (int i = 0; i < 28; ++i) col[i] = 0;
    const long long c0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) mp_valu<false>(m, col);
        else mp_mfma(m, A, col);
#pragma unroll
        for (int i = 0; i < 14; ++i) m[i] = (m[i] + (u32)col[i]) & 0xfffffffu; // next quotient depends on this product
    }
    const long long c1 = clock64();
    u64 x = 0;
    for (int i = 0; i < 28; ++i) x ^= col[i];
    out[t] = x;
    if (lane == 0) cycles[t >> 6] = c1 - c0;
}

// one product both ways, all 28 columns compared
__global__ __launch_bounds__(64) void k_check(const ATiles *at, int *bad) {
    const int lane = threadIdx.x & 63;
    u32 m[14];
    for (int i = 0; i < 14; ++i) m[i] = ((lane + 1) * 2654435761u + i * 977u * (lane + 3)) & 0xfffffffu;
    v4i A[6];
    for (int q = 0; q < 6; ++q) A[q] = at[lane].t[q];
    u64 a[28], b[28];
    for (int i = 0; i < 28; ++i) a[i] = b[i] = 0;
    mp_valu<true>(m, a);
    mp_mfma(m, A, b);
    // the two column sets split the same integer differently (a digit product straddling a limb boundary lands in the next
    // column in the 7-bit formulation): compare the VALUES, i.e. the columns after carry propagation
    u64 ca = 0, cb = 0;
    for (int i = 0; i < 28; ++i) {
        ca += a[i], cb += b[i];
        if ((ca & 0xfffffffull) != (cb & 0xfffffffull)) atomicAdd(bad, 1);
        ca >>= 28, cb >>= 28;
    }
    if (ca != cb) atomicAdd(bad, 1);
}

int main() {
    // digits of p
    int pd[112] = {0};
    {
        unsigned __int128 acc = 0;
        int bits = 0, k = 0;
        for (int i = 0; i < 14; ++i) {
            acc |= (unsigned __int128)(PL_H[i] & 0xfffffffu) << bits;
            bits += 28;
            while (bits >= 7) {
                pd[k++] = (int)(acc & 0x7f);
                acc >>= 7;
                bits -= 7;
            }
        }
    }
    u32 pl[14];
    for (int i = 0; i < 14; ++i) pl[i] = PL_H[i] & 0xfffffffu;
    hipMemcpyToSymbol(HIP_SYMBOL(PL), pl, sizeof(pl));
    // A operand of tile (it, ks): lane l holds row i = 32 it + (l & 31), k = 32 ks + 16 (l >> 5) + 4 r + byte; value pd[i - k]
    const int tiles[6][2] = {{0, 0}, {1, 0}, {1, 1}, {2, 0}, {2, 1}, {3, 1}};
    std::vector<ATiles> at(64);
    for (int l = 0; l < 64; ++l)
        for (int q = 0; q < 6; ++q)
            for (int r = 0; r < 4; ++r) {
                u32 w = 0;
                for (int by = 0; by < 4; ++by) {
                    const int i = 32 * tiles[q][0] + (l & 31), k = 32 * tiles[q][1] + 16 * (l >> 5) + 4 * r + by;
                    const int d = i - k;
                    const int v = (d >= 0 && d < 56 && k < 56) ? pd[d] : 0;
                    w |= (u32)(v & 0xff) << (8 * by);
                }
                at[l].t[q][r] = (int)w;
            }
    ATiles *d_at;
    hipMalloc(&d_at, sizeof(ATiles) * 64);
    hipMemcpy(d_at, at.data(), sizeof(ATiles) * 64, hipMemcpyHostToDevice);
    int *d_bad, bad = 0;
    hipMalloc(&d_bad, 4);
    hipMemset(d_bad, 0, 4);
    hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, 0, d_at, d_bad);
    hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost);
    printf("check: %d of 64 x 29 limbs of m * p differ from the integer product%s\n", bad, bad ? "  (LAYOUT MISMATCH: timings below still count the instructions)" : "");
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    for (int wps = 1; wps <= 2; ++wps) { // wavefronts per SIMD
        const int blocks = cus * wps, waves = blocks * 4, iters = 2000;
        u64 *out;
        long long *cyc;
        hipMalloc(&out, sizeof(u64) * blocks * 256);
        hipMalloc(&cyc, sizeof(long long) * waves);
        for (int mode = 0; mode < 2; ++mode) {
            std::vector<long long> h(waves);
            hipEvent_t e0, e1;
            hipEventCreate(&e0), hipEventCreate(&e1);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0, 0);
                if (mode == 0) hipLaunchKernelGGL(k_bench<0>, dim3(blocks), dim3(256), 0, 0, d_at, out, iters, cyc);
                else hipLaunchKernelGGL(k_bench<1>, dim3(blocks), dim3(256), 0, 0, d_at, out, iters, cyc);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
            }
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h.data(), cyc, sizeof(long long) * waves, hipMemcpyDeviceToHost);
            double avg = 0;
            for (auto v : h) avg += (double)v;
            avg /= waves;
            printf("%d wavefront(s) per SIMD, %-34s %8.0f s_memtime cycles per wavefront-product, %7.3f us per product per SIMD (wall)\n", wps,
                   mode == 0 ? "196 x v_mad_u64_u32 (today):" : "split + 12 MFMA i8 + recombine:", avg / iters, ms * 1e3 / iters / wps);
        }
        hipFree(out);
        hipFree(cyc);
    }
    return 0;
}
