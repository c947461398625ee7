// Round 4 probe: what would the FIRST batch-affine level of the 2^20 BLS12-381 G1 MSM cost in memory traffic alone?
// The level adds the 15.7 M table entries of the sorted (bucket, entry) stream pairwise in affine form: 7.86 M additions, each
//   pass 1  gathers its TWO summands from the 2 GB window table (128 B records at random addresses), forms the denominator x2 - x1
//           and the lane's running product of denominators, and stashes both points contiguously (2 x 104 B) next to the 48 B prefix
//           product, so that the second pass need not gather again;
//   (one inversion per ~16 k additions through a product tree over the per-lane totals: negligible traffic, not modelled)
//   pass 2  reads the stash (208 B) and the prefix product (48 B), finishes lambda, x3, y3 and writes the affine sum (104 B).
// This probe runs exactly those loads and stores with NO field arithmetic (an XOR keeps the data flow alive): the time is a FLOOR for
// the level. The XYZZ accumulate kernel does ALL 15.7 M mixed additions of the MSM in 2.49 ms (profiles/r04_limbs_13x30_ab.txt).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint32_t u32;
typedef uint64_t u64;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__device__ __forceinline__ u32 mix(u32 x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
// pass 1: one addition per lane-iteration; L additions per lane (the lane's running product would chain through them)
__global__ __launch_bounds__(256) void pass1(const uint4 *__restrict__ table, u32 n_entries, uint4 *__restrict__ stash, uint4 *__restrict__ prefix,
                                             u32 n_adds, u32 L) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    uint4 run = make_uint4(t, 1, 2, 3);
    for (u32 j = 0; j < L; ++j) {
        const u32 a = t * L + j;
        if (a >= n_adds) return;
        const u32 i0 = mix(2 * a) % n_entries, i1 = mix(2 * a + 1) % n_entries; // (the sorted stream addresses the table at random)
        const uint4 *p = table + (size_t)i0 * 8, *q = table + (size_t)i1 * 8;   // 128 B records
        uint4 v[13];
#pragma unroll
        for (int k = 0; k < 7; ++k) v[k] = p[k];       // x | y of the first summand: 104 B of the record
#pragma unroll
        for (int k = 0; k < 6; ++k) v[7 + k] = q[k];
        const uint4 q6 = q[6];
#pragma unroll
        for (int k = 0; k < 13; ++k) stash[(size_t)a * 13 + k] = v[k]; // 208 B: both points
        run.x ^= v[0].x ^ v[7].x ^ q6.x; run.y ^= v[3].y ^ v[9].y; run.z += v[5].z; run.w ^= v[12].w;
        prefix[(size_t)a * 3 + 0] = run, prefix[(size_t)a * 3 + 1] = v[1], prefix[(size_t)a * 3 + 2] = v[8]; // 48 B running product
    }
}
__global__ __launch_bounds__(256) void pass2(const uint4 *__restrict__ stash, const uint4 *__restrict__ prefix, uint4 *__restrict__ out, u32 n_adds) {
    const u32 a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= n_adds) return;
    uint4 v[13], pr[3];
#pragma unroll
    for (int k = 0; k < 13; ++k) v[k] = stash[(size_t)a * 13 + k];
#pragma unroll
    for (int k = 0; k < 3; ++k) pr[k] = prefix[(size_t)a * 3 + k];
#pragma unroll
    for (int k = 0; k < 7; ++k) { // 104 B result in a 128 B record
        uint4 r = v[k];
        r.x ^= v[(k + 6) % 13].x ^ pr[k % 3].x, r.y ^= pr[(k + 1) % 3].y;
        out[(size_t)a * 8 + k] = r;
    }
}
__global__ void fill(uint4 *p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint4((u32)i, (u32)(i >> 7), 3, 4);
}
int main() {
    const u32 n_entries = 15u << 20, n_adds = n_entries / 2;
    uint4 *table, *stash, *prefix, *out;
    CK(hipMalloc(&table, (size_t)n_entries * 128));
    CK(hipMalloc(&stash, (size_t)n_adds * 208));
    CK(hipMalloc(&prefix, (size_t)n_adds * 48));
    CK(hipMalloc(&out, (size_t)n_adds * 128));
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, table, (size_t)n_entries * 8);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1, e2;
    hipEventCreate(&e0), hipEventCreate(&e1), hipEventCreate(&e2);
    for (u32 L : {1u, 4u, 8u, 16u}) {
        const u32 lanes = (n_adds + L - 1) / L;
        float best1 = 1e9f, best2 = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(pass1, dim3((lanes + 255) / 256), dim3(256), 0, 0, table, n_entries, stash, prefix, n_adds, L);
            hipEventRecord(e1);
            hipLaunchKernelGGL(pass2, dim3((n_adds + 255) / 256), dim3(256), 0, 0, stash, prefix, out, n_adds);
            hipEventRecord(e2);
            CK(hipEventSynchronize(e2));
            float a, b;
            hipEventElapsedTime(&a, e0, e1), hipEventElapsedTime(&b, e1, e2);
            best1 = a < best1 ? a : best1, best2 = b < best2 ? b : best2;
        }
        const double by1 = (double)n_adds * (256 + 208 + 48), by2 = (double)n_adds * (208 + 48 + 104);
        printf("additions per lane %2u: pass 1 %.3f ms (%.2f TB/s over %d B per addition), pass 2 %.3f ms (%.2f TB/s over %d B), level floor %.3f ms for %.2f M additions = %.3f ns per addition\n",
               L, best1, by1 / best1 / 1e9, 256 + 208 + 48, best2, by2 / best2 / 1e9, 208 + 48 + 104, best1 + best2, n_adds / 1e6, (best1 + best2) * 1e6 / n_adds);
    }
    return 0;
}
