// Element-wise prime-field arithmetic on the GPU through the kernels' own device functions -- the direct parity
// surface of SURVEY.md row a-10 (ark-ff 0.3 Fp256 / Fp384 Montgomery arithmetic, reached from
// manta-crypto/src/arkworks/groth16.rs:597): `mg_field_op` runs one operation over arrays of field elements in
// either of the two device representations,
//   repr 0: Fp<C>   saturated 32-bit limbs, R = 2^(32 N), always fully reduced (the ABI / arkworks memory format;
//                   NTT, SpMV, conversions)
//   repr 1: FpR<C>  reduced radix, lazily reduced (fpr_dev.h; the MSM kernels' internal format). Operands enter in
//                   the ABI format, are converted with from_std, then `lazy_a` / `lazy_b` times p is added limb-wise so
//                   that the operation sees a NON-canonical representative a + k p (what the kernels' intermediates
//                   look like), and the result is brought back with to_std (canonical).
// so that every function the Montgomery / lazy-reduction scheme consists of is compared with the CPU oracle (and with
// the reference-held BLS12-381 Fr Poseidon known answers) on edge values, not only through whole MSMs.
#include "engine.h"
#include "fpr_dev.h"
#include "params_gen.h"

namespace mg {

enum { FOP_ADD = 0, FOP_SUB = 1, FOP_MUL = 2, FOP_SQR = 3, FOP_NEG = 4, FOP_FROM_CANONICAL = 5, FOP_TO_CANONICAL = 6, FOP_INV = 7 };

template <class C> __device__ FpR<C> lazy_rep(const Fp<C> &s, int k) { // from_std(s) + k p, limbs renormalised
    typedef FpR<C> R;
    R r = R::from_std(s);
    if (k > 0) {
        R kp;
#pragma unroll
        for (int i = 0; i < R::K; ++i) {
            u32 v = 0;
#pragma unroll
            for (int m = 0; m <= C::RR_MAXM; ++m) v = (m == k) ? C::RR_MULT[m][i] : v;
            kp.v[i] = v;
        }
        r = R::add(r, kp);
    }
    return r;
}

template <class C>
__global__ __launch_bounds__(256) void field_op_kernel(int op, int repr, int lazy_a, int lazy_b, const u32 *__restrict__ a,
                                                       const u32 *__restrict__ b, u32 *__restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    typedef Fp<C> S;
    typedef FpR<C> R;
    const S x = S::load(a + i * S::N);
    S y = S::zero();
    if (b) y = S::load(b + i * S::N);
    S r = S::zero();
    if (repr == 0) {
        switch (op) {
        case FOP_ADD: r = S::add(x, y); break;
        case FOP_SUB: r = S::sub(x, y); break;
        case FOP_MUL: r = S::mul(x, y); break;
        case FOP_SQR: r = S::sqr(x); break;
        case FOP_NEG: r = S::neg(x); break;
        case FOP_FROM_CANONICAL: r = S::to_mont(x); break;
        case FOP_TO_CANONICAL: r = S::from_mont(x); break;
        case FOP_INV: r = S::inv(x); break;
        }
    } else {
        // bounds: the representatives are < (lazy + 1) p <= 4 p, so products see operand bounds <= 4 x 4 = 16 <= RR_LIM
        // for all four fields (70 .. 2520), sums < 8 p, and sub<4> / neg<4> add a multiple that covers the subtrahend
        const R xr = lazy_rep<C>(x, lazy_a), yr = lazy_rep<C>(y, lazy_b);
        R t = R::zero();
        switch (op) {
        case FOP_ADD: t = R::add(xr, yr); break;                 // < 8 p
        case FOP_SUB: t = R::template sub<4>(xr, yr); break;     // a + 4p - b < 8 p
        case FOP_MUL: t = R::mul(xr, yr); break;                 // < 2 p
        case FOP_SQR: t = R::sqr(xr); break;
        case FOP_NEG: t = R::template neg<4>(xr); break;         // 4p - a
        case FOP_INV: t = R::inv(R::template reduce<4>(xr)); break;
        default: t = xr; break;                                  // 5 / 6: the conversion round trip itself
        }
        // every result above is < 8 p: reduce<8> brings it below 2 p (exercising the quotient estimate), to_std is the
        // final almost-Montgomery product with one conditional subtraction
        r = R::template reduce<8>(t).to_std();
        if (op == FOP_FROM_CANONICAL) r = S::to_mont(r);
        if (op == FOP_TO_CANONICAL) r = S::from_mont(r);
    }
    r.store(out + i * S::N);
}

template <class C> static int run(int op, int repr, int lazy_a, int lazy_b, const u32 *a, const u32 *b, size_t n, u32 *out) {
    const size_t bytes = n * Fp<C>::N * 4;
    u32 *da = nullptr, *db = nullptr, *dout = nullptr;
    hipError_t e = hipMalloc((void **)&da, bytes);
    if (e == hipSuccess) e = hipMalloc((void **)&dout, bytes);
    if (e == hipSuccess && b) e = hipMalloc((void **)&db, bytes);
    if (e == hipSuccess) e = memcpy_sync(da, a, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess && b) e = memcpy_sync(db, b, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL((field_op_kernel<C>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, setup_stream(), op, repr, lazy_a, lazy_b, da, db,
                           dout, n);
        e = memcpy_sync(out, dout, bytes, hipMemcpyDeviceToHost);
    }
    hipFree(da);
    hipFree(db);
    hipFree(dout);
    if (e != hipSuccess) {
        set_last_hip_error(e, "mg_field_op", __FILE__, __LINE__);
        return e == hipErrorOutOfMemory ? MG_ERR_OOM : MG_ERR_HIP;
    }
    return MG_OK;
}

int field_op(int field, int op, int repr, int lazy_a, int lazy_b, const u32 *a, const u32 *b, size_t n, u32 *out) {
    if (!a || !out || n == 0 || op < 0 || op > 7 || repr < 0 || repr > 1) return MG_ERR_ARG;
    if (lazy_a < 0 || lazy_a > 3 || lazy_b < 0 || lazy_b > 3 || (repr == 0 && (lazy_a || lazy_b))) return MG_ERR_ARG;
    const bool binary = op == FOP_ADD || op == FOP_SUB || op == FOP_MUL;
    if (binary && !b) return MG_ERR_ARG;
    if (!binary) b = nullptr;
    switch (field) {
    case 0: return run<Bn254FrCfg>(op, repr, lazy_a, lazy_b, a, b, n, out);
    case 1: return run<Bn254FqCfg>(op, repr, lazy_a, lazy_b, a, b, n, out);
    case 2: return run<Bls381FrCfg>(op, repr, lazy_a, lazy_b, a, b, n, out);
    case 3: return run<Bls381FqCfg>(op, repr, lazy_a, lazy_b, a, b, n, out);
    }
    return MG_ERR_ARG;
}

} // namespace mg

// ---- clock probe (bench.py): what does the chip clock at under the accumulate kernel's kind of load? Every SIMD gets two
// wavefronts spinning on eight independent v_mad_u64_u32 chains each (the multiplier the MSM is bound by); each wavefront reads the
// shader-clock counter (s_memtime) and the constant-rate wall clock (s_memrealtime) before and after. ticks(s_memtime) per
// wall-clock second = the frequency s_memtime counts at during the load.
namespace mg {
__global__ __launch_bounds__(256) void clock_probe_kernel(u32 iters, u32 seed, unsigned long long *__restrict__ out) {
    unsigned long long acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = seed + threadIdx.x + 977u * k;
    const u32 m = 0x9e3779b9u ^ blockIdx.x;
    const long long c0 = clock64();
    const unsigned long long w0 = wall_clock64();
    for (u32 i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] = (unsigned long long)(u32)acc[k] * m + acc[k]; // v_mad_u64_u32, eight independent chains
    }
    const long long c1 = clock64();
    const unsigned long long w1 = wall_clock64();
    unsigned long long a = 0, b = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) a ^= acc[k];
    if ((threadIdx.x & 63) == 0) {
        const size_t w = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        out[3 * w] = (unsigned long long)(c1 - c0);
        out[3 * w + 1] = w1 - w0;
        out[3 * w + 2] = a ^ b; // keeps the chain alive
    }
}
// -> MHz of the s_memtime counter and of v_mad_u64_u32 issue (mads per SIMD-microsecond), duration of the probe in ms
int clock_probe(u32 iters, double *memtime_mhz, double *mad_issue_per_us_per_simd, double *ms) {
    int dev = 0, cus = 0, wall_khz = 0;
    MG_HIP(hipGetDevice(&dev));
    MG_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    MG_HIP(hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, dev));
    const u32 blocks = (u32)cus * 2; // 256 threads = 4 wavefronts per block, two blocks per CU: two wavefronts per SIMD
    const size_t waves = (size_t)blocks * 4;
    unsigned long long *d = nullptr;
    MG_HIP(hipMalloc((void **)&d, waves * 3 * sizeof(unsigned long long)));
    hipEvent_t e0, e1;
    MG_HIP(hipEventCreate(&e0));
    MG_HIP(hipEventCreate(&e1));
    hipLaunchKernelGGL(clock_probe_kernel, dim3(blocks), dim3(256), 0, setup_stream(), 64u, 1u, d); // warm-up
    MG_HIP(hipEventRecord(e0, setup_stream()));
    hipLaunchKernelGGL(clock_probe_kernel, dim3(blocks), dim3(256), 0, setup_stream(), iters, 2u, d);
    MG_HIP(hipEventRecord(e1, setup_stream()));
    hipError_t e = hipEventSynchronize(e1);
    std::vector<unsigned long long> h(waves * 3);
    if (e == hipSuccess) e = memcpy_sync(h.data(), d, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    float t = 0.f;
    hipEventElapsedTime(&t, e0, e1);
    hipFree(d);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    if (e != hipSuccess) {
        set_last_hip_error(e, "clock probe", __FILE__, __LINE__);
        return MG_ERR_HIP;
    }
    double ratio = 0;
    for (size_t w = 0; w < waves; ++w) ratio += (double)h[3 * w] / (double)(h[3 * w + 1] ? h[3 * w + 1] : 1);
    ratio /= (double)waves;
    if (memtime_mhz) *memtime_mhz = ratio * (double)wall_khz / 1e3;
    // 32 multiply-adds per iteration and lane-wavefront, two wavefronts per SIMD
    if (mad_issue_per_us_per_simd) *mad_issue_per_us_per_simd = t > 0 ? 2.0 * 32.0 * (double)iters / ((double)t * 1e3) : 0;
    if (ms) *ms = t;
    return MG_OK;
}
} // namespace mg
