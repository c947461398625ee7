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
    if (e == hipSuccess) e = hipMemcpy(da, a, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess && b) e = hipMemcpy(db, b, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL((field_op_kernel<C>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, op, repr, lazy_a, lazy_b, da, db,
                           dout, n);
        e = hipMemcpy(out, dout, bytes, hipMemcpyDeviceToHost);
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
