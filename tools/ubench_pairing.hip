// dev tool: cost of the wave-cooperative pairing's building blocks (manta_rs_amd/csrc/pairing_coop.h) on one wavefront:
// Fq12 product / squaring / cyclotomic squaring, line multiplication, Frobenius, the inversion, G2Prepared::from (all its doubling
// and addition steps). BN254.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I manta_rs_amd/csrc -I include tools/ubench_pairing.hip -o tools/ubench_pairing
#include "pairing_coop.h"
#include <cstdio>
using namespace mg;
typedef Bn254Pairing K;
typedef PairingWave<K> PW;
typedef Pairing<K> P;

template <int MODE> __global__ __launch_bounds__(64) void bench(u32 *buf, int iters) {
    if (threadIdx.x < 6) {
        typename P::F2 v = P::F2::load(buf + threadIdx.x * P::F2W);
        PW::st(PW::R(0) + threadIdx.x, v);
        PW::st(PW::R(1) + threadIdx.x, v);
    }
    if (threadIdx.x < 3) P::F2::load(buf + threadIdx.x * P::F2W).store(PW::ring_slot(0) + threadIdx.x * P::F2W); // a line for `ell`
    PW::sync();
    const auto st = PW::sq_tab();
    const auto et = PW::ell_tab();
    const auto ct = PW::cyc_tab();
    const typename P::F px = P::F::load(buf), py = P::F::load(buf + P::N);
    const typename P::F2 qx = P::F2::load(buf), qy = P::F2::load(buf + P::F2W);
    typename P::F2 qacc = qy;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) PW::mul12(PW::R(0), PW::R(0), PW::R(1));
        if (MODE == 1) PW::ell(PW::R(0), 0, et);
        if (MODE == 2) PW::template frob12<1>(PW::R(0));
        if (MODE == 3) PW::inv12(PW::R(1), PW::R(0), PW::R(2), PW::R(3));
        if (MODE == 4) PW::sqr12(PW::R(0), PW::R(0), st);
        if (MODE == 5) PW::cyc_sqr12(PW::R(0), PW::R(0), ct);
        if (MODE == 6) PW::template prepare<false>(qx, qy, buf + 8192, px, py, px);
        if (MODE == 7) PW::conj12(PW::R(0));
        if (MODE == 8) qacc = PW::mul2(qacc, qx);
        if (MODE == 9) qacc = PW::mul2_xi(qacc);
        if (MODE == 10) qacc = P::F2::add(qacc, qx);
        if (MODE == 11) qacc.c0 = P::F::mul(qacc.c0, qx.c0);
        if (MODE == 13) qacc.c0 = PW::inv_euclid(P::F::add(qacc.c0, qx.c0));
        if (MODE == 12) { PW::st(PW::R(1), qacc); PW::sync(); qacc = PW::ld(PW::R(1) + 1); }
    }
    if (threadIdx.x < 6) PW::ld(PW::R(0) + threadIdx.x).store(buf + 2048 + threadIdx.x * P::F2W);
    if (MODE >= 8) qacc.store(buf + 4096 + threadIdx.x * P::F2W);
}
int main() {
    u32 *buf;
    hipMalloc(&buf, 1 << 18);
    std::vector<u32> h(1 << 14);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (u32)(i * 2654435761u) & 0x0fffffffu; // words below p's top word
    hipMemcpy(buf, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    const char *names[14] = {"mul12", "ell (line product)", "frob12<1>", "inv12 (lane 0)", "sqr12", "cyc_sqr12", "G2Prepared::from (all steps)", "conj12", "Fq2 product (registers)", "xi * Fq2", "Fq2 add", "Fq product", "LDS store + sync + load", "Fq inverse (binary Euclid, lane 0)"};
    auto run = [&](int mode, auto kern, int iters) {
        hipLaunchKernelGGL(kern, dim3(1), dim3(64), PW::miller_lds_bytes(), 0, buf, 1);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(1), dim3(64), PW::miller_lds_bytes(), 0, buf, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s %9.2f us each (%d iterations)\n", names[mode], ms * 1e3 / iters, iters);
    };
    run(0, bench<0>, 2000); run(1, bench<1>, 2000); run(2, bench<2>, 2000); run(3, bench<3>, 20);
    run(4, bench<4>, 2000); run(5, bench<5>, 2000); run(6, bench<6>, 20); run(7, bench<7>, 2000);
    run(8, bench<8>, 2000); run(9, bench<9>, 2000); run(10, bench<10>, 2000); run(11, bench<11>, 2000); run(12, bench<12>, 2000); run(13, bench<13>, 50);
    return 0;
}
