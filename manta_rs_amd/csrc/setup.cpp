// Groth16 key generation on the MI355X: the part of `Groth16::compile` (manta-crypto/src/arkworks/groth16.rs:571-586
// -> ark-groth16 0.3 generate_parameters) that costs time -- 3V + D + const fixed-base multiplications in G1 and
// V in G2 -- runs on the GPU; the O(D + nnz) scalar preparation (Lagrange coefficients at tau, QAP evaluation)
// stays on the host. Toxic waste and the two group generators are inputs: the reference draws them from its RNG
// (alpha, beta, gamma, delta, g1, g2, then tau) and the shim passes them on, so that a seeded RNG gives the
// same key. Key conventions as readable in-repo at manta-trusted-setup/src/groth16/mpc.rs:251-431.
// SURVEY.md section 8(f-3).
#include "prover.h"
#include <cstring>

namespace mg {

static int mul_all(GroupEngine *e, const u64 *gen, const std::vector<u64> &scalars, std::vector<u32> &out) {
    const size_t n = scalars.size() / 4, aw = (size_t)e->affine_words();
    u32 *d_s = nullptr, *d_o = nullptr;
    hipError_t er = hipMalloc((void **)&d_s, n * 32);
    if (er == hipSuccess) er = hipMalloc((void **)&d_o, n * aw * 4);
    if (er == hipSuccess) er = memcpy_sync(d_s, scalars.data(), n * 32, hipMemcpyHostToDevice);
    int rc = MG_OK;
    if (er == hipSuccess) rc = e->fixed_base_mul((const u32 *)gen, d_s, n, d_o, nullptr);
    out.resize(n * aw);
    if (er == hipSuccess && !rc) er = memcpy_sync(out.data(), d_o, n * aw * 4, hipMemcpyDeviceToHost);
    if (d_s) hipFree(d_s);
    if (d_o) hipFree(d_o);
    if (er != hipSuccess) {
        set_last_hip_error(er, "groth16_setup", __FILE__, __LINE__);
        return er == hipErrorOutOfMemory ? MG_ERR_OOM : MG_ERR_HIP;
    }
    return rc;
}

int groth16_setup(int curve, const mg_csr *a, const mg_csr *b, const mg_csr *c, u64 m, u64 V, u64 P, const u64 *toxic5,
                  const u64 *g1_gen, const u64 *g2_gen, const mg_pk_out *out) {
    FrEngine *fr = get_ntt_engine(curve);
    GroupEngine *g1 = get_engine(curve, 1), *g2 = get_engine(curve, 2);
    if (!fr || !g1 || !g2 || !a || !b || !c || !toxic5 || !g1_gen || !g2_gen || !out || m == 0) return MG_ERR_ARG;
    if (!out->alpha_g1 || !out->beta_g1 || !out->delta_g1 || !out->beta_g2 || !out->gamma_g2 || !out->delta_g2 ||
        !out->gamma_abc_g1 || !out->a_query || !out->b_g1_query || !out->b_g2_query || !out->h_query || !out->l_query)
        return MG_ERR_ARG;
    unsigned lg = 0;
    while (((u64)1 << lg) < m + P) ++lg; // GeneralEvaluationDomain::new(m + P)
    if ((int)lg > fr->two_adicity()) return MG_ERR_DOMAIN;
    const size_t D = (size_t)1 << lg;
    std::vector<u64> s1, s2;
    int rc = fr->setup_scalars(a, b, c, m, V, P, lg, toxic5, s1, s2);
    if (rc) return rc;
    std::vector<u32> p1, p2;
    if ((rc = mul_all(g1, g1_gen, s1, p1)) || (rc = mul_all(g2, g2_gen, s2, p2))) return rc;
    const size_t w1 = (size_t)g1->affine_words() * 4, w2 = (size_t)g2->affine_words() * 4; // bytes per point
    const unsigned char *q1 = (const unsigned char *)p1.data(), *q2 = (const unsigned char *)p2.data();
    std::memcpy(out->alpha_g1, q1, w1);
    std::memcpy(out->beta_g1, q1 + w1, w1);
    std::memcpy(out->delta_g1, q1 + 2 * w1, w1);
    size_t o = 3;
    std::memcpy(out->gamma_abc_g1, q1 + o * w1, P * w1), o += P;
    std::memcpy(out->a_query, q1 + o * w1, V * w1), o += V;
    std::memcpy(out->b_g1_query, q1 + o * w1, V * w1), o += V;
    std::memcpy(out->h_query, q1 + o * w1, (D - 1) * w1), o += D - 1;
    std::memcpy(out->l_query, q1 + o * w1, (V - P) * w1);
    std::memcpy(out->beta_g2, q2, w2);
    std::memcpy(out->gamma_g2, q2 + w2, w2);
    std::memcpy(out->delta_g2, q2 + 2 * w2, w2);
    std::memcpy(out->b_g2_query, q2 + 3 * w2, V * w2);
    return MG_OK;
}

} // namespace mg
