// Proving-key wire format: arkworks 0.3 `ProvingKey::deserialize_unchecked` (uncompressed points, no
// curve/subgroup checks) as used by `ProvingContext::decode` (manta-crypto/src/arkworks/groth16.rs:268-288,
// writer :290-303; files produced by manta-pay/src/bin/generate_parameters.rs:162-217 and shipped as
// manta-parameters/data/pay/proving/*.lfs). SURVEY.md section 8(f-1), layout App. A.3 / App. C:
//   vk { alpha_g1, beta_g2, gamma_g2, delta_g2, gamma_abc_g1: Vec<G1> }, beta_g1, delta_g1,
//   a_query: Vec<G1>, b_g1_query: Vec<G1>, b_g2_query: Vec<G2>, h_query: Vec<G1>, l_query: Vec<G1>
//   point = x || y, little-endian canonical integers (Fq2: c0 then c1); infinity = flag bit 6 of the last
//   byte; Vec<T> = u64 LE length + elements.
// Parsed straight into the Montgomery limb arrays the C ABI takes, then handed to prover_create.
#include "host_ec.h"
#include "prover.h"
#include <cstring>
#include <vector>

namespace mg {
namespace {

struct Reader {
    const uint8_t *p;
    size_t left;
    bool ok = true;
    const uint8_t *take(size_t n) {
        if (!ok || left < n) {
            ok = false;
            return nullptr;
        }
        const uint8_t *q = p;
        p += n;
        left -= n;
        return q;
    }
    uint64_t u64le() {
        const uint8_t *q = take(8);
        if (!q) return 0;
        uint64_t v = 0;
        for (int i = 0; i < 8; ++i) v |= (uint64_t)q[i] << (8 * i);
        return v;
    }
};

template <class C> bool read_fp(const uint8_t *b, unsigned char top_mask, uint64_t *out_mont) {
    typedef host::HFp<C> HF;
    HF c = HF::zero();
    for (int i = 0; i < HF::BYTES; ++i) {
        uint8_t v = b[i];
        if (i == HF::BYTES - 1) v &= (uint8_t)~top_mask;
        c.v[i >> 3] |= (uint64_t)v << ((i & 7) * 8);
    }
    if (HF::geq_p(c.v)) return false; // Fp::from_repr fails on non-canonical input
    HF m = HF::to_mont(c);
    std::memcpy(out_mont, m.v, sizeof(m.v));
    return true;
}

// one uncompressed point with `deg` base-field coordinates per coordinate (1 = G1, 2 = G2)
template <class C> bool read_point(Reader &r, int deg, uint64_t *out) {
    typedef host::HFp<C> HF;
    const int nb = HF::BYTES, n64 = HF::N;
    const uint8_t *b = r.take((size_t)2 * deg * nb);
    if (!b) return false;
    const uint8_t flags = b[2 * deg * nb - 1];
    if (flags & 0x40) { // infinity
        std::memset(out, 0, (size_t)2 * deg * n64 * 8);
        return true;
    }
    for (int k = 0; k < 2 * deg; ++k) {
        const bool last = k == 2 * deg - 1;
        if (!read_fp<C>(b + (size_t)k * nb, last ? 0xC0 : 0, out + (size_t)k * n64)) return false;
    }
    return true;
}

template <class C> bool read_vec(Reader &r, int deg, std::vector<uint64_t> &out, uint64_t &count) {
    typedef host::HFp<C> HF;
    count = r.u64le();
    if (!r.ok) return false;
    const size_t per = (size_t)2 * deg * HF::N;
    if (count > r.left / ((size_t)2 * deg * HF::BYTES)) return false;
    out.resize(count * per);
    for (uint64_t i = 0; i < count; ++i)
        if (!read_point<C>(r, deg, out.data() + i * per)) return false;
    return true;
}

template <class Curve> int decode(int curve, const uint8_t *bytes, size_t len, Prover **out, const ProverOptions &o) {
    typedef typename Curve::Fq C;
    typedef host::HFp<C> HF;
    Reader r{bytes, len};
    const size_t g1 = 2 * HF::N, g2 = 4 * HF::N;
    std::vector<uint64_t> alpha(g1), beta2(g2), gamma2(g2), delta2(g2), beta1(g1), delta1(g1), gabc, a, b1, b2, h, l;
    uint64_t n_abc = 0, n_a = 0, n_b1 = 0, n_b2 = 0, n_h = 0, n_l = 0;
    bool ok = read_point<C>(r, 1, alpha.data()) && read_point<C>(r, 2, beta2.data()) &&
              read_point<C>(r, 2, gamma2.data()) && read_point<C>(r, 2, delta2.data()) &&
              read_vec<C>(r, 1, gabc, n_abc) && read_point<C>(r, 1, beta1.data()) &&
              read_point<C>(r, 1, delta1.data()) && read_vec<C>(r, 1, a, n_a) && read_vec<C>(r, 1, b1, n_b1) &&
              read_vec<C>(r, 2, b2, n_b2) && read_vec<C>(r, 1, h, n_h) && read_vec<C>(r, 1, l, n_l);
    if (!ok || r.left != 0) return MG_ERR_ARG;
    if (n_a != n_b1 || n_a != n_b2 || n_abc == 0 || n_abc >= n_a || n_l != n_a - n_abc || n_h == 0) return MG_ERR_ARG;
    mg_pk_view v;
    v.n_vars = n_a;
    v.n_inputs = n_abc;
    v.h_len = n_h;
    v.alpha_g1 = alpha.data();
    v.beta_g1 = beta1.data();
    v.delta_g1 = delta1.data();
    v.beta_g2 = beta2.data();
    v.delta_g2 = delta2.data();
    v.a_query = a.data();
    v.b_g1_query = b1.data();
    v.b_g2_query = b2.data();
    v.h_query = h.data();
    v.l_query = l.data();
    return prover_create_ex(curve, &v, o, out);
}

} // namespace

int prover_create_from_bytes_ex(int curve, const uint8_t *bytes, size_t len, const ProverOptions &o, Prover **out) {
    if (!bytes || !out) return MG_ERR_ARG;
    if (curve == 0) return decode<Bn254>(curve, bytes, len, out, o);
    if (curve == 1) return decode<Bls381>(curve, bytes, len, out, o);
    return MG_ERR_ARG;
}
int prover_create_from_bytes(int curve, const uint8_t *bytes, size_t len, Prover **out, const int *devices, int n_devices) {
    ProverOptions o;
    if (devices && n_devices > 0) o.devices = devices, o.n_devices = n_devices;
    return prover_create_from_bytes_ex(curve, bytes, len, o, out);
}

} // namespace mg
