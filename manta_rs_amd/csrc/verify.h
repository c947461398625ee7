// Groth16 verification on the MI355X: pairing engine interface + the verifying-key object.
#pragma once
#include "../../include/mantagpu.h"
#include "engine.h"

namespace mg {

class PairingEngine {
  public:
    virtual ~PairingEngine() {}
    virtual int n_coeffs() const = 0;    // line-coefficient triples per prepared G2 point (91 BN254 / 68 BLS12-381)
    virtual int coeff_words() const = 0; // u32 per triple
    virtual int f12_words() const = 0;
    virtual int fq_words() const = 0;
    // G2Prepared::from for n affine points (host, Montgomery) -> device array n x n_coeffs x coeff_words (hipFree it)
    virtual int prepare(const u32 *q_affine_host, size_t n, u32 **d_out) = 0;
    // out = [final_exponentiation] ( prod_i MillerLoop(P_i, Q_i) ): P host affine G1; Q_i either prepared (d_coeffs[i] a
    // device pointer from prepare()) or, where d_coeffs[i] is null, the affine G2 point q_affine_host[i] (prepared on the
    // fly next to its Miller loop; q_affine_host may be null when every pair is prepared);
    // skip[i] != 0 leaves pair i out (its G2 point was infinity)
    virtual int pairing_product(const u32 *p_affine_host, const u32 *const *d_coeffs, const u32 *q_affine_host,
                                const unsigned char *skip, size_t n, bool do_final_exp, u32 *out_f12_host) = 0;
    // the same with the G1 points of the first n_xyzz pairs in DEVICE memory as (X, Y, ZZ, ZZZ), 4 fq_words() each -- left there
    // by a kernel on the HIP stream `producer` (the lines of such a pair are taken times ZZ ZZZ, which the final
    // exponentiation removes: no inversion between a scalar multiplication and its pairing)
    virtual int pairing_product_xyzz(const u32 *p_affine_host, const u32 *d_p_xyzz, size_t n_xyzz, void *producer,
                                     const u32 *const *d_coeffs, const u32 *q_affine_host, const unsigned char *skip, size_t n,
                                     bool do_final_exp, u32 *out_f12_host) = 0;
    // the same in two steps: the Miller loops of the first n_early pairs start at once (P points of those only; coefficients /
    // Q / skip of all n); end() takes the G1 points of the other pairs -- which must have prepared coefficients -- and finishes.
    // A handle is consumed by exactly one end() or abandon().
    virtual int pairing_product_begin(const u32 *p_affine_host, const u32 *const *d_coeffs, const u32 *q_affine_host,
                                      const unsigned char *skip, size_t n, size_t n_early, void **handle) = 0;
    virtual int pairing_product_end(void *handle, const u32 *p_late_affine_host, bool do_final_exp, u32 *out_f12_host) = 0;
    virtual void pairing_product_abandon(void *handle) = 0;
    // *ok = (prod_i e(P_i, Q_i) == 1): all points host affine Montgomery words, every Q_i prepared on the fly; a pair with an
    // infinity member (all-zero words) contributes 1
    virtual int product_is_one(const u32 *p_affine_host, const u32 *q_affine_host, size_t n, int *ok) = 0;
};
PairingEngine *make_pairing_engine_bn254();
PairingEngine *make_pairing_engine_bls381();
PairingEngine *get_pairing_engine(int curve); // per (device, curve)

class Verifier {
  public:
    virtual ~Verifier() {}
    virtual u64 n_inputs() const = 0;
    // ark-groth16 verify_proof: inputs = P - 1 public inputs (Montgomery Fr), proof = a | b | c affine Montgomery
    virtual int verify(const u64 *inputs_mont, const u64 *proof_points, int *ok) = 0;
    // k proofs at once by random linear combination; rand = k x 2 u64 (128-bit coefficients from the caller's RNG)
    virtual int verify_batch(u64 k, const u64 *inputs_mont, const u64 *proof_points, const u64 *rand128, int *ok) = 0;
    virtual size_t encoded_size() const = 0;
    virtual int encode(uint8_t *out) const = 0;          // VerifyingContext wire format, groth16.rs:337-361
    virtual int alpha_beta_bytes(uint8_t *out) const = 0; // 12 canonical Fq elements, arkworks order
};
int verifier_create(int curve, const u64 *alpha_g1, const u64 *beta_g2, const u64 *gamma_g2, const u64 *delta_g2,
                    const u64 *gamma_abc_g1, u64 n_inputs, Verifier **out);
int verifier_create_from_bytes(int curve, const uint8_t *bytes, size_t len, Verifier **out);
// Proof::deserialize (compressed a | b | c) -> affine Montgomery points; checks canonical encoding, curve and subgroup
int proof_decode(int curve, const uint8_t *bytes, u64 *points_out);

} // namespace mg
