"""GPU verification (SURVEY f-2): `Groth16::verify`, `VerifyingContext` and its codec on the MI355X.

The strongest check needs no oracle at all: the reference's committed verifying-key files hold, next to the key, the
RESULTS of arkworks' pairing code on it -- e(alpha_g1, beta_g2) (384 B) and the 2 x 91 G2Prepared line-coefficient triples
of -gamma_g2 and -delta_g2 (34 944 B). A context built from the five key components alone recomputes all of that on the
GPU, and its encoding must equal the reference's file, byte for byte, for all six committed keys."""
import numpy as np
import pytest

import helpers as H
import oracle_lib as O
from manta_rs_amd import keygen, synth
from vk_fixtures import VK, VK_FILES

pytestmark = pytest.mark.gpu


class _Key:
    pass


@pytest.mark.parametrize("name", sorted(VK_FILES))
def test_gpu_reproduces_the_reference_verifying_key_files(gpu, name):
    vk = VK(name)
    k = _Key()
    k.alpha_g1, k.beta_g2, k.gamma_g2, k.delta_g2, k.gamma_abc_g1 = vk.alpha, vk.beta, vk.gamma, vk.delta, np.stack(vk.abc)
    ctx = gpu.VerifyingContext(0, k)                      # VerifyingContext::new(&vk): everything derived on the GPU
    assert ctx.num_inputs == vk.P
    assert ctx.alpha_g1_beta_g2() == vk.alpha_beta_bytes   # Miller loop + arkworks' BN final exponentiation
    enc = ctx.encode()
    assert len(enc) == len(vk.raw)
    assert enc == vk.raw                                   # incl. both G2Prepared blocks (91 line coefficients each)
    # Decode for VerifyingContext (groth16.rs:498-517) and re-encode: the codec round trip on the reference's bytes
    dec = gpu.VerifyingContext.decode(0, vk.raw)
    assert dec.encode() == vk.raw and dec.alpha_g1_beta_g2() == vk.alpha_beta_bytes
    for bad in (vk.raw[:-1], vk.raw + b"\0", vk.raw[:40] + bytes([vk.raw[40] ^ 1]) + vk.raw[41:]):
        with pytest.raises(gpu.MantaGpuError):
            gpu.VerifyingContext.decode(0, bad)


@pytest.mark.parametrize("curve", [0, 1])
def test_gpu_verify_accepts_and_rejects_like_the_reference(gpu, curve):
    """`Groth16::verify` on GPU proofs: accepted with the right public inputs, rejected with fuzzed inputs / fuzzed proof
    (manta-pay/src/test/transfer.rs:346-417 fuzzes exactly these); r = 0 proofs, a key over non-standard generators, and
    agreement with the oracle's pairing check on every case."""
    c = synth.make_circuit(curve, 600, 420, 7, seed=900 + curve)
    r = synth.FR_MODULUS[curve]
    g1 = O.g_mul(curve, 1, keygen.generator(curve, 1), synth.ints_to_limbs([0xabcdef % r], 4)[0])
    g2 = O.g_mul(curve, 2, keygen.generator(curve, 2), synth.ints_to_limbs([0x123457 % r], 4)[0])
    pk = keygen.generate(c, synth.from_mont(H.toxic(curve, seed=31), r), g1, g2)
    ctx = gpu.ProvingContext(curve, pk)
    ctx.set_r1cs(gpu.R1CS.from_circuit(c))
    vctx = gpu.VerifyingContext(curve, pk)
    rs = H.rand_fr_mont(curve, 4, seed=96)
    inputs = c.z[1:c.P]
    for r_, s_ in ((rs[0], rs[1]), (np.zeros(4, np.uint64), rs[2])):
        proof = gpu.Groth16.prove_with_randomness(ctx, c.z, r_, s_)
        assert O.groth16_verify(curve, pk, inputs, proof) == 1
        assert gpu.groth16_verify(vctx, inputs, proof) is True
        for j in (0, c.P - 2):                                        # any fuzzed public input invalidates it
            bad = inputs.copy()
            bad[j] = rs[3]
            assert gpu.groth16_verify(vctx, bad, proof) is False and O.groth16_verify(curve, pk, bad, proof) == 0
        pts = gpu.proof_decode(curve, proof)
        w1 = gpu.affine_limbs(curve, 1)
        for sl in (slice(0, w1), slice(w1, 3 * w1), slice(3 * w1, 4 * w1)):   # replace a, b or c by another valid point
            fuzz = pts.copy()
            g = 2 if sl.stop - sl.start == 2 * w1 else 1
            fuzz[sl] = O.g_mul(curve, g, pts[sl], synth.ints_to_limbs([3], 4)[0])
            assert gpu.groth16_verify(vctx, inputs, fuzz) is False
    # proofs of another assignment verify against ITS inputs only
    c2 = synth.reassign(c, seed=77)
    p2 = gpu.Groth16.prove_with_randomness(ctx, c2.z, rs[0], rs[1])
    assert gpu.groth16_verify(vctx, c2.z[1:c.P], p2) and not gpu.groth16_verify(vctx, inputs, p2)
    with pytest.raises(ValueError):
        gpu.groth16_verify(vctx, inputs[:-1], p2)
    # the key's wire format round-trips through the GPU-built context and decodes to an equally good verifier
    again = gpu.VerifyingContext.decode(curve, vctx.encode())
    assert again.encode() == vctx.encode() and gpu.groth16_verify(again, c2.z[1:c.P], p2)
    # non-canonical / off-curve proof bytes are refused at decode, like Proof::deserialize
    bad = bytearray(p2)
    bad[-1] |= 0x3f if curve == 0 else 0x1f                          # x of c >= q
    with pytest.raises(gpu.MantaGpuError):
        gpu.proof_decode(curve, bytes(bad))


@pytest.mark.parametrize("curve", [0, 1])
def test_gpu_pairing_is_bilinear_and_matches_the_oracle_up_to_the_fixed_exponent(gpu, curve):
    """e(a G1, b G2) computed on the GPU (vk.alpha_g1_beta_g2): equal for (a, b) and (ab, 1), different otherwise, and
    equal to the oracle's textbook pairing raised to the fixed multiple arkworks' final exponentiation computes
    (BN254: 2x(6x^2+3x+1), pinned by the reference's key files; BLS12-381: 3, Hayashida-Hayasaka-Teruya)."""
    r = synth.FR_MODULUS[curve]
    G1, G2 = O.generator(curve, 1), O.generator(curve, 2)
    lim = lambda k: synth.ints_to_limbs([k % r], 4)[0]

    def ab_bytes(a, b):
        k = _Key()
        k.alpha_g1, k.beta_g2 = O.g_mul(curve, 1, G1, lim(a)), O.g_mul(curve, 2, G2, lim(b))
        k.gamma_g2, k.delta_g2, k.gamma_abc_g1 = G2, G2, np.stack([G1])
        return gpu.VerifyingContext(curve, k).alpha_g1_beta_g2(), k

    a, b = 0x1234567890abcdef, 0xfedcba9876543211
    e1, k1 = ab_bytes(a, b)
    e2, _ = ab_bytes(a * b, 1)
    e3, _ = ab_bytes(a + 1, b)
    assert e1 == e2 and e1 != e3
    if curve == 0:
        assert e1 == O.pairing_bytes(0, k1.alpha_g1, k1.beta_g2, ark_exp=True)
    else:
        # the oracle's textbook loop runs over |x| without arkworks' final conjugation for the negative BLS12-381
        # parameter, i.e. it computes the inverse pairing: compare with e(-P, Q)^3
        neg_alpha = O.g_mul(1, 1, k1.alpha_g1, lim(r - 1))
        assert e1 == O.pairing_bytes_pow(1, neg_alpha, k1.beta_g2, 3)
    one = ab_bytes(0, b)[0]  # e(infinity, Q) = 1
    nb = len(one) // 12
    assert one == (1).to_bytes(nb, "little") + bytes(11 * nb)


def test_gpu_verify_from_several_threads_at_once(gpu):
    """Verifications from four host threads on one context: each call takes a pooled pairing workspace (two streams, the
    third pair's point in the kernel arguments) and an MSM workspace of the shared engine; valid proofs stay valid, a
    foreign proof and fuzzed inputs stay rejected, whatever is in flight next to them."""
    from concurrent.futures import ThreadPoolExecutor
    curve = 0
    c = synth.make_circuit(curve, 700, 500, 9, seed=977)
    pk = keygen.generate(c, synth.from_mont(H.toxic(curve, seed=33), synth.FR_MODULUS[curve]))
    ctx = gpu.ProvingContext(curve, pk)
    ctx.set_r1cs(gpu.R1CS.from_circuit(c))
    vctx = gpu.VerifyingContext(curve, pk)
    R = synth.Reassigner(c)
    cs = [R.assign(7000 + q) for q in range(6)]
    rs = H.rand_fr_mont(curve, 12, seed=98)
    proofs = [gpu.Groth16.prove_with_randomness(ctx, x.z, rs[2 * q], rs[2 * q + 1]) for q, x in enumerate(cs)]

    def job(t):
        good = bad = 0
        for it in range(12):
            q = (t + it) % len(cs)
            good += gpu.groth16_verify(vctx, cs[q].z[1:c.P], proofs[q]) is True
            wrong = proofs[(q + 1) % len(cs)] if it % 2 else proofs[q]
            inputs = cs[q].z[1:c.P].copy()
            if not it % 2:
                inputs[it % (c.P - 1)] = rs[0]
            bad += gpu.groth16_verify(vctx, inputs, wrong) is False
        return good, bad

    with ThreadPoolExecutor(max_workers=4) as ex:
        res = list(ex.map(job, range(4)))
    assert res == [(12, 12)] * 4


@pytest.mark.parametrize("curve", [0, 1])
def test_gpu_pairing_at_the_edges_of_the_scalar_range(gpu, curve):
    """The pairing's one-product-per-lane levels (table-driven signed sums, lazily reduced small-coefficient combinations,
    cyclotomic squarings, the width-4 NAF of x, the almost-inverse) on the points with the smallest and largest multiples:
    e(-G1, -G2) = e(G1, G2) = e(2 G1, (r+1)/2 G2), e(-G1, G2) = e(G1, -G2) = e(G1, G2)^-1 (the conjugate), and BN254's values
    against the oracle's arkworks-exponent pairing."""
    r = synth.FR_MODULUS[curve]
    G1, G2 = O.generator(curve, 1), O.generator(curve, 2)
    lim = lambda k: synth.ints_to_limbs([k % r], 4)[0]

    def e(a, b):
        k = _Key()
        k.alpha_g1, k.beta_g2 = O.g_mul(curve, 1, G1, lim(a)), O.g_mul(curve, 2, G2, lim(b))
        k.gamma_g2, k.delta_g2, k.gamma_abc_g1 = G2, G2, np.stack([G1])
        return gpu.VerifyingContext(curve, k).alpha_g1_beta_g2(), k

    base, kb = e(1, 1)
    assert e(r - 1, r - 1)[0] == base and e(2, (r + 1) // 2)[0] == base and e((r + 1) // 2, 2)[0] == base
    inv1, inv2 = e(r - 1, 1)[0], e(1, r - 1)[0]
    assert inv1 == inv2 != base
    nb = len(base) // 12
    words = lambda bs: [int.from_bytes(bs[i * nb:(i + 1) * nb], "little") for i in range(12)]
    q = synth.FQ_MODULUS[curve] if hasattr(synth, "FQ_MODULUS") else None
    if q is not None:  # the inverse of a unitary element is its conjugate: the odd powers of w change sign
        wb, wi = words(base), words(inv1)
        # arkworks' memory order c0.(c0, c1, c2), c1.(c0, c1, c2), an Fq2 each: the second half is c1 (the odd powers of w)
        assert wi[:6] == wb[:6] and all((x + y) % q == 0 for x, y in zip(wi[6:], wb[6:]))
    if curve == 0:
        assert base == O.pairing_bytes(0, kb.alpha_g1, kb.beta_g2, ark_exp=True)
        k2 = e(r - 1, 1)[1]
        assert inv1 == O.pairing_bytes(0, k2.alpha_g1, k2.beta_g2, ark_exp=True)


@pytest.mark.parametrize("curve,k", [(0, 1), (0, 8), (0, 64), (1, 5)])
def test_gpu_batch_verification(gpu, curve, k):
    """mg_groth16_verify_batch: k proofs with distinct assignments / inputs in one pass (k + 3 Miller loops, one final
    exponentiation). All valid -> accepted; any single invalid member (wrong inputs, foreign proof) -> rejected."""
    c0 = synth.make_circuit(curve, 500, 380, 6, seed=950 + curve)
    pk = keygen.generate(c0, synth.from_mont(H.toxic(curve, seed=32), synth.FR_MODULUS[curve]))
    ctx = gpu.ProvingContext(curve, pk)
    ctx.set_r1cs(gpu.R1CS.from_circuit(c0))
    vctx = gpu.VerifyingContext(curve, pk)
    R = synth.Reassigner(c0)
    cs = [R.assign(5000 + q) for q in range(k)]
    rs = H.rand_fr_mont(curve, 2 * k, seed=97)
    proofs = gpu.Groth16.prove_batch(ctx, np.stack([x.z for x in cs]), rs[:k], rs[k:])
    inputs = np.stack([x.z[1:c0.P] for x in cs])
    rnd = np.random.RandomState(5).randint(1, 1 << 62, size=(k, 2)).astype(np.uint64)
    assert gpu.groth16_verify_batch(vctx, inputs, proofs, rnd) is True
    assert all(gpu.groth16_verify(vctx, inputs[q], proofs[q]) for q in range(min(k, 3)))
    for victim in sorted({0, k // 2, k - 1}):
        bad = inputs.copy()
        bad[victim, 1] = rs[0]
        assert gpu.groth16_verify_batch(vctx, bad, proofs, rnd) is False, victim
    if k > 1:
        swapped = list(proofs)
        swapped[0], swapped[1] = swapped[1], swapped[0]
        assert gpu.groth16_verify_batch(vctx, inputs, swapped, rnd) is False
    zero = rnd.copy()
    zero[k - 1] = 0
    with pytest.raises(gpu.MantaGpuError):
        gpu.groth16_verify_batch(vctx, inputs, proofs, zero)


@pytest.mark.parametrize("curve", [0, 1])
def test_valid_pairing_ratio_like_the_reference(gpu, curve):
    """manta-crypto/src/arkworks/pairing.rs:295-330 `{bls12_381,bn254}_has_valid_pairing_ratio`: for random g1, g2 and
    scalar, E::same((g1, g2 * scalar), (g1 * scalar, g2)) holds -- on the GPU (`mg_pairing_check`), for several draws,
    and fails as soon as one of the four points is off."""
    from manta_rs_amd import ceremony
    r = synth.FR_MODULUS[curve]
    rng = synth.XorShift(0x5EED0 + curve)
    G1, G2 = O.generator(curve, 1), O.generator(curve, 2)
    lim = lambda k: synth.ints_to_limbs([k % r], 4)[0]
    for _ in range(3):
        g1, g2 = O.g_mul(curve, 1, G1, lim(rng.field(r))), O.g_mul(curve, 2, G2, lim(rng.field(r)))  # random group elements
        k = rng.field(r)
        g1k, g2k = O.g_mul(curve, 1, g1, lim(k)), O.g_mul(curve, 2, g2, lim(k))
        assert ceremony.same(curve, (g1, g2k), (g1k, g2)) is True
        assert ceremony.same(curve, (g1, g2k), (O.g_mul(curve, 1, g1, lim(k + 1)), g2)) is False
        assert ceremony.same(curve, (g1, g2), (g1k, g2)) is False
