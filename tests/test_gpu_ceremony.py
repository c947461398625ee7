"""GPU parity of the trusted-setup bulk group operations (SURVEY f-4) and of the table-driven fixed-base multiplication
behind key generation (f-3), against the oracle; and end to end: a key assembled from powers of tau by
`ceremony.initialize` (group IFFTs on the GPU), then moved by a `contribute`, proves on the GPU and verifies on the GPU."""
import numpy as np
import pytest

import helpers as H
import oracle_lib as O
from manta_rs_amd import ceremony, keygen, synth

pytestmark = pytest.mark.gpu
CASES = [(0, 1), (0, 2), (1, 1), (1, 2)]


@pytest.mark.parametrize("curve,group", CASES)
def test_group_ntt_matches_oracle(gpu, curve, group):
    """mg_group_ntt = Radix2EvaluationDomain::{fft, ifft} over points (mpc.rs:378-381): every size 2^0..2^6, infinity
    entries, repeated points; the round trip returns the input."""
    for lg in (0, 1, 2, 6):
        n = 1 << lg
        pts = H.random_points(curve, group, n, seed=40 + lg)
        if n >= 4:
            pts[1] = 0
            pts[3] = pts[2]
        for inv in (False, True):
            got = gpu.group_ntt(curve, group, pts, inverse=inv)
            assert (got == O.group_ntt(curve, group, pts, inverse=inv)).all(), (lg, inv)
        assert (gpu.group_ntt(curve, group, gpu.group_ntt(curve, group, pts), inverse=True) == pts).all()


@pytest.mark.parametrize("curve,group", CASES)
def test_batch_mul_fixed_scalar_and_pointwise(gpu, curve, group):
    r = synth.FR_MODULUS[curve]
    pts = H.random_points(curve, group, 50, seed=60)
    pts[7] = 0
    for k in (0, 1, 2, r - 1, 0xdeadbeefcafebabe1234567, r - 12345):
        got = ceremony.batch_mul_fixed_scalar(curve, group, pts, k)
        kk = synth.ints_to_limbs([k % r], 4)[0]
        for i in (0, 7, 13, 49):
            assert (got[i] == O.g_mul(curve, group, pts[i], kk)).all(), (k, i)
    ks = [synth.XorShift(i).field(r) for i in range(50)]
    got = ceremony.batch_mul_pointwise(curve, group, pts, ks)
    for i in range(0, 50, 7):
        assert (got[i] == O.g_mul(curve, group, pts[i], synth.ints_to_limbs([ks[i]], 4)[0])).all()


@pytest.mark.parametrize("curve,group", [(0, 1), (1, 1), (0, 2), (1, 2)])
def test_fixed_base_mul_table_path_matches_oracle(gpu, curve, group):
    """n >= 16 384 multiples of one base take the windowed-table kernels (32 table additions per scalar); all of them
    against the oracle's fixed-base multiply, with the digit edge cases 0, 1, 255, 256, 2^248, r - 1 in front."""
    r = synth.FR_MODULUS[curve]
    n = 16384 + 37 if group == 1 else 16384
    rng = synth.XorShift(70 + curve)
    ks = [0, 1, 255, 256, 257, 1 << 248, (1 << 248) - 1, r - 1, r - 2, 0xff00ff00ff00ff00] + [rng.field(r) for _ in range(n - 10)]
    lim = synth.ints_to_limbs(ks, 4)
    G = O.generator(curve, group)
    base = O.g_mul(curve, group, G, synth.ints_to_limbs([0x1337], 4)[0])
    got = gpu.fixed_base_mul(curve, group, base, gpu.DeviceBuffer.from_numpy(lim), n).to_numpy(shape=(n, gpu.affine_limbs(curve, group)))
    want = O.fixed_base_mul(curve, group, base, lim)
    assert (got == want).all()
    small = gpu.fixed_base_mul(curve, group, base, gpu.DeviceBuffer.from_numpy(lim[:500]), 500).to_numpy(shape=(500, gpu.affine_limbs(curve, group)))
    assert (small == want[:500]).all()   # the per-lane double-and-add path for short batches


@pytest.mark.parametrize("curve", [0, 1])
def test_ceremony_initialize_contribute_prove_verify(gpu, curve):
    """mpc.rs `initialize` + `contribute` on the GPU, end to end: powers of tau (built with the oracle) -> kzg update with
    a second (tau, alpha, beta) -> phase-2 key of a circuit via group IFFTs -> equals the oracle's scalar-side setup at
    the combined toxic waste with gamma = delta = 1 -> a delta contribution -> GPU proofs under the contributed key
    verify on the GPU against the contributed verifying key, and not against the old one."""
    r = synth.FR_MODULUS[curve]
    c = synth.make_circuit(curve, 27, 20, 4, seed=81)       # D = 32
    D = c.D
    G1, G2 = O.generator(curve, 1), O.generator(curve, 2)
    lim = lambda ks: synth.ints_to_limbs([k % r for k in ks], 4)
    t0, a0, b0 = 0x1111111111111111222, 0x3333333333333333444, 0x5555555555555555666
    tp = [pow(t0, i, r) for i in range(2 * D - 1)]
    acc = ceremony.Accumulator(curve, O.fixed_base_mul(curve, 1, G1, lim(tp)), O.fixed_base_mul(curve, 2, G2, lim(tp[:D])),
                               O.fixed_base_mul(curve, 1, G1, lim([a0 * t for t in tp[:D]])),
                               O.fixed_base_mul(curve, 1, G1, lim([b0 * t for t in tp[:D]])), O.g_mul(curve, 2, G2, lim([b0])[0]))
    t1, a1, b1 = 0x777777777777777888, 0x999999999999999aaa, 0xbbbbbbbbbbbbbbbccc
    acc.update(t1, a1, b1)                                   # kzg.rs:444-468
    tau, alpha, beta = t0 * t1 % r, a0 * a1 % r, b0 * b1 % r
    assert acc.check_powers() == ""                          # kzg.rs:508-521 on the GPU (MSMs + pairing products)
    keep = acc.alpha_tau_powers_g1.copy()
    acc.alpha_tau_powers_g1[D // 2] = acc.tau_powers_g1[D // 2]
    assert acc.check_powers() == "AlphaG1Powers"
    acc.alpha_tau_powers_g1 = keep
    assert (acc.tau_powers_g1[5] == O.g_mul(curve, 1, G1, lim([pow(tau, 5, r)])[0])).all()
    assert (acc.alpha_tau_powers_g1[3] == O.g_mul(curve, 1, G1, lim([alpha * pow(tau, 3, r)])[0])).all()
    assert (acc.tau_powers_g2[D - 1] == O.g_mul(curve, 2, G2, lim([pow(tau, D - 1, r)])[0])).all()
    assert (acc.beta_g2[0] == O.g_mul(curve, 2, G2, lim([beta])[0])).all()
    pk = ceremony.initialize(acc, c)
    want = O.groth16_setup(c, synth.to_mont([tau, alpha, beta, 1, 1], r, 4))
    for f in ("alpha_g1", "beta_g1", "delta_g1", "beta_g2", "gamma_g2", "delta_g2", "gamma_abc_g1", "a_query", "b_g1_query",
              "b_g2_query", "h_query", "l_query"):
        assert (np.asarray(getattr(pk, f)).reshape(-1) == np.asarray(getattr(want, f)).reshape(-1)).all(), f
    delta = 0xdddddddddddddddeee
    pk2 = ceremony.contribute(curve, pk, delta)
    want2 = O.groth16_setup(c, synth.to_mont([tau, alpha, beta, 1, delta], r, 4))
    for f in ("delta_g1", "delta_g2", "h_query", "l_query"):
        assert (np.asarray(getattr(pk2, f)).reshape(-1) == np.asarray(getattr(want2, f)).reshape(-1)).all(), f
    ctx = gpu.ProvingContext(curve, pk2)
    ctx.set_r1cs(gpu.R1CS.from_circuit(c))
    rs = H.rand_fr_mont(curve, 2, seed=98)
    proof = gpu.Groth16.prove_with_randomness(ctx, c.z, rs[0], rs[1])
    assert proof == O.groth16_prove(c, want2, rs[0], rs[1])
    assert gpu.groth16_verify(gpu.VerifyingContext(curve, pk2), c.z[1:c.P], proof) is True
    assert gpu.groth16_verify(gpu.VerifyingContext(curve, pk), c.z[1:c.P], proof) is False


@pytest.mark.parametrize("curve", [0, 1])
def test_verify_transform_checks_on_a_contribution(gpu, curve):
    """mpc.rs:487-508 `verify_transform`: the same-ratio checks between the key before and after `contribute` -- delta_g1
    against delta_g2, and one random linear combination each of the whole h and l queries (`merge_pairs_affine`: two MSMs
    over the same scalars) -- all on the GPU (`mg_msm`, `mg_pairing_check`). A consistent contribution passes; a key whose
    h / l / delta_g1 was scaled by something else is rejected with the reference's error variant."""
    r = synth.FR_MODULUS[curve]
    c = synth.make_circuit(curve, 120, 90, 5, seed=82)
    pk = O.groth16_setup(c, H.toxic(curve, seed=21))
    delta = 0x1234567890abcdef1234567
    nxt = ceremony.contribute(curve, pk, delta)
    G1 = O.generator(curve, 1)
    lim = lambda ks: synth.ints_to_limbs([k % r for k in ks], 4)
    ratio = (O.g_mul(curve, 1, G1, lim([77])[0]), O.g_mul(curve, 1, G1, lim([77 * delta])[0]))  # what a RatioProof certifies
    rho = [pow(5, i + 1, r) for i in range(max(pk.h_query.shape[0], pk.l_query.shape[0]))]
    assert ceremony.check_transform(curve, pk, nxt, ratio) == ""          # OS randomness, like the reference
    # merge_pairs_affine against the oracle's MSM on the same scalars
    L, R = ceremony.merge_pairs_affine(curve, 1, nxt.h_query, pk.h_query, rho[:pk.h_query.shape[0]])
    can = synth.ints_to_limbs(rho[:pk.h_query.shape[0]], 4)
    assert (L == O.msm(curve, 1, nxt.h_query, can)).all() and (R == O.msm(curve, 1, pk.h_query, can)).all()
    # wrong ratio, then one field at a time scaled by a different scalar
    bad_ratio = (ratio[0], O.g_mul(curve, 1, G1, lim([78 * delta])[0]))
    assert ceremony.check_transform(curve, pk, nxt, bad_ratio) == "InconsistentDeltaChange"
    import copy
    for field, err in (("delta_g1", "InconsistentDeltaChange"), ("h_query", "InconsistentHChange"), ("l_query", "InconsistentLChange")):
        t = copy.copy(nxt)
        pts = np.asarray(getattr(nxt, field)).reshape(-1, gpu.affine_limbs(curve, 1)).copy()
        pts[-1] = O.g_mul(curve, 1, pts[-1], lim([3])[0])  # one entry off
        setattr(t, field, pts if field != "delta_g1" else pts.reshape(-1))
        assert ceremony.check_transform(curve, pk, t, ratio) == err, field
    # same_ratio itself: e(a P, Q) == e(P, a Q), infinity pairs contribute 1
    G2 = O.generator(curve, 2)
    a = 0xabcdef12345
    aP, aQ = O.g_mul(curve, 1, G1, lim([a])[0]), O.g_mul(curve, 2, G2, lim([a])[0])
    assert ceremony.same_ratio(curve, (G1, aP), (G2, aQ)) is True
    assert ceremony.same_ratio(curve, (G1, aP), (G2, O.g_mul(curve, 2, G2, lim([a + 1])[0]))) is False
    assert gpu.pairing_check(curve, np.stack([G1, np.zeros_like(G1)]), np.stack([np.zeros_like(G2), G2])) is True
    assert gpu.pairing_check(curve, G1.reshape(1, -1), G2.reshape(1, -1)) is False


@pytest.mark.parametrize("curve", [0, 1])
def test_trusted_setup_phase_two_is_valid_like_the_reference(gpu, curve):
    """manta-trusted-setup/src/groth16/test/mod.rs:258-286 `trusted_setup_phase_two_is_valid`: five rounds of
    contribute -> verify_transform on a small circuit, then a proof under the final key verifies (and not under the
    initial one). Every bulk group operation, the merged-query checks and both pairings-based verifications run on the
    GPU; the hash-to-group ratio proof is replaced by the pair it certifies."""
    r = synth.FR_MODULUS[curve]
    c = synth.make_circuit(curve, 13, 10, 2, seed=83)
    first = state = O.groth16_setup(c, H.toxic(curve, seed=22))
    G1 = O.generator(curve, 1)
    lim = lambda k: synth.ints_to_limbs([k % r], 4)[0]
    rng = synth.XorShift(0xCE4E0 + curve)
    for rnd in range(5):
        prev, delta, t = state, rng.field(r), rng.field(r)
        state = ceremony.contribute(curve, prev, delta)
        ratio = (O.g_mul(curve, 1, G1, lim(t)), O.g_mul(curve, 1, G1, lim(t * delta)))
        assert ceremony.check_transform(curve, prev, state, ratio) == "", rnd
    assert ceremony.check_transform(curve, first, state) == ""      # verify_transform_all's end-to-end checks, mpc.rs:543-559
    ctx = gpu.ProvingContext(curve, state)
    ctx.set_r1cs(gpu.R1CS.from_circuit(c))
    rs = H.rand_fr_mont(curve, 2, seed=99)
    proof = gpu.Groth16.prove_with_randomness(ctx, c.z, rs[0], rs[1])
    assert gpu.groth16_verify(gpu.VerifyingContext(curve, state), c.z[1:c.P], proof) is True
    assert gpu.groth16_verify(gpu.VerifyingContext(curve, first), c.z[1:c.P], proof) is False
