"""GPU parity: NTT, witness map and whole Groth16 proofs vs the CPU oracle, bit-exact."""
import numpy as np
import pytest

import helpers as H
import oracle_lib as O
from manta_rs_amd import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("curve", [0, 1])
@pytest.mark.parametrize("log_n", [0, 1, 2, 5, 10, 13])
def test_ntt_matches_oracle(gpu, curve, log_n):
    n = 1 << log_n
    x = H.rand_fr_mont(curve, n, seed=log_n + 1)
    dom = gpu.Radix2EvaluationDomain(curve, n)
    for inverse in (False, True):
        for coset in (False, True):
            want = O.ntt(curve, x, inverse=inverse, coset=coset)
            got = dom._run(x, inverse, coset)
            assert (got == want).all(), (inverse, coset)
    assert (dom.ifft(dom.fft(x)) == x).all()
    assert (dom.coset_ifft(dom.coset_fft(x)) == x).all()


@pytest.mark.parametrize("curve", [0, 1])
@pytest.mark.parametrize("m,V,P", [(100, 70, 5), (1000, 700, 13), (3000, 3500, 27)])
def test_prove_matches_oracle_and_verifies(gpu, curve, m, V, P):
    c = synth.make_circuit(curve, m, V, P, seed=m)
    assert synth.check_satisfied(c)
    pk = O.groth16_setup(c, H.toxic(curve))
    ctx = gpu.ProvingContext(curve, pk)
    r1cs = gpu.R1CS.from_circuit(c)
    ctx.set_r1cs(r1cs)
    assert ctx.domain_size == c.D
    assert (ctx.witness_map(c.z) == O.witness_map(c)).all()
    rs = H.rand_fr_mont(curve, 2, seed=99)
    it = iter(rs)
    proof = gpu.Groth16.prove(ctx, r1cs, lambda: next(it))
    want = O.groth16_prove(c, pk, rs[0], rs[1], msm_algo=1)
    assert proof == want
    assert O.groth16_verify(curve, pk, c.z[1:c.P], proof) == 1
    # r = 0 skips g1_b (App. B.1) and must still agree
    zero = np.zeros(4, dtype=np.uint64)
    assert gpu.Groth16.prove_with_randomness(ctx, c.z, zero, rs[1]) == O.groth16_prove(c, pk, zero, rs[1])
    # an unsatisfying witness is not an error: same bytes as the reference algorithm, proof rejected
    z_bad = c.z.copy()
    z_bad[c.P + 1] = rs[0]
    bad = gpu.Groth16.prove_with_randomness(ctx, z_bad, rs[0], rs[1])
    assert bad == O.groth16_prove(c, pk, rs[0], rs[1], z=z_bad)
