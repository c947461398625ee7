"""GPU parity for the group-operation primitives the MSM kernels are made of -- the menu the reference itself
benchmarks and cross-checks (manta-benchmark/src/ecc.rs:30-128, consistency tests :138-172; SURVEY.md row
a-12): mixed addition, general addition, doubling, scalar multiplication and batch normalisation, all through
`mg_ec_elementwise`, every output affine point bit-exact against the CPU oracle."""
import numpy as np
import pytest

import helpers as H
import oracle_lib as O
from manta_rs_amd import synth

pytestmark = pytest.mark.gpu

CASES = [(0, 1), (1, 1), (0, 2), (1, 2)]


def _neg(curve, group, p):
    """-P through the oracle: [r-1]P."""
    r = synth.FR_MODULUS[curve]
    return O.g_mul(curve, group, p, synth.ints_to_limbs([r - 1], 4)[0])


@pytest.mark.parametrize("curve,group", CASES)
def test_addition_is_consistent_for_mixed_and_general_formulas(gpu, curve, group):
    """ecc.rs:138-156: affine+affine, projective+=affine and projective+=projective agree -- here each GPU
    formula against the oracle's sum, including the exceptional inputs (P+P, P+(-P), infinity on either side)."""
    n = 150
    a = H.random_points(curve, group, n, seed=301)
    b = H.random_points(curve, group, n, seed=302)
    b[3] = a[3]                      # P + P  -> must take the doubling path
    b[4] = _neg(curve, group, a[4])  # P + (-P) = infinity
    a[5] = 0                         # infinity + Q
    b[6] = 0                         # P + infinity
    a[7] = 0
    b[7] = 0                         # infinity + infinity
    want = np.stack([O.g_add(curve, group, a[i], b[i]) for i in range(n)])
    assert (want[4] == 0).all() and (want[7] == 0).all()
    mixed = gpu.ec_elementwise(curve, group, gpu.EC_ADD_MIXED, a, b)
    full = gpu.ec_elementwise(curve, group, gpu.EC_ADD, a, b)
    assert (mixed == want).all()
    assert (full == want).all()
    # subtraction = addition of the negated point
    wsub = np.stack([O.g_add(curve, group, a[i], _neg(curve, group, b[i])) if b[i].any() else a[i] for i in range(n)])
    assert (gpu.ec_elementwise(curve, group, gpu.EC_SUB_MIXED, a, b) == wsub).all()


@pytest.mark.parametrize("curve,group", CASES)
def test_doubling_matches_addition_with_itself(gpu, curve, group):
    n = 70
    a = H.random_points(curve, group, n, seed=303)
    a[2] = 0
    want = np.stack([O.g_add(curve, group, a[i], a[i]) for i in range(n)])
    assert (gpu.ec_elementwise(curve, group, gpu.EC_DOUBLE, a) == want).all()
    assert (gpu.ec_elementwise(curve, group, gpu.EC_ADD, a, a) == want).all()


@pytest.mark.parametrize("curve,group", CASES)
def test_scalar_multiplication_is_consistent(gpu, curve, group):
    """ecc.rs:158-172: affine*scalar == projective*scalar -- here double-and-add on the GPU against the oracle,
    with the edge scalars 0, 1, 2, r-1 and r (= the group order: [r]P = infinity)."""
    n = 40
    a = H.random_points(curve, group, n, seed=304)
    k = synth.msm_scalars(curve, n, "U", seed=305)
    r = synth.FR_MODULUS[curve]
    for i, v in enumerate([0, 1, 2, r - 1, r]):
        k[i] = synth.ints_to_limbs([v], 4)[0]
    a[9] = 0
    want = np.stack([O.g_mul(curve, group, a[i], k[i]) for i in range(n)])
    assert (want[0] == 0).all() and (want[4] == 0).all() and (want[1] == a[1]).all()
    got = gpu.ec_elementwise(curve, group, gpu.EC_MUL, a, k)
    assert (got == want).all()


def test_batch_normalisation_of_2_16_points(gpu):
    """ecc.rs:114-119 normalises 2^16 projective points at once: 2^16 doublings on the GPU, every affine output on
    the curve, spot-checked against the oracle, and the sum of all outputs equal to twice the sum of the inputs
    (a size-independent linearity check)."""
    curve, group, n = 1, 1, 1 << 16
    base = H.random_points(curve, group, 64, seed=306)
    a = np.ascontiguousarray(np.tile(base, (n // 64, 1)))
    got = gpu.ec_elementwise(curve, group, gpu.EC_DOUBLE, a)
    for i in (0, 1, 63, 64, n - 1):
        assert (got[i] == O.g_add(curve, group, a[i], a[i])).all()
    assert (got[:64] == got[-64:]).all()
    s_in = gpu.points_sum(curve, group, base)
    s_out = gpu.points_sum(curve, group, got[:64])
    assert (s_out == O.g_add(curve, group, s_in, s_in)).all()


def test_ec_elementwise_rejects_bad_arguments(gpu):
    a = H.random_points(0, 1, 4, seed=1)
    with pytest.raises(gpu.MantaGpuError):
        gpu.ec_elementwise(0, 1, 9, a, a)
    with pytest.raises(gpu.MantaGpuError):
        gpu.ec_elementwise(0, 1, gpu.EC_ADD, a, None)
    with pytest.raises(gpu.MantaGpuError):
        gpu.ec_elementwise(0, 3, gpu.EC_ADD, a, a)
