"""GPU tests that feed REFERENCE-HELD data to the HIP path ("the VK fixtures never touch the HIP path"):
  * the BLS12-381 Fr Poseidon known answers of manta-pay through the device field arithmetic, in both device
    representations (saturated Montgomery; reduced-radix lazy with non-canonical representatives);
  * a direct field-level parity surface for SURVEY row a-10 on edge values, all four fields, against the oracle;
  * the points of the six committed verifying keys as MSM bases / group-operation operands on the GPU."""
import itertools
import json
import os

import numpy as np
import pytest

import helpers as H
import oracle_lib as O
from manta_rs_amd import synth
from test_pin import POS, R_BLS, poseidon_permutation
from vk_fixtures import VK, VK_FILES

pytestmark = pytest.mark.gpu

FIELDS = {"bn254_fr": (synth.FR_MODULUS[0], 4), "bn254_fq": (synth.FQ_MODULUS[0], 4),
          "bls381_fr": (synth.FR_MODULUS[1], 4), "bls381_fq": (synth.FQ_MODULUS[1], 6)}


@pytest.mark.parametrize("repr_,lazy", [(0, (0, 0)), (1, (0, 0)), (1, (3, 1)), (1, (2, 3))])
def test_gpu_field_arithmetic_reproduces_the_reference_poseidon_vectors(gpu, repr_, lazy):
    """hash.rs:249-258's known answer -- Poseidon(3, 1, 2) over BLS12-381 Fr, 63 rounds, ~600 field multiplications
    in sequence -- computed by the GPU's field functions: any wrong limb anywhere changes the three output words."""
    to_f = lambda ints: synth.to_mont(ints, R_BLS, 4)
    add = lambda a, b: gpu.field_op("bls381_fr", "add", a, b, repr=repr_, lazy_a=lazy[0], lazy_b=lazy[1])
    mul = lambda a, b: gpu.field_op("bls381_fr", "mul", a, b, repr=repr_, lazy_a=lazy[0], lazy_b=lazy[1])
    got = poseidon_permutation(add, mul, to_f, lambda a: synth.from_mont(a, R_BLS))
    assert got == [int(x) for x in POS["output"]]
    sums = to_f([i + 3 + j for i in range(3) for j in range(3)])  # MDS = Cauchy matrix 1/(i + 3 + j): inverse KAT
    inv = gpu.field_op("bls381_fr", "inv", sums, repr=repr_, lazy_a=lazy[0])
    assert synth.from_mont(inv, R_BLS) == [int(x) for row in POS["mds"] for x in row]
    ints = [int(x) for x in POS["round_constants"][:32]]
    assert (gpu.field_op("bls381_fr", "from_canonical", synth.ints_to_limbs(ints, 4), repr=repr_) == to_f(ints)).all()
    assert synth.limbs_to_ints(gpu.field_op("bls381_fr", "to_canonical", to_f(ints), repr=repr_)) == ints


@pytest.mark.parametrize("field", sorted(FIELDS))
def test_gpu_field_ops_match_oracle_on_edge_values(gpu, field):
    """SURVEY a-10, directly: every device field function vs the oracle on 0, 1, 2, p-1, p-2, (p-1)/2, R mod p, values
    with all-ones limbs, and random elements -- all pairs -- in the saturated representation and in the reduced-radix
    lazy one with every combination of lazy representatives a + i p, b + j p (i, j <= 3)."""
    p, nl = FIELDS[field]
    rng = synth.XorShift(99)
    vals = [0, 1, 2, p - 1, p - 2, (p - 1) // 2, (1 << (64 * nl)) % p, (1 << (64 * nl - 1)) % p, ((1 << (32 * nl)) - 1) % p,
            (1 << 64) - 1, (1 << 128) - 1] + [rng.field(p) for _ in range(20)]
    pairs = list(itertools.product(vals, vals))
    a = synth.to_mont([x for x, _ in pairs], p, nl)
    b = synth.to_mont([y for _, y in pairs], p, nl)
    configs = [(0, 0, 0)] + [(1, i, j) for i in range(4) for j in range(4)]
    for op in ("add", "sub", "mul"):
        want = O.field_op(field, op, a, b)
        for repr_, la, lb in configs:
            got = gpu.field_op(field, op, a, b, repr=repr_, lazy_a=la, lazy_b=lb)
            assert (got == want).all(), (op, repr_, la, lb)
    one = synth.to_mont(vals, p, nl)
    for op in ("sqr", "neg"):
        want = O.field_op(field, op, one)
        for repr_, la, _ in configs[:5]:
            assert (gpu.field_op(field, op, one, repr=repr_, lazy_a=la) == want).all(), (op, repr_, la)
    nz = synth.to_mont([v for v in vals if v], p, nl)
    for repr_, la in ((0, 0), (1, 0), (1, 3)):
        assert (gpu.field_op(field, "inv", nz, repr=repr_, lazy_a=la) == O.field_op(field, "inv", nz)).all()
        ints = synth.ints_to_limbs(vals, nl)
        assert (gpu.field_op(field, "from_canonical", ints, repr=repr_) == one).all()
        assert (gpu.field_op(field, "to_canonical", one, repr=repr_) == ints).all()
    with pytest.raises(gpu.MantaGpuError):
        gpu.field_op(field, "mul", a, b, repr=0, lazy_a=1)  # lazy representatives exist in the reduced-radix form only


@pytest.mark.parametrize("name", sorted(VK_FILES))
def test_reference_vk_points_on_the_gpu(gpu, name):
    """The G1 / G2 points the reference ships (decompressed from the committed verifying keys) as operands of the GPU
    group law and as MSM bases: gamma_abc_g1 . (1, public inputs) is exactly the input-preparation MSM of
    `Groth16::verify` (groth16.rs:603-609 -> ark-groth16 prepare_inputs)."""
    vk = VK(name)
    abc = np.stack(vk.abc)
    P = vk.P
    r = synth.FR_MODULUS[0]
    rng = synth.XorShift(hash(name) & 0xffff)
    sc = synth.ints_to_limbs([1] + [rng.field(r) for _ in range(P - 1)], 4)
    want = O.msm(0, 1, abc, sc)
    for pre in (0, 5):
        assert (gpu.VariableBaseMSM.multi_scalar_mul(gpu.Bases(0, 1, abc, precompute_window_bits=pre), sc) == want).all()
    # the same points tiled to 2^12 bases (repeated bases = P + P inside buckets), witness-like scalars
    n = 1 << 12
    big = abc[np.arange(n) % P]
    scb = synth.msm_scalars(0, n, "W", seed=7)
    assert (gpu.VariableBaseMSM.multi_scalar_mul(gpu.Bases(0, 1, big, precompute_window_bits=8), scb) == O.msm(0, 1, big, scb)).all()
    # group law element-wise: abc[i] + abc[i+1], 2 abc[i], [k] abc[i], abc[i] - abc[i]
    nxt = np.roll(abc, -1, axis=0)
    add = gpu.ec_elementwise(0, 1, gpu.EC_ADD_MIXED, abc, nxt)
    gen = gpu.ec_elementwise(0, 1, gpu.EC_ADD, abc, nxt)
    dbl = gpu.ec_elementwise(0, 1, gpu.EC_DOUBLE, abc)
    mul = gpu.ec_elementwise(0, 1, gpu.EC_MUL, abc, sc)
    for i in range(P):
        assert (add[i] == O.g_add(0, 1, abc[i], nxt[i])).all() and (gen[i] == add[i]).all()
        assert (dbl[i] == O.g_add(0, 1, abc[i], abc[i])).all()
        assert (mul[i] == O.g_mul(0, 1, abc[i], sc[i])).all()
    assert not gpu.ec_elementwise(0, 1, gpu.EC_SUB_MIXED, abc, abc).any()
    # G2: beta, gamma, delta as a 3-term MSM and pairwise sums
    g2 = np.stack(vk.g2)
    sc2 = synth.ints_to_limbs([rng.field(r) for _ in range(3)], 4)
    assert (gpu.VariableBaseMSM.multi_scalar_mul(gpu.Bases(0, 2, g2), sc2) == O.msm(0, 2, g2, sc2)).all()
    s2 = gpu.ec_elementwise(0, 2, gpu.EC_ADD, g2, np.roll(g2, -1, axis=0))
    for i in range(3):
        assert (s2[i] == O.g_add(0, 2, g2[i], g2[(i + 1) % 3])).all()
    # product encoder on GPU results: serialise what the GPU computed, compare with the oracle's bytes
    assert gpu.point_serialize(0, 1, add[0], True) == O.serialize(0, 1, O.g_add(0, 1, abc[0], nxt[0]), True)
