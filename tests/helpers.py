"""Shared test helpers: seeded inputs for the parity tests (oracle = checker, HIP library = product)."""
import numpy as np

import oracle_lib as O
from manta_rs_amd import synth


def random_points(curve, group, n, seed=1):
    """n pseudo-random affine points: [k_i]G with seeded k_i via the oracle's fixed-base multiply."""
    rng = synth.XorShift(seed)
    p = synth.FR_MODULUS[curve]
    ks = synth.ints_to_limbs([rng.field(p) for _ in range(n)], 4)
    return O.fixed_base_mul(curve, group, O.generator(curve, group), ks)


def toxic(curve, seed=7):
    rng = synth.XorShift(seed)
    p = synth.FR_MODULUS[curve]
    return synth.to_mont([rng.field(p) for _ in range(5)], p, 4)


def rand_fr_mont(curve, n, seed=11):
    rng = synth.XorShift(seed)
    p = synth.FR_MODULUS[curve]
    return synth.to_mont([rng.field(p) for _ in range(n)], p, 4)
