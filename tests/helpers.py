"""Shared test helpers: seeded inputs for the parity tests (oracle = checker, HIP library = product)."""
import os

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


def knob_env(knobs, base=None, strip_prefix=None):
    """Environment of a child process that runs under `knobs`. The names of the tuning table (mg_tuning_env_names) act on the SHIPPED
    library; every other MANTA_* / MG_DIAG_* knob is an A/B switch of a measurement campaign that the shipped library has compiled in
    at its default -- it exists in the diagnosis twin (every unit built with -DMG_DIAG), which the child then loads through MANTA_LIB.
    strip_prefix: drop inherited variables with this prefix first."""
    from manta_rs_amd import api
    env = dict(os.environ if base is None else base)
    if strip_prefix:
        env = {k: v for k, v in env.items() if not k.startswith(strip_prefix)}
    env.update(knobs)
    shipped = set(api.tuning_env_names())
    if any(k not in shipped for k in knobs if k.startswith(("MANTA_", "MG_DIAG_")) and k != "MANTA_LIB"):
        diag = os.path.join(os.path.dirname(api.LIB_PATH), "libmantagpu_diag.so")
        assert os.path.exists(diag), "manta_rs_amd/csrc/Makefile builds the diagnosis twin next to the library"
        env.setdefault("MANTA_LIB", diag)
    return env


def proof_diff(got, want, knobs=None, child_stdout=None, tag="parity"):
    """A readable account of a proof mismatch: which proofs of the list differ and in which of A | B | C (BN254: 32 | 64 | 32 bytes,
    hex strings) -- and the child's whole output saved under gpurun_out/ so that an intermittent failure on the GPU box leaves evidence
    (round 6: one unexplained wrong B element in one child of one suite run, 230 repetitions clean: tools/first_proof_stress.py)."""
    rows = []
    for i, (g, w) in enumerate(zip(got, want)):
        if g != w:
            n = len(w) // 4
            rows.append((i, [k for k, (a, b) in zip("ABC", ((0, n), (n, 3 * n), (3 * n, 4 * n))) if g[a:b] != w[a:b]]))
    msg = "knobs %r: %d of %d proofs differ: %s (lengths %d / %d)" % (knobs, len(rows), len(want), rows[:8], len(got), len(want))
    if child_stdout is not None:
        try:
            root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
            os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
            path = os.path.join(root, "gpurun_out", "%s_failure.txt" % tag)
            with open(path, "a") as f:
                f.write("==== %s\n%s\n" % (msg, child_stdout))
            msg += " -- child output appended to " + path
        except OSError:
            pass
    return msg
