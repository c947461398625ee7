"""Whole-proof parity on witnesses of STATED density.

The reference proves the assignment `Transfer::known_constraints` synthesises (manta-accounting/src/transfer/mod.rs:667-673,
driven by manta-pay/src/test/payment.rs:222-273): Poseidon states and in-circuit curve coordinates are dense field elements,
bits are booleans. No such witness can be captured here, so synth offers three profiles with the density measured on the
resulting z (synth.histogram): "sparse" (rounds 1-3: 69 % zeros / 23 % ones), "W" (40 / 25 / 10 / 25) and
"dense" (config 1: a multiplication chain, no trivial scalar). The pair-count dependent paths of the MSMs -- digit compaction,
the chunk length derived on the device, merge sizing, full tables against bucket tables -- see 4x (W) and 11x (dense) the
digit pairs of the sparse profile; every case below is byte-compared with the CPU oracle through the C ABI."""
import numpy as np
import pytest

import helpers as H
import oracle_lib as O
from manta_rs_amd import keygen, synth

pytestmark = pytest.mark.gpu


def _fast_oracle():
    O.set_threads(O.usable_cpus())  # the dense PrivateTransfer proof is ~4 s on one core


@pytest.fixture(scope="module")
def pt_keys():
    """one key per profile for the PrivateTransfer shape (the matrices differ per profile, so do the keys)"""
    out = {}
    for prof in ("W", "dense"):
        c = synth.make_shape(0, "private_transfer", profile=prof)
        out[prof] = (c, keygen.generate(c, synth.from_mont(H.toxic(0, seed=40 + len(prof)), synth.FR_MODULUS[0])))
    return out


def test_profiles_have_the_stated_density():
    for shape in ("to_private", "private_transfer"):
        w = synth.histogram(synth.make_shape(0, shape, profile="W").z_int)
        assert abs(w["zero"] - 0.40) < 2e-3 and abs(w["one"] - 0.25) < 2e-3 and abs(w["small"] - 0.10) < 2e-3 and abs(w["dense"] - 0.25) < 2e-3
        d = synth.histogram(synth.make_shape(0, shape, profile="dense").z_int)
        assert d["zero"] == 0 and d["small"] == 0 and d["one"] * d["n"] == 1  # z_0 = 1 is the only trivial scalar
    s = synth.histogram(synth.make_shape(0, "private_transfer").z_int)
    assert s["zero"] > 0.6  # what rounds 1-3 measured on


@pytest.mark.parametrize("prof", ["W", "dense"])
def test_private_transfer_single_proof_on_full_and_bucket_tables(gpu, pt_keys, prof):
    """single proofs: on the context's full tables (default budget) and with none (`full_table_bytes = 0`: the bucket
    tables, sort + bucket reduce on the chain); eager runs, graph capture and replay all give the oracle's bytes"""
    _fast_oracle()
    c, pk = pt_keys[prof]
    assert synth.check_satisfied(c)
    rs = H.rand_fr_mont(0, 2, seed=77)
    want = O.groth16_prove(c, pk, rs[0], rs[1])
    assert O.groth16_verify(0, pk, c.z[1:c.P], want) == 1
    for budget in (None, 0):
        ctx = gpu.ProvingContext(0, pk, full_table_bytes=budget)
        ctx.set_r1cs(gpu.R1CS.from_circuit(c))
        tb = ctx.table_bytes()
        assert (tb[1] == 0) == (budget == 0)
        for _ in range(4):  # two eager runs size the buffers, the third captures, the fourth replays
            assert gpu.Groth16.prove_with_randomness(ctx, c.z, rs[0], rs[1]) == want, (prof, budget)
        ctx.close()


@pytest.mark.parametrize("prof", ["W", "dense"])
def test_private_transfer_batch_of_32_distinct_assignments(gpu, pt_keys, prof):
    """one pass of 32 proofs, 32 distinct assignments of the profile (Reassigner keeps the density); every member pairing-checked,
    five byte-compared with the oracle; a pass of 5 coalesced-size proofs (narrow tables) as well"""
    _fast_oracle()
    c, pk = pt_keys[prof]
    R = synth.Reassigner(c)
    k = 32
    cs = [c] + [R.assign(0x4D414E5441_2000 + q) for q in range(1, k)]
    h = synth.histogram(cs[7].z_int)
    if prof == "W":
        assert abs(h["zero"] - 0.40) < 0.01 and abs(h["one"] - 0.25) < 0.01 and abs(h["small"] - 0.10) < 0.01
    else:
        assert h["zero"] == 0
    rs = H.rand_fr_mont(0, 2 * k, seed=91)
    ctx = gpu.ProvingContext(0, pk)
    ctx.set_r1cs(gpu.R1CS.from_circuit(c))
    zs = np.stack([x.z for x in cs])
    for rep in range(3):
        got = gpu.Groth16.prove_batch(ctx, zs, rs[:k], rs[k:])
        if rep == 0:
            for q in range(k):
                assert O.groth16_verify(0, pk, cs[q].z[1:c.P], got[q]) == 1, q
            for q in (0, 1, 13, 30, 31):
                assert got[q] == O.groth16_prove(cs[q], pk, rs[q], rs[k + q]), q
            first = got
        assert got == first
    got5 = gpu.Groth16.prove_batch(ctx, zs[3:8], rs[3:8], rs[k + 3:k + 8])
    assert got5 == first[3:8]
    ctx.close()


@pytest.mark.parametrize("prof", ["W", "dense"])
def test_private_transfer_sharded_8_ways(gpu, pt_keys, prof):
    """BASELINE configs[3] on the profile: every MSM range-sharded over 8 device entries (device 0 eight times: one GPU per
    box), partial points summed; bytes equal the oracle's and the single-device context's"""
    _fast_oracle()
    c, pk = pt_keys[prof]
    rs = H.rand_fr_mont(0, 4, seed=93)
    ctx = gpu.ProvingContext(0, pk, devices=[0] * 8)
    ctx.set_r1cs(gpu.R1CS.from_circuit(c))
    assert ctx.num_shards == 8
    p0 = gpu.Groth16.prove_with_randomness(ctx, c.z, rs[0], rs[1])
    assert p0 == O.groth16_prove(c, pk, rs[0], rs[1])
    got = gpu.Groth16.prove_batch(ctx, np.stack([c.z, c.z]), rs[0:4:2], rs[1:4:2])
    assert got[0] == p0 and got[1] == O.groth16_prove(c, pk, rs[2], rs[3])
    ctx.close()


@pytest.mark.parametrize("prof", ["W", "dense"])
def test_bls12_381_2_15_circuit(gpu, prof):
    """the 2^15 BLS12-381 circuit (D = V = 2^15, P = 16: the small sibling of BASELINE configs[2]) on the profile: single
    proof and a batch of 3 against the oracle"""
    _fast_oracle()
    curve, D, P = 1, 1 << 15, 16
    c = synth.make_circuit(curve, D - P, D, P, seed=0x4D414E5441_0315, profile=prof)
    pk = keygen.generate(c, synth.from_mont(H.toxic(curve, seed=52), synth.FR_MODULUS[curve]))
    ctx = gpu.ProvingContext(curve, pk)
    ctx.set_r1cs(gpu.R1CS.from_circuit(c))
    rs = H.rand_fr_mont(curve, 6, seed=95)
    want = O.groth16_prove(c, pk, rs[0], rs[1])
    for _ in range(4):
        assert gpu.Groth16.prove_with_randomness(ctx, c.z, rs[0], rs[1]) == want
    assert O.groth16_verify(curve, pk, c.z[1:c.P], want) == 1
    R = synth.Reassigner(c)
    cs = [c, R.assign(5), R.assign(6)]
    got = gpu.Groth16.prove_batch(ctx, np.stack([x.z for x in cs]), rs[0:6:2], rs[1:6:2])
    assert got[0] == want
    for q in (1, 2):
        assert got[q] == O.groth16_prove(cs[q], pk, rs[2 * q], rs[2 * q + 1])
    ctx.close()
