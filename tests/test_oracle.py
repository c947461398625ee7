"""CPU tests (no GPU): the C oracle against (a) the reference's committed BN254 verifying-key fixtures
(tests/golden/*.dat = manta-parameters/data/pay/verifying/*.dat; layout groth16.rs:337-361) and (b) the
golden vectors produced by the independent pure-Python restatement (tests/golden/gen_golden.py)."""
import json
import os
import struct

import numpy as np
import pytest

import oracle_lib as O
from manta_rs_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))
VEC = json.load(open(os.path.join(HERE, "golden", "vectors.json")))
NAMES = {0: "bn254", 1: "bls12_381"}
VK_FILES = {"to-private": 13, "private-transfer": 27, "to-public": 19,
            # the archived testnet keys (manta-parameters/data/archive/testnet/verifying): three more reference-held KATs
            "testnet-to-private": 13, "testnet-private-transfer": 27, "testnet-to-public": 19}


def canon(field, ints, nl):
    return synth.ints_to_limbs(ints, nl)


def fr_mont(curve, ints):
    return synth.to_mont(ints, synth.FR_MODULUS[curve], 4)


def pt_g1(curve, hx):
    if hx is None:
        return np.zeros(2 * synth.FQ_LIMBS[curve], dtype=np.uint64)
    q, nl = synth.FQ_MODULUS[curve], synth.FQ_LIMBS[curve]
    return synth.to_mont([int(hx[0], 16), int(hx[1], 16)], q, nl).reshape(-1)


def pt_g2(curve, hx):
    if hx is None:
        return np.zeros(4 * synth.FQ_LIMBS[curve], dtype=np.uint64)
    q, nl = synth.FQ_MODULUS[curve], synth.FQ_LIMBS[curve]
    return synth.to_mont([int(hx[0][0], 16), int(hx[0][1], 16), int(hx[1][0], 16), int(hx[1][1], 16)], q,
                         nl).reshape(-1)


# ---------------------------------------------------------------- reference fixtures (BN254 VKs)
@pytest.mark.parametrize("name,P", sorted(VK_FILES.items()))
def test_vk_fixture_points_and_pairing_kat(name, P):
    d = open(os.path.join(HERE, "golden", name + ".dat"), "rb").read()
    ok, alpha = O.deserialize(0, 1, d[0:32])
    assert ok and O.on_curve(0, 1, alpha)
    g2 = []
    for lo in (32, 96, 160):
        ok, pt = O.deserialize(0, 2, d[lo:lo + 64])
        assert ok and O.on_curve(0, 2, pt)
        assert O.serialize(0, 2, pt) == d[lo:lo + 64]  # round trip incl. the Fq2 sign rule (c1, then c0)
        g2.append(pt)
    beta, gamma, delta = g2
    if not name.startswith("testnet-"):  # the current keys come from the MPC (mpc.rs:419 sets gamma_g2 = the G2 generator);
        assert (gamma == O.generator(0, 2)).all()  # the archived testnet keys from ark-groth16's random setup (random gamma)
    (cnt,) = struct.unpack("<Q", d[224:232])
    assert cnt == P
    off = 232
    for _ in range(P):
        ok, g = O.deserialize(0, 1, d[off:off + 32])
        assert ok and O.on_curve(0, 1, g)
        assert O.serialize(0, 1, g) == d[off:off + 32]
        off += 32
    # alpha_g1_beta_g2 = e(alpha_g1, beta_g2) with arkworks' BN final exponentiation
    assert O.pairing_bytes(0, alpha, beta, ark_exp=True) == d[off:off + 384]
    assert len(d) == off + 384 + 2 * (8 + 91 * 192 + 1)


# ---------------------------------------------------------------- independent golden vectors
@pytest.mark.parametrize("curve", [0, 1])
def test_field_vectors(curve):
    v = VEC[NAMES[curve]]
    fq = "bn254_fq" if curve == 0 else "bls381_fq"
    fr = "bn254_fr" if curve == 0 else "bls381_fr"
    nl = synth.FQ_LIMBS[curve]
    q, r = synth.FQ_MODULUS[curve], synth.FR_MODULUS[curve]
    for a, b, c in v["fq_mul"]:
        am, bm = synth.to_mont([int(a, 16)], q, nl), synth.to_mont([int(b, 16)], q, nl)
        got = synth.from_mont(O.field_op(fq, "mul", am, bm), q)
        assert got == [int(c, 16)]
    for a, b, c in v["fr_mul"]:
        am, bm = fr_mont(curve, [int(a, 16)]), fr_mont(curve, [int(b, 16)])
        assert synth.from_mont(O.field_op(fr, "mul", am, bm), r) == [int(c, 16)]
        assert synth.limbs_to_ints(O.field_op(fr, "to_canonical", am)) == [int(a, 16)]
        assert (O.field_op(fr, "from_canonical", synth.ints_to_limbs([int(a, 16)], 4)) == am).all()
    for a, ai in v["fr_inv"]:
        assert synth.from_mont(O.field_op(fr, "inv", fr_mont(curve, [int(a, 16)])), r) == [int(ai, 16)]


@pytest.mark.parametrize("curve", [0, 1])
def test_group_vectors(curve):
    v = VEC[NAMES[curve]]
    for grp, key, conv in ((1, "g1", pt_g1), (2, "g2", pt_g2)):
        G = O.generator(curve, grp)
        assert O.on_curve(curve, grp, G)
        for k, hx in v[key + "_mul"]:
            kk = synth.ints_to_limbs([int(k, 16)], 4)[0]
            assert (O.g_mul(curve, grp, G, kk) == conv(curve, hx)).all()
        for k, ser in v[key + "_ser"]:
            kk = synth.ints_to_limbs([int(k, 16)], 4)[0]
            pt = O.g_mul(curve, grp, G, kk)
            assert O.serialize(curve, grp, pt).hex() == ser
            ok, back = O.deserialize(curve, grp, bytes.fromhex(ser))
            assert ok and (back == pt).all()
            unc = O.serialize(curve, grp, pt, compressed=False)
            ok, back = O.deserialize(curve, grp, unc, compressed=False)
            assert ok and (back == pt).all()


@pytest.mark.parametrize("curve", [0, 1])
def test_msm_vectors(curve):
    v = VEC[NAMES[curve]]
    for grp, key, conv in ((1, "msm_g1", pt_g1), (2, "msm_g2", pt_g2)):
        G = O.generator(curve, grp)
        bs = synth.ints_to_limbs([int(x, 16) for x in v[key]["base_scalars"]], 4)
        pts = O.fixed_base_mul(curve, grp, G, bs)
        sc = synth.ints_to_limbs([int(x, 16) for x in v[key]["scalars"]], 4)
        want = conv(curve, v[key]["result"])
        assert (O.msm(curve, grp, pts, sc, algo=0) == want).all()
        assert (O.msm(curve, grp, pts, sc, algo=1) == want).all()


@pytest.mark.parametrize("curve", [0, 1])
def test_ntt_vectors(curve):
    v = VEC[NAMES[curve]]["ntt"]
    r = synth.FR_MODULUS[curve]
    x = fr_mont(curve, [int(t, 16) for t in v["input"]])
    for key, inv, cos in (("fft", False, False), ("ifft", True, False), ("coset_fft", False, True),
                          ("coset_ifft", True, True)):
        got = synth.from_mont(O.ntt(curve, x, inverse=inv, coset=cos), r)
        assert got == [int(t, 16) for t in v[key]], key


@pytest.mark.parametrize("curve", [0, 1])
def test_groth16_vector(curve):
    """Toy circuit: the C oracle's setup + witness map + prover reproduce the independent Python proof
    bytes, and the proof satisfies the pairing equation."""
    v = VEC[NAMES[curve]]["groth16"]
    r = synth.FR_MODULUS[curve]
    c = synth.make_circuit(curve, v["m"], v["V"], v["P"], seed=v["seed"])
    assert synth.check_satisfied(c)
    pk = O.groth16_setup(c, fr_mont(curve, [int(t, 16) for t in v["toxic"]]))
    assert synth.from_mont(O.witness_map(c), r) == [int(t, 16) for t in v["h"]]
    rs = fr_mont(curve, [int(v["r"], 16), int(v["s"], 16)])
    for algo in (0, 1):
        assert O.groth16_prove(c, pk, rs[0], rs[1], msm_algo=algo).hex() == v["proof"]
    assert O.groth16_verify(curve, pk, c.z[1:c.P], bytes.fromhex(v["proof"])) == 1


# ---------------------------------------------------------------- oracle self-consistency at larger sizes
@pytest.mark.parametrize("curve,group,n", [(0, 1, 300), (1, 1, 300), (0, 2, 80), (1, 2, 60)])
def test_pippenger_equals_naive(curve, group, n):
    import helpers as H
    pts = H.random_points(curve, group, n, seed=n)
    pts[3] = 0
    sc = synth.msm_scalars(curve, n, "W", seed=n + 1)
    assert (O.msm(curve, group, pts, sc, algo=0) == O.msm(curve, group, pts, sc, algo=1)).all()


@pytest.mark.parametrize("curve", [0, 1])
def test_prove_verify_and_input_fuzzing(curve):
    """Mirrors manta-pay/src/test/transfer.rs:61-109 (prove then verify) and :346-417 (fuzzed public inputs
    must invalidate the proof)."""
    import helpers as H
    c = synth.make_circuit(curve, 300, 250, 9, seed=5)
    pk = O.groth16_setup(c, H.toxic(curve))
    rs = H.rand_fr_mont(curve, 2, seed=17)
    proof = O.groth16_prove(c, pk, rs[0], rs[1])
    assert O.groth16_verify(curve, pk, c.z[1:c.P], proof) == 1
    for j in range(1, c.P):
        bad = c.z[1:c.P].copy()
        bad[j - 1] = rs[0]
        assert O.groth16_verify(curve, pk, bad, proof) == 0
    tampered = bytearray(proof)
    tampered[5] ^= 1
    assert O.groth16_verify(curve, pk, c.z[1:c.P], bytes(tampered)) in (0, -1)
    # a violated constraint gives a proof that is rejected (no error raised -- like the reference in release)
    z_bad = c.z.copy()
    z_bad[c.P + 2] = rs[1]
    assert O.groth16_verify(curve, pk, c.z[1:c.P], O.groth16_prove(c, pk, rs[0], rs[1], z=z_bad)) == 0
