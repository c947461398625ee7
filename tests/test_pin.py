"""CPU tests that pin the oracle (and the product's host code) to REFERENCE-HELD known answers beyond the BN254 pairing
KAT of test_oracle.py:
  * the BLS12-381 Fr Poseidon vectors of manta-pay (tests/golden/poseidon_bls381_fr.json, extracted from
    manta-pay/src/crypto/poseidon/*_hardcoded_test*) -- the only BLS12-381 known answers the reference holds: they pin
    the oracle's BLS12-381 Fr add / mul / inverse and its Montgomery conversions;
  * the key statement readable in-repo, manta-trusted-setup/src/groth16/mpc.rs:355-431 (`initialize`), restated here
    with Python integers -- Lagrange basis = domain.ifft(powers of tau), dummy input rows, query definitions,
    h_query[i] = tau^(i+D) - tau^i -- against the oracle's setup, which every GPU proof test takes its keys from;
  * the six committed verifying keys through the PRODUCT's host serialiser (mg_point_serialize)."""
import json
import os

import numpy as np
import pytest

import helpers as H
import oracle_lib as O
from manta_rs_amd import synth
from vk_fixtures import VK, VK_FILES

HERE = os.path.dirname(os.path.abspath(__file__))
POS = json.load(open(os.path.join(HERE, "golden", "poseidon_bls381_fr.json")))
R_BLS = synth.FR_MODULUS[1]


def poseidon_permutation(add, mul, to_field, from_field):
    """The reference's test permutation (hash.rs:249-258 / permutation_hardcoded_test/poseidonperm_bls381_width3.sage):
    63 rounds of add-round-constants, x^5 S-box (all three words in the 4 + 4 full rounds, word 0 in the 55 partial
    ones), MDS product -- written over abstract field operations on [3, 4]-limb arrays so that the SAME routine drives
    the oracle (here) and the GPU field surface (test_gpu_pin.py)."""
    rc = to_field([int(x) for x in POS["round_constants"]]).reshape(63, 3, 4)
    mds = to_field([int(x) for row in POS["mds"] for x in row]).reshape(3, 3, 4)
    st = to_field([int(x) for x in POS["input"]])
    for r in range(63):
        st = add(st, rc[r])
        x2 = mul(st, st)
        x5 = mul(mul(x2, x2), st)
        if 4 <= r < 59:
            x5[1:] = st[1:]
        st = x5
        acc = None
        for j in range(3):  # new[i] = sum_j mds[i][j] st[j]
            term = mul(np.ascontiguousarray(mds[:, j]), np.repeat(st[j:j + 1], 3, axis=0))
            acc = term if acc is None else add(acc, term)
        st = acc
    return from_field(st)


def test_oracle_bls12_381_fr_reproduces_the_reference_poseidon_vectors():
    to_f = lambda ints: synth.to_mont(ints, R_BLS, 4)
    got = poseidon_permutation(lambda a, b: O.field_op("bls381_fr", "add", a, b), lambda a, b: O.field_op("bls381_fr", "mul", a, b),
                               to_f, lambda a: synth.from_mont(a, R_BLS))
    assert got == [int(x) for x in POS["output"]]
    # the MDS matrix is the Cauchy matrix 1 / (x_i + y_j), x_i = i, y_j = width + j (mds.rs generate_mds): an inverse KAT
    sums = to_f([i + 3 + j for i in range(3) for j in range(3)])
    assert synth.from_mont(O.field_op("bls381_fr", "inv", sums), R_BLS) == [int(x) for row in POS["mds"] for x in row]
    # Montgomery conversions on the same numbers
    ints = [int(x) for x in POS["round_constants"][:16]]
    assert (O.field_op("bls381_fr", "from_canonical", synth.ints_to_limbs(ints, 4)) == to_f(ints)).all()
    assert synth.limbs_to_ints(O.field_op("bls381_fr", "to_canonical", to_f(ints))) == ints


@pytest.mark.parametrize("curve", [0, 1])
def test_oracle_setup_equals_the_in_repo_key_statement(curve):
    """manta-trusted-setup/src/groth16/mpc.rs `initialize` (:355-431) builds a proving key from powers of tau with
    gamma = delta = 1 (:418-419, :424):
        domain = Radix2EvaluationDomain::new(m + P)                                  (:367-368)
        h_query[i] = tau^(i+D) G - tau^i G, i < D                                    (:372-377)
        tau_lagrange = domain.ifft(tau powers)  ->  L_i(tau) G  (likewise alpha, beta) (:378-381)
        a_g1[j] = L_{m+j}(tau) for j < P  (add_dummy_constraints :299-312)
        a_g1[j] += A[i][j] L_i,  b += B[i][j] L_i,  ext[j] += beta A L_i + alpha B L_i + C L_i  (:251-294)
        gamma_abc_g1 = ext[..P], l_query = ext[P..]                                  (:413-414, :421, :430)
    Restated with Python integers (the inverse DFT written out as the O(D^2) sum over the domain's root) and compared
    point for point with the oracle's setup at gamma = delta = 1; also at general gamma, delta (ark-groth16's
    generate_parameters divides the cross terms by gamma / delta and h by delta)."""
    r = synth.FR_MODULUS[curve]
    c = synth.make_circuit(curve, 11, 9, 3, seed=42)  # m + P = 14 -> D = 16
    D, m, P, V = c.D, c.m, c.P, c.V
    assert D == 16
    Rinv = pow(1 << 256, -1, r)
    rows = lambda M: [[(int(M.col[k]), synth.limbs_to_ints(M.val[k:k + 1])[0] * Rinv % r) for k in range(M.row_ptr[i], M.row_ptr[i + 1])]
                      for i in range(m)]
    A, B, C = rows(c.A), rows(c.B), rows(c.C)
    gen, s = (5, 28) if curve == 0 else (7, 32)           # ark-bn254 / ark-bls12-381 Fr: GENERATOR, TWO_ADICITY (App. A.1)
    w = pow(gen, (r - 1) >> s, r)                         # 2^s-th root of unity
    w = pow(w, 1 << (s - 4), r)                           # omega_D for D = 2^4
    assert pow(w, D, r) == 1 and pow(w, D // 2, r) != 1
    for tox_seed, unit in ((1, True), (2, False)):
        rng = synth.XorShift(1000 + tox_seed)
        tau, alpha, beta = rng.field(r), rng.field(r), rng.field(r)
        gamma, delta = (1, 1) if unit else (rng.field(r), rng.field(r))
        Dinv = pow(D, -1, r)
        # domain.ifft(powers)[i] = D^-1 sum_k tau^k w^(-i k)
        lag = [Dinv * sum(pow(tau, k, r) * pow(w, -i * k, r) for k in range(D)) % r for i in range(D)]
        a = [0] * V
        b = [0] * V
        cc = [0] * V
        for j in range(P):
            a[j] = lag[m + j]
        for i in range(m):
            for j, cf in A[i]:
                a[j] = (a[j] + cf * lag[i]) % r
            for j, cf in B[i]:
                b[j] = (b[j] + cf * lag[i]) % r
            for j, cf in C[i]:
                cc[j] = (cc[j] + cf * lag[i]) % r
        ext = [(beta * a[j] + alpha * b[j] + cc[j]) % r for j in range(V)]
        ginv, dinv = pow(gamma, -1, r), pow(delta, -1, r)
        pk = O.groth16_setup(c, synth.to_mont([tau, alpha, beta, gamma, delta], r, 4))
        G1, G2 = O.generator(curve, 1), O.generator(curve, 2)
        mul1 = lambda k: O.g_mul(curve, 1, G1, synth.ints_to_limbs([k % r], 4)[0])
        mul2 = lambda k: O.g_mul(curve, 2, G2, synth.ints_to_limbs([k % r], 4)[0])
        assert (pk.alpha_g1[0] == mul1(alpha)).all() and (pk.beta_g1[0] == mul1(beta)).all() and (pk.delta_g1[0] == mul1(delta)).all()
        assert (pk.beta_g2[0] == mul2(beta)).all() and (pk.gamma_g2[0] == mul2(gamma)).all() and (pk.delta_g2[0] == mul2(delta)).all()
        for j in range(V):
            assert (pk.a_query[j] == mul1(a[j])).all(), ("a", j)
            assert (pk.b_g1_query[j] == mul1(b[j])).all(), ("b1", j)
            assert (pk.b_g2_query[j] == mul2(b[j])).all(), ("b2", j)
            if j < P:
                assert (pk.gamma_abc_g1[j] == mul1(ext[j] * ginv)).all(), ("abc", j)
            else:
                assert (pk.l_query[j - P] == mul1(ext[j] * dinv)).all(), ("l", j)
        assert pk.h_query.shape[0] == D - 1  # ark setup: D - 1 entries; the MPC key carries one more (mpc.rs:372-377)
        for i in range(D - 1):
            assert (pk.h_query[i] == mul1((pow(tau, i + D, r) - pow(tau, i, r)) * dinv)).all(), ("h", i)
        # and the prover over this key verifies (the verification equation is the reference's own acceptance test)
        rs = H.rand_fr_mont(curve, 2, seed=tox_seed)
        assert O.groth16_verify(curve, pk, c.z[1:P], O.groth16_prove(c, pk, rs[0], rs[1])) == 1


@pytest.mark.parametrize("name", sorted(VK_FILES))
def test_product_serialiser_reproduces_the_reference_vk_bytes(name):
    """Every G1 / G2 point of the six committed verifying keys through the PRODUCT's encoder (mg_point_serialize, the code
    that writes proof bytes): decompressed limbs -> the file's bytes, compressed; uncompressed = x || y little-endian."""
    from manta_rs_amd import api
    vk = VK(name)
    assert vk.P == VK_FILES[name]
    assert api.point_serialize(0, 1, vk.alpha, True) == vk.alpha_bytes
    for pt, b in zip(vk.g2, vk.g2_bytes):
        assert api.point_serialize(0, 2, pt, True) == b
        assert api.point_serialize(0, 2, pt, False) == O.serialize(0, 2, pt, False)
    for pt, b in zip(vk.abc, vk.abc_bytes):
        assert api.point_serialize(0, 1, pt, True) == b
        assert api.point_serialize(0, 1, pt, False)[:32] == b[:31] + bytes([b[31] & 0x3F])  # x without the flag bits
    # the host-side group law of the product on reference-held points: sum of gamma_abc = oracle's sum
    assert (api.points_sum(0, 1, np.stack(vk.abc)) == O.g_sum(0, 1, np.stack(vk.abc))).all()


# BLAKE3 digests of the verifying-key files as the reference lists them (manta-parameters/data.checkfile:15-17,35-37; the
# files themselves are tests/golden/*.dat) and of the empty input (the checkfile's digest of its zero-length parameter files)
CHECKFILE = {
    "testnet-private-transfer.dat": "6ab2557f70f5583779f7cbcd27e73e66f8fcb5704ab21792a4126e89cc15b793",
    "testnet-to-private.dat": "5e2e618e067c9414fed3ca5b570f8f2b33c9a0a42b925dc723f6fdfdd7436c5c",
    "testnet-to-public.dat": "d1467307aa8b51b26fb0247eede05cdb3a8d94a3db2a5939f5e181da6024a9a5",
    "private-transfer.dat": "117d2789bd52fcc66b39f1526a876c23570ae39fcc67b27ba2846e9767e458e2",
    "to-private.dat": "c9c8333f74f83c600c37f18f5c64538c99450a317c0ecb3eef9eb43ac58817b2",
    "to-public.dat": "399e3b65fdc16e068472c429315964bd5a12683c3e67fdfe2b2aede92b164887",
}


def test_product_blake3_reproduces_the_reference_checkfile():
    """`manta_parameters::verify` is `blake3::hash(data) == checksum` (manta-parameters/src/lib.rs:173-177). The library's
    BLAKE3 (host code behind mg_ctx_create_from_bytes_checked; no GPU needed) must give the digests data.checkfile holds for
    the six key files the repository carries -- 36 KB each: 36 chunks, i.e. the tree mode -- and for the empty input."""
    from manta_rs_amd import api
    assert api.blake3(b"").hex() == "af1349b9f5f9a1a6a0404dea36dcc9499bcb25c9adc112b7cc9a93cae41f3262"
    for name, want in CHECKFILE.items():
        data = open(os.path.join(HERE, "golden", name), "rb").read()
        assert api.blake3(data).hex() == want, name
        assert api.blake3(data[:-1]).hex() != want
