"""Toy Groth16 key generation from explicit toxic waste, with the group work on the GPU.

Mirrors what `Groth16::compile` -> ark-groth16 `generate_parameters` produces
(manta-crypto/src/arkworks/groth16.rs:571-586; key conventions readable in-repo at
manta-trusted-setup/src/groth16/mpc.rs:251-431): for a circuit with matrices A, B, C over the domain of
size D = next_pow2(m + P),

    a_j = sum_i A[i][j] L_i(tau) (+ L_{m+j}(tau) for j < P),  b_j, c_j likewise
    a_query[j] = a_j G1,  b_g1_query[j] = b_j G1,  b_g2_query[j] = b_j G2
    l_query[j-P] = ((beta a_j + alpha b_j + c_j)/delta) G1  (j >= P),   gamma_abc likewise with gamma (j < P)
    h_query[i] = (tau^i (tau^D - 1)/delta) G1,  i < D-1

Scalars are computed with Python integers on the host; every scalar multiplication runs on the GPU
through `mg_fixed_base_mul` (key generation is a "next" row -- this is the part of it
the bench and tests need to obtain *valid* proving keys of the real manta-pay shapes).
Used by bench.py (prove workload) and tests; never touches oracle/.
"""
from __future__ import annotations

import numpy as np

from . import api, synth

G1_GEN = {
    synth.BN254: (1, 2),
    synth.BLS12_381: (
        0x17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb,
        0x08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1),
}
G2_GEN = {
    synth.BN254: (10857046999023057135944570762232829481370756359578518086990519993285655852781,
                  11559732032986387107991004021392285783925812861821192530917403151452391805634,
                  8495653923123431417604973247489272438418190587263600148770280649306958101930,
                  4082367875863433681332203403145435568316851327593401208105741076214120093531),
    synth.BLS12_381: (
        0x024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8,
        0x13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e,
        0x0ce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c923ac9cc3baca289e193548608b82801,
        0x0606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab3f370d275cec1da1aaa9075ff05f79be),
}
FR_GEN = {synth.BN254: 5, synth.BLS12_381: 7}
FR_TWO_ADICITY = {synth.BN254: 28, synth.BLS12_381: 32}


def generator(curve, group):
    q, nl = synth.FQ_MODULUS[curve], synth.FQ_LIMBS[curve]
    coords = G1_GEN[curve] if group == 1 else G2_GEN[curve]
    return synth.to_mont(list(coords), q, nl).reshape(-1)


class ProvingKey:
    """Host arrays in the C ABI's memory format (affine Montgomery limbs, infinity = zeros)."""
    pass


def _batch_inverse(vals, p):
    pref = [1] * (len(vals) + 1)
    for i, v in enumerate(vals):
        pref[i + 1] = pref[i] * v % p
    inv = pow(pref[-1], -1, p)
    out = [0] * len(vals)
    for i in range(len(vals) - 1, -1, -1):
        out[i] = inv * pref[i] % p
        inv = inv * vals[i] % p
    return out


def _csr_cols(c, M):
    r = synth.FR_MODULUS[c.curve]
    Rinv = pow(1 << 256, -1, r)
    # distinct coefficient values are few: decode through a cache keyed by the raw limbs
    cache = {}
    vals = []
    raw = np.ascontiguousarray(M.val).view(np.uint64).reshape(-1, 4)
    for row in raw:
        k = row.tobytes()
        v = cache.get(k)
        if v is None:
            v = int.from_bytes(k, "little") * Rinv % r
            cache[k] = v
        vals.append(v)
    return vals


def generate(c: synth.Circuit, toxic):
    """toxic = (tau, alpha, beta, gamma, delta) as Python ints. Returns a ProvingKey whose fields are what
    `api.ProvingContext` consumes, plus gamma_g2 / gamma_abc_g1 for verification."""
    curve = c.curve
    r = synth.FR_MODULUS[curve]
    tau, alpha, beta, gamma, delta = [int(t) % r for t in toxic]
    D, m, P, V = c.D, c.m, c.P, c.V
    lg = D.bit_length() - 1
    w = pow(pow(FR_GEN[curve], (r - 1) >> FR_TWO_ADICITY[curve], r), 1 << (FR_TWO_ADICITY[curve] - lg), r)
    Zt = (pow(tau, D, r) - 1) % r
    # Lagrange coefficients L_i(tau) = Z(tau)/D * w^i / (tau - w^i)
    pw = [1] * D
    for i in range(1, D):
        pw[i] = pw[i - 1] * w % r
    den = _batch_inverse([(tau - x) % r for x in pw], r)
    zd = Zt * pow(D, -1, r) % r
    L = [zd * x % r * d % r for x, d in zip(pw, den)]
    a = [0] * V
    b = [0] * V
    cc = [0] * V
    for M, acc in ((c.A, a), (c.B, b), (c.C, cc)):
        vals = _csr_cols(c, M)
        rp, col = M.row_ptr, M.col
        for i in range(m):
            Li = L[i]
            for k in range(rp[i], rp[i + 1]):
                j = col[k]
                acc[j] = (acc[j] + vals[k] * Li) % r
    for j in range(P):
        a[j] = (a[j] + L[m + j]) % r
    ginv, dinv = pow(gamma, -1, r), pow(delta, -1, r)
    ext = [(beta * a[j] + alpha * b[j] + cc[j]) % r for j in range(V)]
    gabc = [ext[j] * ginv % r for j in range(P)]
    lq = [ext[j] * dinv % r for j in range(P, V)]
    hz = Zt * dinv % r
    hq = [0] * (D - 1)
    cur = hz
    for i in range(D - 1):
        hq[i] = cur
        cur = cur * tau % r
    fixed1 = [alpha, beta, delta]
    fixed2 = [beta, gamma, delta]
    # one batched fixed-base multiply per group
    s1 = fixed1 + gabc + a + b + hq + lq
    s2 = fixed2 + b
    G1, G2 = generator(curve, 1), generator(curve, 2)
    w1, w2 = api.affine_limbs(curve, 1), api.affine_limbs(curve, 2)
    d1 = api.fixed_base_mul(curve, 1, G1, api.DeviceBuffer.from_numpy(synth.ints_to_limbs(s1, 4)), len(s1))
    p1 = d1.to_numpy(shape=(len(s1), w1))
    d2 = api.fixed_base_mul(curve, 2, G2, api.DeviceBuffer.from_numpy(synth.ints_to_limbs(s2, 4)), len(s2))
    p2 = d2.to_numpy(shape=(len(s2), w2))
    pk = ProvingKey()
    pk.curve, pk.V, pk.P, pk.D, pk.h_len = curve, V, P, D, D - 1
    o = 0
    pk.alpha_g1, pk.beta_g1, pk.delta_g1 = p1[0:1].copy(), p1[1:2].copy(), p1[2:3].copy()
    o = 3
    pk.gamma_abc_g1 = p1[o:o + P].copy(); o += P
    pk.a_query = p1[o:o + V].copy(); o += V
    pk.b_g1_query = p1[o:o + V].copy(); o += V
    pk.h_query = p1[o:o + D - 1].copy(); o += D - 1
    pk.l_query = p1[o:o + V - P].copy(); o += V - P
    pk.beta_g2, pk.gamma_g2, pk.delta_g2 = p2[0:1].copy(), p2[1:2].copy(), p2[2:3].copy()
    pk.b_g2_query = p2[3:3 + V].copy()
    return pk
