"""Toy Groth16 key generation from explicit toxic waste, with the group work on the GPU.

Mirrors what `Groth16::compile` -> ark-groth16 `generate_parameters` produces
(manta-crypto/src/arkworks/groth16.rs:571-586; key conventions readable in-repo at
manta-trusted-setup/src/groth16/mpc.rs:251-431): for a circuit with matrices A, B, C over the domain of
size D = next_pow2(m + P),

    a_j = sum_i A[i][j] L_i(tau) (+ L_{m+j}(tau) for j < P),  b_j, c_j likewise
    a_query[j] = a_j G1,  b_g1_query[j] = b_j G1,  b_g2_query[j] = b_j G2
    l_query[j-P] = ((beta a_j + alpha b_j + c_j)/delta) G1  (j >= P),   gamma_abc likewise with gamma (j < P)
    h_query[i] = (tau^i (tau^D - 1)/delta) G1,  i < D-1

Thin wrapper over `mg_groth16_setup` (csrc/setup.cpp): the O(D + nnz) scalar preparation runs on the host in
C++, every scalar multiplication on the GPU. Used by bench.py (prove workload) and
tests to obtain *valid* proving keys of the real manta-pay shapes; never touches oracle/.
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


ProvingKey = api.ProvingKey


def generate(c: synth.Circuit, toxic, g1_generator=None, g2_generator=None):
    """toxic = (tau, alpha, beta, gamma, delta) as Python ints (the order of the oracle's setup). Returns a
    ProvingKey whose fields are what `api.ProvingContext` consumes, plus gamma_g2 / gamma_abc_g1 for verification.
    The generators default to the curves' standard ones."""
    curve = c.curve
    r = synth.FR_MODULUS[curve]
    tau, alpha, beta, gamma, delta = [int(t) % r for t in toxic]
    tox = synth.to_mont([alpha, beta, gamma, delta, tau], r, 4)
    g1 = generator(curve, 1) if g1_generator is None else g1_generator
    g2 = generator(curve, 2) if g2_generator is None else g2_generator
    return api.groth16_setup(api.R1CS.from_circuit(c), c.V, tox, g1, g2)
