"""Bulk group operations of the trusted-setup ceremony on the GPU -- host mirror of
manta-trusted-setup's hot loops, every group operation through the C ABI (`mg_ec_elementwise`, `mg_group_ntt`):

    batch_mul_fixed_scalar      manta-trusted-setup/src/util.rs:440-445   every point times ONE scalar
    batch_mul_pointwise         util.rs:447-455                           point i times scalar i
    contribute                  groth16/mpc.rs:451-468                    l_query, h_query *= 1/delta; delta_g1, delta_g2 *= delta
    accumulator_update          groth16/kzg.rs:444-468                    powers of tau times tau^i (and alpha, beta)
    lagrange_basis / initialize groth16/mpc.rs:355-431                    group-domain IFFTs of the powers, then the QAP sums
    merge_pairs_affine          util.rs:314-332                           one random linear combination of two point vectors (2 MSMs)
    same_ratio                  manta-crypto/src/arkworks/pairing.rs:88-109   e(a0, b1) == e(a1, b0) as one pairing product (`mg_pairing_check`)
    check_transform             groth16/mpc.rs:487-508                    the verifier's consistency checks of one contribution
    power_pairs / check_powers  util.rs:333-346, groth16/kzg.rs:508-521   consecutive powers differ by one ratio (kzg verifier)

Scalars are Python integers (the ceremony's RNG stays with the caller); points are [n, limbs] uint64 affine Montgomery
arrays, infinity = zeros -- the C ABI's format. Not used by the prover; a ceremony is a one-off, which is why SURVEY.md
ranks it last.
"""
from __future__ import annotations

import numpy as np

from . import api, synth


def _scalar(curve, k):
    return synth.ints_to_limbs([int(k) % synth.FR_MODULUS[curve]], 4)


def batch_mul_fixed_scalar(curve, group, points, scalar) -> np.ndarray:
    """util.rs:440-445: `cfg_iter_mut!(points).for_each(|point| scalar_mul(point, scalar))`"""
    return api.ec_elementwise(curve, group, api.EC_MUL_FIXED, points, _scalar(curve, scalar))


def batch_mul_pointwise(curve, group, points, scalars) -> np.ndarray:
    """util.rs:447-455"""
    r = synth.FR_MODULUS[curve]
    return api.ec_elementwise(curve, group, api.EC_MUL, points, synth.ints_to_limbs([int(k) % r for k in scalars], 4))


def contribute(curve, pk, delta):
    """mpc.rs:451-468 `contribute` (the RatioProof is the caller's business): returns the key after a contribution with
    the private scalar `delta`. A copy: the input key is left untouched."""
    import copy
    r = synth.FR_MODULUS[curve]
    delta = int(delta) % r
    dinv = pow(delta, -1, r)
    out = copy.copy(pk)
    out.l_query = batch_mul_fixed_scalar(curve, 1, pk.l_query, dinv)
    out.h_query = batch_mul_fixed_scalar(curve, 1, pk.h_query, dinv)
    out.delta_g1 = batch_mul_fixed_scalar(curve, 1, np.asarray(pk.delta_g1).reshape(1, -1), delta)
    out.delta_g2 = batch_mul_fixed_scalar(curve, 2, np.asarray(pk.delta_g2).reshape(1, -1), delta)
    return out


def g1_neg(curve, point) -> np.ndarray:
    """-P for an affine G1 point in Montgomery limbs (negation is linear: the representative of -y is p - y R)."""
    q, nl = synth.FQ_MODULUS[curve], synth.FQ_LIMBS[curve]
    pt = np.asarray(point, dtype=np.uint64).reshape(-1).copy()
    if not pt.any():
        return pt
    y = synth.limbs_to_ints(pt[nl:].reshape(1, nl))[0]
    pt[nl:] = synth.ints_to_limbs([(q - y) % q], nl)[0]
    return pt


def merge_pairs_affine(curve, group, lhs, rhs, rho=None):
    """util.rs:314-332 `merge_pairs_affine`: (sum rho_i lhs_i, sum rho_i rhs_i) for one vector of random scalars -- two
    MSMs over the same scalars (the reference multiplies point by point and folds; the sums are the same group elements).
    rho: the scalars as integers (tests), else drawn from the OS like the reference's `OsRng`."""
    import secrets
    lhs, rhs = np.asarray(lhs, dtype=np.uint64), np.asarray(rhs, dtype=np.uint64)
    n = lhs.shape[0]
    if rhs.shape[0] != n or n == 0:
        raise ValueError("merge_pairs_affine: two non-empty vectors of the same length")
    r = synth.FR_MODULUS[curve]
    rho = [secrets.randbelow(r) for _ in range(n)] if rho is None else [int(k) % r for k in rho]
    sc = synth.ints_to_limbs(rho, 4)
    out = []
    for pts in (lhs, rhs):
        bases = api.Bases(curve, group, pts)
        out.append(api.VariableBaseMSM.multi_scalar_mul(bases, sc))
        bases.close()
    return out[0], out[1]


def same_ratio(curve, lhs, rhs) -> bool:
    """pairing.rs:101-109 `same_ratio((a0, a1), (b0, b1))`: e(a0, b1) == e(a1, b0), evaluated as the single product
    e(a0, b1) e(-a1, b0) == 1 (two Miller loops and one final exponentiation on the GPU)."""
    a0, a1 = lhs
    b0, b1 = rhs
    return api.pairing_check(curve, np.stack([np.asarray(a0, dtype=np.uint64).reshape(-1), g1_neg(curve, a1)]),
                             np.stack([np.asarray(b1, dtype=np.uint64).reshape(-1), np.asarray(b0, dtype=np.uint64).reshape(-1)]))


def same(curve, lhs, rhs) -> bool:
    """pairing.rs `PairingEngineExt::same((a, b), (c, d))`: e(a, b) == e(c, d), as e(a, b) e(-c, d) == 1."""
    (a, b), (c, d) = lhs, rhs
    return api.pairing_check(curve, np.stack([np.asarray(a, dtype=np.uint64).reshape(-1), g1_neg(curve, c)]),
                             np.stack([np.asarray(b, dtype=np.uint64).reshape(-1), np.asarray(d, dtype=np.uint64).reshape(-1)]))


def power_pairs(curve, group, points, rho=None):
    """util.rs:333-346 `power_pairs`: one random linear combination of all but the last and of all but the first point."""
    points = np.asarray(points, dtype=np.uint64)
    return merge_pairs_affine(curve, group, points[:-1], points[1:], rho)


def check_transform(curve, prev, nxt, ratio=None, rho=None) -> str:
    """mpc.rs:487-508, the consistency checks of `verify_transform` between two states of the key (the ratio proof's own
    verification -- a hash to the curve -- stays with the caller, who passes its (ratio_0, ratio_1) pair or None):
    returns "" if the contribution is consistent, else the name of the reference's error variant."""
    d2 = (np.asarray(prev.delta_g2).reshape(-1), np.asarray(nxt.delta_g2).reshape(-1))
    if ratio is not None and not same_ratio(curve, ratio, d2):
        return "InconsistentDeltaChange"
    if not same_ratio(curve, (np.asarray(prev.delta_g1).reshape(-1), np.asarray(nxt.delta_g1).reshape(-1)), d2):
        return "InconsistentDeltaChange"
    if not same_ratio(curve, merge_pairs_affine(curve, 1, nxt.h_query, prev.h_query, rho), d2):
        return "InconsistentHChange"
    if not same_ratio(curve, merge_pairs_affine(curve, 1, nxt.l_query, prev.l_query, rho), d2):
        return "InconsistentLChange"
    return ""


class Accumulator:
    """kzg::Accumulator (kzg.rs:425-440): tau_powers_g1[2D-1... here `n1`], tau_powers_g2[n2], alpha/beta tau powers in G1[n2],
    beta_g2."""

    def __init__(self, curve, tau_powers_g1, tau_powers_g2, alpha_tau_powers_g1, beta_tau_powers_g1, beta_g2):
        self.curve = curve
        self.tau_powers_g1, self.tau_powers_g2 = np.asarray(tau_powers_g1), np.asarray(tau_powers_g2)
        self.alpha_tau_powers_g1, self.beta_tau_powers_g1 = np.asarray(alpha_tau_powers_g1), np.asarray(beta_tau_powers_g1)
        self.beta_g2 = np.asarray(beta_g2).reshape(1, -1)

    def update(self, tau, alpha, beta):
        """kzg.rs:444-468 `Accumulator::update`: element i of every vector is multiplied by tau^i (times alpha / beta)."""
        curve, r = self.curve, synth.FR_MODULUS[self.curve]
        n1, n2 = self.tau_powers_g1.shape[0], self.tau_powers_g2.shape[0]
        tp = [1] * n1
        for i in range(1, n1):
            tp[i] = tp[i - 1] * tau % r
        self.tau_powers_g1 = batch_mul_pointwise(curve, 1, self.tau_powers_g1, tp)
        self.tau_powers_g2 = batch_mul_pointwise(curve, 2, self.tau_powers_g2, tp[:n2])
        self.alpha_tau_powers_g1 = batch_mul_pointwise(curve, 1, self.alpha_tau_powers_g1, [t * alpha % r for t in tp[:n2]])
        self.beta_tau_powers_g1 = batch_mul_pointwise(curve, 1, self.beta_tau_powers_g1, [t * beta % r for t in tp[:n2]])
        self.beta_g2 = batch_mul_fixed_scalar(curve, 2, self.beta_g2, beta)


    def check_powers(self) -> str:
        """kzg.rs:508-521, the power checks of the accumulator's `verify_transform`: every vector is a sequence of consecutive
        powers of one tau (merged by `power_pairs`, compared with the first two G2 / G1 powers). "" or the error variant."""
        c = self.curve
        lhs, rhs = power_pairs(c, 2, self.tau_powers_g2)
        if not same(c, (self.tau_powers_g1[0], rhs), (self.tau_powers_g1[1], lhs)):
            return "TauG1Powers"
        t2 = (self.tau_powers_g2[1], self.tau_powers_g2[0])
        for vec, err in ((self.tau_powers_g1, "TauG2Powers"), (self.alpha_tau_powers_g1, "AlphaG1Powers"),
                         (self.beta_tau_powers_g1, "BetaG1Powers")):
            lhs, rhs = power_pairs(c, 1, vec)
            if not same(c, (lhs, t2[0]), (rhs, t2[1])):
                return err
        return ""


def lagrange_basis(curve, group, powers, D) -> np.ndarray:
    """mpc.rs:378-381: `domain.ifft(&batch_into_projective(&powers.tau_powers_g1))` -- the first D powers of tau become
    L_i(tau) G, i < D."""
    return api.group_ntt(curve, group, np.ascontiguousarray(powers[:D]), inverse=True)


def initialize(acc: Accumulator, c: synth.Circuit):
    """mpc.rs:355-431 `initialize`: the phase-2 proving key of circuit `c` from a powers-of-tau accumulator, gamma = delta =
    1 (the generators). Lagrange bases by group IFFTs on the GPU; the sparse QAP sums (specialize_to_phase_2, :251-294) as
    element-wise scalar multiplications on the GPU followed by per-variable sums."""
    curve, D, m, P, V = acc.curve, c.D, c.m, c.P, c.V
    r = synth.FR_MODULUS[curve]
    w1, w2 = api.affine_limbs(curve, 1), api.affine_limbs(curve, 2)
    assert acc.tau_powers_g1.shape[0] >= 2 * D - 1 and acc.tau_powers_g2.shape[0] >= D
    # h_query[i] = tau^(i+D) G - tau^i G (:372-377); the MPC key carries D entries where ark-groth16's has D - 1
    h = api.ec_elementwise(curve, 1, api.EC_SUB_MIXED, acc.tau_powers_g1[D:2 * D - 1], acc.tau_powers_g1[:D - 1])
    tau1 = lagrange_basis(curve, 1, acc.tau_powers_g1, D)
    tau2 = lagrange_basis(curve, 2, acc.tau_powers_g2, D)
    al1 = lagrange_basis(curve, 1, acc.alpha_tau_powers_g1, D)
    be1 = lagrange_basis(curve, 1, acc.beta_tau_powers_g1, D)
    Rinv = pow(1 << 256, -1, r)

    def column_sums(M, basis, group, width):
        """out[j] = sum_i M[i][j] basis[i]: one GPU scalar multiplication per non-zero, then sums per variable"""
        out = np.zeros((V, width), dtype=np.uint64)
        nnz = len(M.col)
        if nnz == 0:
            return out
        rows = np.repeat(np.arange(m), np.diff(M.row_ptr.astype(np.int64)))
        coeff = [v * Rinv % r for v in synth.limbs_to_ints(M.val)]
        prod = api.ec_elementwise(curve, group, api.EC_MUL, basis[rows], synth.ints_to_limbs(coeff, 4))
        order = np.argsort(M.col, kind="stable")
        cols = M.col[order]
        bounds = np.flatnonzero(np.diff(cols)) + 1
        for seg in np.split(order, bounds):
            out[M.col[seg[0]]] = api.points_sum(curve, group, prod[seg])
        return out

    def padd(a, b, group):
        return api.ec_elementwise(curve, group, api.EC_ADD, a, b)

    a_g1 = column_sums(c.A, tau1, 1, w1)
    a_g1[:P] = padd(a_g1[:P], tau1[m:m + P], 1)          # add_dummy_constraints (:299-312)
    b_g1 = column_sums(c.B, tau1, 1, w1)
    b_g2 = column_sums(c.B, tau2, 2, w2)
    ext = padd(padd(column_sums(c.A, be1, 1, w1), column_sums(c.B, al1, 1, w1), 1), column_sums(c.C, tau1, 1, w1), 1)
    ext[:P] = padd(ext[:P], be1[m:m + P], 1)
    pk = api.ProvingKey()
    pk.curve, pk.V, pk.P, pk.D, pk.h_len = curve, V, P, D, D - 1
    from . import keygen
    pk.alpha_g1 = acc.alpha_tau_powers_g1[0:1].copy()
    pk.beta_g1 = acc.beta_tau_powers_g1[0:1].copy()
    pk.beta_g2 = acc.beta_g2.copy()
    pk.delta_g1 = keygen.generator(curve, 1).reshape(1, -1)
    pk.gamma_g2 = keygen.generator(curve, 2).reshape(1, -1)
    pk.delta_g2 = keygen.generator(curve, 2).reshape(1, -1)
    pk.gamma_abc_g1, pk.l_query = ext[:P].copy(), ext[P:].copy()
    pk.a_query, pk.b_g1_query, pk.b_g2_query, pk.h_query = a_g1, b_g1, b_g2, h
    return pk
