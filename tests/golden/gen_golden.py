#!/usr/bin/env python3
"""Independent pure-Python big-integer restatement of the hot path, used ONLY to generate the small
golden vectors in tests/golden/vectors.json (committed). It shares no code with oracle/ (C) or with the
HIP product: affine curve formulas with modular inverses, O(n^2) DFTs, naive double-and-add MSM,
textbook Groth16 equations (SURVEY.md App. B.1 / mpc.rs:251-431 conventions).

    python tests/golden/gen_golden.py        # rewrites tests/golden/vectors.json

The reference itself holds no MSM/NTT/proof vectors, and its arithmetic crates cannot be
built here (F2); these vectors pin the C oracle against a second, independent derivation. The oracle is
additionally pinned by the reference's own verifying-key fixtures (tests/golden/*.dat).
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from manta_rs_amd import synth  # circuit generator + xoshiro (host-side product helper, no field code shared)

CURVES = {
    0: dict(name="bn254",
            q=0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47,
            r=0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001,
            b=3, g1=(1, 2), gen=5, s=28,
            g2=((10857046999023057135944570762232829481370756359578518086990519993285655852781,
                 11559732032986387107991004021392285783925812861821192530917403151452391805634),
                (8495653923123431417604973247489272438418190587263600148770280649306958101930,
                 4082367875863433681332203403145435568316851327593401208105741076214120093531)),
            qbytes=32),
    1: dict(name="bls12_381",
            q=0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab,
            r=0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001,
            b=4, g1=(0x17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb,
                     0x08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1),
            gen=7, s=32,
            g2=((0x024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8,
                 0x13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e),
                (0x0ce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c923ac9cc3baca289e193548608b82801,
                 0x0606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab3f370d275cec1da1aaa9075ff05f79be)),
            qbytes=48),
}


# ---- generic affine group law over Fq (ints) or Fq2 (pairs), y^2 = x^3 + b, a = 0 ----
class Fq2:
    def __init__(self, q):
        self.q = q

    def add(s, a, b): return ((a[0] + b[0]) % s.q, (a[1] + b[1]) % s.q)
    def sub(s, a, b): return ((a[0] - b[0]) % s.q, (a[1] - b[1]) % s.q)
    def mul(s, a, b): return ((a[0] * b[0] - a[1] * b[1]) % s.q, (a[0] * b[1] + a[1] * b[0]) % s.q)
    def inv(s, a):
        d = pow(a[0] * a[0] + a[1] * a[1], -1, s.q)
        return (a[0] * d % s.q, -a[1] * d % s.q)
    def neg(s, a): return (-a[0] % s.q, -a[1] % s.q)
    def small(s, k): return (k % s.q, 0)
    def is_zero(s, a): return a[0] == 0 and a[1] == 0


class Fq1:
    def __init__(self, q):
        self.q = q

    def add(s, a, b): return (a + b) % s.q
    def sub(s, a, b): return (a - b) % s.q
    def mul(s, a, b): return a * b % s.q
    def inv(s, a): return pow(a, -1, s.q)
    def neg(s, a): return -a % s.q
    def small(s, k): return k % s.q
    def is_zero(s, a): return a == 0


def pt_add(F, P, Q):
    if P is None: return Q
    if Q is None: return P
    (x1, y1), (x2, y2) = P, Q
    if x1 == x2:
        if F.is_zero(F.add(y1, y2)): return None
        lam = F.mul(F.mul(F.small(3), F.mul(x1, x1)), F.inv(F.mul(F.small(2), y1)))
    else:
        lam = F.mul(F.sub(y2, y1), F.inv(F.sub(x2, x1)))
    x3 = F.sub(F.sub(F.mul(lam, lam), x1), x2)
    return (x3, F.sub(F.mul(lam, F.sub(x1, x3)), y1))


def pt_mul(F, P, k):
    R = None
    while k:
        if k & 1: R = pt_add(F, R, P)
        P = pt_add(F, P, P)
        k >>= 1
    return R


def pt_neg(F, P):
    return None if P is None else (P[0], F.neg(P[1]))


def msm(F, pts, ks):
    acc = None
    for P, k in zip(pts, ks):
        acc = pt_add(F, acc, pt_mul(F, P, k))
    return acc


# ---- arkworks canonical compressed serialisation ----
def ser_fq(v, nb): return v.to_bytes(nb, "little")


def ser_point(cur, P, g2):
    nb = cur["qbytes"]
    q = cur["q"]
    if P is None:
        out = bytearray(nb * (2 if g2 else 1))
        out[-1] |= 0x40
        return bytes(out)
    x, y = P
    if g2:
        out = bytearray(ser_fq(x[0], nb) + ser_fq(x[1], nb))
        ny = ((-y[0]) % q, (-y[1]) % q)
        high = (y[1] > ny[1]) if y[1] != ny[1] else (y[0] > ny[0])
    else:
        out = bytearray(ser_fq(x, nb))
        high = y > (-y) % q
    if high: out[-1] |= 0x80
    return bytes(out)


def hexpt(P, g2):
    if P is None: return None
    if g2: return [[hex(P[0][0]), hex(P[0][1])], [hex(P[1][0]), hex(P[1][1])]]
    return [hex(P[0]), hex(P[1])]


# ---- Fr: naive DFT domain ops with arkworks' omega / coset conventions (App. A.1, B.3) ----
def domain(cur, D):
    r = cur["r"]
    w = pow(cur["gen"], (r - 1) >> cur["s"], r)
    lg = D.bit_length() - 1
    return pow(w, 1 << (cur["s"] - lg), r)


def dft(cur, v, inverse=False, coset=False):
    r, D = cur["r"], len(v)
    w = domain(cur, D)
    g = cur["gen"]
    v = list(v)
    if not inverse:
        if coset: v = [x * pow(g, i, r) % r for i, x in enumerate(v)]
        return [sum(v[j] * pow(w, i * j, r) for j in range(D)) % r for i in range(D)]
    wi = pow(w, -1, r)
    out = [sum(v[j] * pow(wi, i * j, r) for j in range(D)) * pow(D, -1, r) % r for i in range(D)]
    if coset:
        gi = pow(g, -1, r)
        out = [x * pow(gi, i, r) % r for i, x in enumerate(out)]
    return out


def csr_rows(c, M):
    r = synth.FR_MODULUS[c.curve]
    Rinv = pow(1 << 256, -1, r)
    vals = [v * Rinv % r for v in synth.limbs_to_ints(M.val)] if len(M.col) else []
    return [[(int(M.col[k]), vals[k]) for k in range(M.row_ptr[i], M.row_ptr[i + 1])] for i in range(c.m)]


def groth16(cur, c, toxic, rr, ss):
    """Setup from toxic waste + prove; returns proof bytes (arkworks compressed A||B||C) and h."""
    r = cur["r"]
    F1, F2 = Fq1(cur["q"]), Fq2(cur["q"])
    G1, G2 = cur["g1"], cur["g2"]
    tau, alpha, beta, gamma, delta = toxic
    A, B, C = csr_rows(c, c.A), csr_rows(c, c.B), csr_rows(c, c.C)
    D, m, P, V, z = c.D, c.m, c.P, c.V, c.z_int
    w = domain(cur, D)
    Zt = (pow(tau, D, r) - 1) % r
    L = [Zt * pow(D, -1, r) % r * pow(w, i, r) % r * pow((tau - pow(w, i, r)) % r, -1, r) % r for i in range(D)]
    a = [0] * V; b = [0] * V; cc = [0] * V
    for i in range(m):
        for (j, v) in A[i]: a[j] = (a[j] + v * L[i]) % r
        for (j, v) in B[i]: b[j] = (b[j] + v * L[i]) % r
        for (j, v) in C[i]: cc[j] = (cc[j] + v * L[i]) % r
    for j in range(P): a[j] = (a[j] + L[m + j]) % r
    dinv = pow(delta, -1, r)
    a_q = [pt_mul(F1, G1, x) for x in a]
    b1_q = [pt_mul(F1, G1, x) for x in b]
    b2_q = [pt_mul(F2, G2, x) for x in b]
    l_q = [pt_mul(F1, G1, (beta * a[j] + alpha * b[j] + cc[j]) * dinv % r) for j in range(P, V)]
    h_q = [pt_mul(F1, G1, pow(tau, i, r) * Zt % r * dinv % r) for i in range(D - 1)]
    alpha1, beta1, delta1 = pt_mul(F1, G1, alpha), pt_mul(F1, G1, beta), pt_mul(F1, G1, delta)
    beta2, delta2 = pt_mul(F2, G2, beta), pt_mul(F2, G2, delta)
    # witness map
    def rowdot(rows, i): return sum(v * z[j] for (j, v) in rows[i]) % r
    ea = [rowdot(A, i) for i in range(m)] + [z[j] for j in range(P)] + [0] * (D - m - P)
    eb = [rowdot(B, i) for i in range(m)] + [0] * (D - m)
    ec = [rowdot(C, i) for i in range(m)] + [0] * (D - m)
    ca, cb, c3 = dft(cur, ea, True), dft(cur, eb, True), dft(cur, ec, True)
    fa, fb, fc = dft(cur, ca, False, True), dft(cur, cb, False, True), dft(cur, c3, False, True)
    zinv = pow((pow(cur["gen"], D, r) - 1) % r, -1, r)
    hv = dft(cur, [(x * y - w_) * zinv % r for x, y, w_ in zip(fa, fb, fc)], True, True)
    assert hv[D - 1] == 0
    h_acc = msm(F1, h_q, hv[:D - 1])
    l_acc = msm(F1, l_q, z[P:])
    g_a = pt_add(F1, pt_add(F1, pt_add(F1, pt_mul(F1, delta1, rr), a_q[0]), msm(F1, a_q[1:], z[1:])), alpha1)
    g1_b = pt_add(F1, pt_add(F1, pt_add(F1, pt_mul(F1, delta1, ss), b1_q[0]), msm(F1, b1_q[1:], z[1:])), beta1)
    g2_b = pt_add(F2, pt_add(F2, pt_add(F2, pt_mul(F2, delta2, ss), b2_q[0]), msm(F2, b2_q[1:], z[1:])), beta2)
    g_c = pt_add(F1, pt_mul(F1, g_a, ss), pt_mul(F1, g1_b, rr))
    g_c = pt_add(F1, g_c, pt_neg(F1, pt_mul(F1, delta1, rr * ss % r)))
    g_c = pt_add(F1, pt_add(F1, g_c, l_acc), h_acc)
    proof = ser_point(cur, g_a, False) + ser_point(cur, g2_b, True) + ser_point(cur, g_c, False)
    return proof, hv


def main():
    out = {}
    for cid, cur in CURVES.items():
        r, q = cur["r"], cur["q"]
        F1, F2 = Fq1(q), Fq2(q)
        rng = synth.XorShift(0xC0FFEE + cid)
        v = {}
        # field KATs (canonical integers)
        fa, fb = [rng.field(q) for _ in range(4)], [rng.field(q) for _ in range(4)]
        v["fq_mul"] = [[hex(a), hex(b), hex(a * b % q)] for a, b in zip(fa, fb)]
        ra, rb = [rng.field(r) for _ in range(4)], [rng.field(r) for _ in range(4)]
        v["fr_mul"] = [[hex(a), hex(b), hex(a * b % r)] for a, b in zip(ra, rb)]
        v["fr_inv"] = [[hex(a), hex(pow(a, -1, r))] for a in ra]
        # group KATs
        ks = [1, 2, 3, rng.field(r), r - 1]
        v["g1_mul"] = [[hex(k), hexpt(pt_mul(F1, cur["g1"], k), False)] for k in ks]
        v["g2_mul"] = [[hex(k), hexpt(pt_mul(F2, cur["g2"], k), True)] for k in ks]
        v["g1_ser"] = [[hex(k), ser_point(cur, pt_mul(F1, cur["g1"], k), False).hex()] for k in ks + [r]]
        v["g2_ser"] = [[hex(k), ser_point(cur, pt_mul(F2, cur["g2"], k), True).hex()] for k in ks + [r]]
        # MSM n = 12 with zero / one / repeated-base / cancelling terms
        bs = [rng.field(r) for _ in range(12)]
        bs[7] = bs[6]
        sc = [rng.field(r) for _ in range(12)]
        sc[0], sc[1], sc[2] = 0, 1, 1
        sc[7] = (r - sc[6]) % r
        v["msm_g1"] = dict(base_scalars=[hex(x) for x in bs], scalars=[hex(x) for x in sc],
                           result=hexpt(msm(F1, [pt_mul(F1, cur["g1"], x) for x in bs], sc), False))
        v["msm_g2"] = dict(base_scalars=[hex(x) for x in bs[:6]], scalars=[hex(x) for x in sc[:6]],
                           result=hexpt(msm(F2, [pt_mul(F2, cur["g2"], x) for x in bs[:6]], sc[:6]), True))
        # NTT n = 16, four variants
        x = [rng.field(r) for _ in range(16)]
        v["ntt"] = dict(input=[hex(t) for t in x],
                        fft=[hex(t) for t in dft(cur, x)], ifft=[hex(t) for t in dft(cur, x, True)],
                        coset_fft=[hex(t) for t in dft(cur, x, False, True)],
                        coset_ifft=[hex(t) for t in dft(cur, x, True, True)])
        # toy Groth16: 12 constraints, 10 variables, 3 instance -> D = 16
        c = synth.make_circuit(cid, 12, 10, 3, seed=0xABCD + cid)
        toxic = [rng.field(r) for _ in range(5)]
        rr, ss = rng.field(r), rng.field(r)
        proof, hv = groth16(cur, c, toxic, rr, ss)
        v["groth16"] = dict(m=12, V=10, P=3, seed=0xABCD + cid, toxic=[hex(t) for t in toxic], r=hex(rr), s=hex(ss),
                            h=[hex(t) for t in hv], proof=proof.hex())
        out[cur["name"]] = v
        print(cur["name"], "done", file=sys.stderr)
    with open(os.path.join(HERE, "vectors.json"), "w") as f:
        json.dump(out, f, indent=0)
    print("wrote", os.path.join(HERE, "vectors.json"))


if __name__ == "__main__":
    main()
