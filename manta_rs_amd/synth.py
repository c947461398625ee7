"""Synthetic, shape-exact R1CS circuits and witnesses for the Groth16 hot path.

The reference builds its circuits with `Transfer::known_constraints`
(manta-accounting/src/transfer/mod.rs:667-673) and hands the prover an `R1CS<F>` carrying the
sparse matrices A, B, C plus the full assignment z = instance || witness
(manta-crypto/src/arkworks/constraint/mod.rs:94-135,199-217). No witness can be captured offline
(no Rust toolchain), so benches and tests use deterministic synthetic circuits with
the exact (D, V, P) of the real manta-pay shapes:

    ToPrivate        D = 2^14, V =  8 253, P = 13
    ToPublic         D = 2^15, V = 27 945, P = 19
    PrivateTransfer  D = 2^16, V = 35 175, P = 27

Rows are a satisfiable mix of multiplication gates, boolean gates (b*(1-b)=0, giving the 0/1-heavy
witness real circuits have) and linear gates. Pure Python big-int arithmetic + numpy packing; no
dependency on the oracle.

Witness profiles (`profile=` of make_circuit / make_shape; the density of z decides how much work four of a proof's
five MSMs have -- arkworks and this library both skip zero scalars):

    "sparse"  the round-1..3 generator: gate kinds drawn at random; multiplication gates over earlier variables
              cascade zeros, so the RESULTING z is about 69 % zeros / 23 % ones / 8 % anything else -- a lower bound
              on a proof's MSM work, kept for comparison
    "W"       SURVEY.md 8(d) config 2's witness-like distribution enforced on the RESULTING z: 40 % zeros, 25 % ones,
              10 % below 2^64, 25 % uniform field elements -- structurally: boolean gates (the zeros and ones),
              bit-packing style linear gates over booleans with small coefficients (the small values) and
              multiplication / linear gates over dense operands only (no zero can enter them), so every further
              assignment of the same matrices (Reassigner) keeps the distribution
    "dense"   SURVEY.md 8(d) config 1: a satisfiable multiplication chain over non-zero random values; 0 % trivial
              scalars (z_0 = 1 aside) -- the upper bound on a proof's MSM work

`histogram(z)` measures what was actually produced; bench.py prints it next to every proofs/s figure.
"""
from __future__ import annotations

import dataclasses
import numpy as np

BN254 = 0
BLS12_381 = 1

FR_MODULUS = {
    BN254: 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001,
    BLS12_381: 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001,
}
FQ_MODULUS = {
    BN254: 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47,
    BLS12_381: 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab,
}
FR_BITS = {BN254: 254, BLS12_381: 255}
FQ_LIMBS = {BN254: 4, BLS12_381: 6}

SHAPES = {
    "to_private": (1 << 14, 8253, 13),
    "to_public": (1 << 15, 27945, 19),
    "private_transfer": (1 << 16, 35175, 27),
}


class XorShift:
    """xoshiro256** -- the deterministic generator SURVEY.md section 8(d) names for synthetic inputs."""

    def __init__(self, seed: int):
        # splitmix64 seeding
        s = []
        x = seed & 0xFFFFFFFFFFFFFFFF
        for _ in range(4):
            x = (x + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
            z = x
            z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
            z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
            s.append(z ^ (z >> 31))
        self.s = s

    def next(self) -> int:
        s = self.s
        M = 0xFFFFFFFFFFFFFFFF
        r = (s[1] * 5) & M
        r = (((r << 7) | (r >> 57)) & M) * 9 & M
        t = (s[1] << 17) & M
        s[2] ^= s[0]
        s[3] ^= s[1]
        s[1] ^= s[2]
        s[0] ^= s[3]
        s[2] ^= t
        s[3] = ((s[3] << 45) | (s[3] >> 19)) & M
        return r

    def below(self, n: int) -> int:
        return self.next() % n

    def field(self, p: int) -> int:
        v = 0
        for _ in range((p.bit_length() + 63) // 64 + 1):
            v = (v << 64) | self.next()
        return v % p


# ------------------------------------------------------------------ int <-> limb packing
def ints_to_limbs(vals, nlimbs: int) -> np.ndarray:
    """list of python ints -> (len, nlimbs) uint64 little-endian limbs."""
    nb = 8 * nlimbs
    buf = b"".join(int(v).to_bytes(nb, "little") for v in vals)
    return np.frombuffer(buf, dtype=np.uint64).reshape(len(vals), nlimbs).copy()


def limbs_to_ints(arr: np.ndarray):
    arr = np.ascontiguousarray(arr, dtype=np.uint64)
    n = arr.shape[-1]
    flat = arr.reshape(-1, n)
    raw = flat.tobytes()
    nb = 8 * n
    return [int.from_bytes(raw[i * nb:(i + 1) * nb], "little") for i in range(flat.shape[0])]


def to_mont(vals, p: int, nlimbs: int) -> np.ndarray:
    R = (1 << (64 * nlimbs)) % p
    return ints_to_limbs([(v * R) % p for v in vals], nlimbs)


def from_mont(arr: np.ndarray, p: int):
    n = arr.shape[-1]
    Rinv = pow(1 << (64 * n), -1, p)
    return [(v * Rinv) % p for v in limbs_to_ints(arr)]


@dataclasses.dataclass
class CSR:
    row_ptr: np.ndarray  # uint32 [m+1]
    col: np.ndarray      # uint32 [nnz]
    val: np.ndarray      # uint64 [nnz, 4] Montgomery Fr


@dataclasses.dataclass
class Circuit:
    curve: int
    m: int          # constraints
    P: int          # instance variables incl. the constant 1
    V: int          # all variables
    D: int          # FFT domain = next_pow2(m + P)
    A: CSR
    B: CSR
    C: CSR
    z_int: list     # full assignment as python ints (canonical)
    z: np.ndarray   # uint64 [V, 4] Montgomery
    profile: str = "sparse"  # witness profile the circuit was generated for (module docstring)


def _pack_csr(rows, p, coeff_cache):
    row_ptr = np.zeros(len(rows) + 1, dtype=np.uint32)
    cols = []
    vals = []
    k = 0
    for i, r in enumerate(rows):
        for (c, v) in r:
            cols.append(c)
            vals.append(v % p)
            k += 1
        row_ptr[i + 1] = k
    # montgomery-pack distinct coefficients once
    R = (1 << 256) % p
    uniq = {}
    idx = np.empty(len(vals), dtype=np.int64)
    for j, v in enumerate(vals):
        if v not in uniq:
            uniq[v] = len(uniq)
        idx[j] = uniq[v]
    table = ints_to_limbs([(v * R) % p for v in uniq.keys()], 4) if uniq else np.zeros((0, 4), dtype=np.uint64)
    val = table[idx] if len(vals) else np.zeros((0, 4), dtype=np.uint64)
    return CSR(row_ptr, np.asarray(cols, dtype=np.uint32), np.ascontiguousarray(val))


PROFILES = ("sparse", "W", "dense")
# the W profile's target shares of the resulting z: zeros, ones, below 2^64, uniform
W_SHARES = (0.40, 0.25, 0.10, 0.25)


def histogram(z_int) -> dict:
    """Shares of an assignment that are 0 / 1 / another value below 2^64 / anything else (the four classes of SURVEY.md
    8(d)'s W distribution; the number `rust/capture` writes next to a captured witness)."""
    n = len(z_int)
    zero = sum(1 for v in z_int if v == 0)
    one = sum(1 for v in z_int if v == 1)
    small = sum(1 for v in z_int if 1 < v < (1 << 64))
    return {"n": n, "zero": zero / n, "one": one / n, "small": small / n, "dense": (n - zero - one - small) / n}


def _finish_circuit(curve, m, V, P, A, B, C, z, profile):
    p = FR_MODULUS[curve]
    D = 1
    while D < m + P:
        D <<= 1
    cache = {}
    return Circuit(curve, m, P, V, D, _pack_csr(A, p, cache), _pack_csr(B, p, cache), _pack_csr(C, p, cache), z,
                   to_mont(z, p, 4), profile)


def _nonzero_field(rng, p):
    v = rng.field(p)
    return v if v else 1


def _extra_rows(rng, p, A, B, C, m, nw, V, bools, coef):
    """rows beyond the defining ones (m > V - P): identities over existing variables, satisfied by every assignment"""
    for k in range(nw, m):
        if bools and rng.below(2):
            b = bools[rng.below(len(bools))]
            A.append([(b, 1)])
            B.append([(b, 1)])
            C.append([(b, 1)])
        else:
            l, r = rng.below(V), rng.below(V)
            c = coef()
            row = [(l, 1), (r, c)] if l != r else [(l, (1 + c) % p)]
            A.append(row)
            B.append([(0, 1)])
            C.append(list(row))


def _make_dense(curve, m, V, P, seed):
    """z[P + i] = (ca z[l]) (cb z[r]) over non-zero values -- a multiplication chain; every scalar of
    every witness MSM is a full-width field element."""
    p = FR_MODULUS[curve]
    rng = XorShift(seed)
    z = [0] * V
    z[0] = 1
    for j in range(1, P):
        z[j] = _nonzero_field(rng, p)
    A, B, C = [], [], []
    nw = V - P
    coeffs = [_nonzero_field(rng, p) for _ in range(8)]

    def coef():
        return 1 if rng.below(4) else coeffs[rng.below(8)]

    for k in range(min(nw, m)):
        v = P + k
        # operands: the previous variable (the chain) and any earlier one; never z_0 on both sides (the product would be a
        # bare coefficient -- still dense, but keep the chain a chain)
        l = v - 1 if v - 1 >= 1 else 0
        r = 1 + rng.below(v - 1) if v > 1 else 0
        ca, cb = coef(), coef()
        z[v] = (ca * z[l] % p) * (cb * z[r] % p) % p
        assert z[v] != 0
        A.append([(l, ca)])
        B.append([(r, cb)])
        C.append([(v, 1)])
    _extra_rows(rng, p, A, B, C, m, nw, V, [], coef)
    for k in range(m, nw):
        z[P + k] = _nonzero_field(rng, p)
    return _finish_circuit(curve, m, V, P, A, B, C, z, "dense")


def _w_plan(V, P, nw_defined):
    """kinds of the defined witness rows so that the RESULTING z (instance variables and z_0 included) meets W_SHARES
    exactly: -> (#boolean zeros, #boolean ones, #small, #dense) among the nw_defined rows"""
    want0 = round(W_SHARES[0] * V)
    want1 = round(W_SHARES[1] * V) - 1  # z_0 = 1 is one of the ones
    wants = round(W_SHARES[2] * V)
    free = V - P - nw_defined           # undefined trailing variables (nw > m): filled dense
    wantd = nw_defined - want0 - want1 - wants
    assert want0 >= 0 and want1 >= 0 and wants >= 0 and wantd >= 0, "shape too small for the W shares"
    _ = free
    return want0, want1, wants, wantd


def _make_w(curve, m, V, P, seed):
    """SURVEY.md 8(d) config 2's W distribution on the resulting assignment; see the module docstring."""
    p = FR_MODULUS[curve]
    rng = XorShift(seed)
    z = [0] * V
    z[0] = 1
    for j in range(1, P):
        z[j] = _nonzero_field(rng, p)
    A, B, C = [], [], []
    nw = V - P
    nd = min(nw, m)
    coeffs = [_nonzero_field(rng, p) for _ in range(8)]

    def coef():
        return 1 if rng.below(4) else coeffs[rng.below(8)]

    n0, n1, ns, ndense = _w_plan(V, P, nd)
    # a random interleaving of the four kinds with exact counts (Fisher-Yates on the multiset)
    kinds = [0] * n0 + [1] * n1 + [2] * ns + [3] * ndense
    for i in range(len(kinds) - 1, 0, -1):
        j = rng.below(i + 1)
        kinds[i], kinds[j] = kinds[j], kinds[i]
    # a small-value gate needs booleans before it and a dense gate prefers dense operands beyond the instance: make sure the
    # first rows supply both kinds (swap them to the front; the multiset is unchanged)
    for want_kind, pos in ((0, 0), (1, 1), (3, 2)):
        if pos < len(kinds) and kinds[pos] != want_kind:
            for j in range(pos + 1, len(kinds)):
                if kinds[j] == want_kind:
                    kinds[pos], kinds[j] = kinds[j], kinds[pos]
                    break
    bools, dense = [], list(range(1, P))  # variables known to be boolean / known to be non-zero field elements
    for k in range(nd):
        v = P + k
        kind = kinds[k]
        if kind <= 1:  # boolean witness: v (1 - v) = 0, value fixed by the plan
            z[v] = kind
            A.append([(v, 1)])
            B.append([(0, 1), (v, p - 1)])
            C.append([])
            bools.append(v)
        elif kind == 2 and bools:  # bit-packing style: z[v] = c0 + sum c_i b_i with small c: 1 < value < 2^64 whatever the bits
            nt = 1 + rng.below(3)
            terms = [(0, 2 + rng.below(1 << 60))]
            for _ in range(nt):
                terms.append((bools[rng.below(len(bools))], 1 + rng.below(1 << 60)))
            row = {}
            for i, cf in terms:
                row[i] = row.get(i, 0) + cf
            z[v] = sum(cf * z[i] for i, cf in row.items()) % p
            assert 1 < z[v] < (1 << 64)
            A.append(sorted(row.items()))
            B.append([(0, 1)])
            C.append([(v, 1)])
        else:  # dense: product (or sum) of dense operands -- no zero can enter
            if not dense:
                dense.append(0)  # P = 1: only the constant is available
            l, r = dense[rng.below(len(dense))], dense[rng.below(len(dense))]
            ca, cb = coef(), coef()
            if rng.below(4):
                z[v] = (ca * z[l] % p) * (cb * z[r] % p) % p
                A.append([(l, ca)])
                B.append([(r, cb)])
            else:  # linear gate; a sum of two dense values is zero with probability 1/p
                z[v] = (ca * z[l] + cb * z[r]) % p
                A.append([(l, ca), (r, cb)] if l != r else [(l, (ca + cb) % p)])
                B.append([(0, 1)])
            if z[v] == 0 or z[v] == 1 or z[v] < (1 << 64):  # (probability ~2^-190; keep the classes exact)
                z[v] = (ca * z[l] % p) * (cb * z[r] % p) % p
                A[-1], B[-1] = [(l, ca)], [(r, cb)]
            C.append([(v, 1)])
            dense.append(v)
    _extra_rows(rng, p, A, B, C, m, nw, V, bools, coef)
    for k in range(m, nw):
        z[P + k] = _nonzero_field(rng, p)
    return _finish_circuit(curve, m, V, P, A, B, C, z, "W")


def make_circuit(curve: int, m: int, V: int, P: int, seed: int = 0x4D414E5441_0001, profile: str = "sparse") -> Circuit:
    """Satisfiable synthetic R1CS with m rows, V variables, P instance variables (z_0 = 1); `profile`: module docstring."""
    assert V > P >= 1 and m >= 1
    if profile == "dense":
        return _make_dense(curve, m, V, P, seed)
    if profile == "W":
        return _make_w(curve, m, V, P, seed)
    if profile != "sparse":
        raise ValueError("profile is one of %r" % (PROFILES,))
    p = FR_MODULUS[curve]
    rng = XorShift(seed)
    z = [0] * V
    z[0] = 1
    for j in range(1, P):
        z[j] = rng.field(p)
    A, B, C = [], [], []
    nw = V - P
    coeffs = [rng.field(p) for _ in range(8)]

    def coef():
        return 1 if rng.below(4) else coeffs[rng.below(8)]

    bools = []
    for k in range(min(nw, m)):
        v = P + k
        kind = rng.below(100)
        if v <= 1 or kind < 45:  # boolean witness
            z[v] = rng.below(2) if rng.below(5) else 0
            A.append([(v, 1)])
            B.append([(0, 1), (v, p - 1)])
            C.append([])
            bools.append(v)
        elif kind < 85:  # multiplication gate
            l, r = rng.below(v), rng.below(v)
            ca, cb = coef(), coef()
            z[v] = (ca * z[l] % p) * (cb * z[r] % p) % p
            A.append([(l, ca)])
            B.append([(r, cb)])
            C.append([(v, 1)])
        else:  # linear gate
            l, r = rng.below(v), rng.below(v)
            c = coef()
            z[v] = (z[l] + c * z[r]) % p
            A.append([(l, 1), (r, c)] if l != r else [(l, (1 + c) % p)])
            B.append([(0, 1)])
            C.append([(v, 1)])
    for k in range(nw, m):  # extra rows over existing variables
        if bools and rng.below(2):
            b = bools[rng.below(len(bools))]
            A.append([(b, 1)])
            B.append([(b, 1)])
            C.append([(b, 1)])
        else:
            l, r = rng.below(V), rng.below(V)
            c = coef()
            row = [(l, 1), (r, c)] if l != r else [(l, (1 + c) % p)]
            A.append(row)
            B.append([(0, 1)])
            C.append(list(row))
    # variables without a defining row (nw > m): free random values
    for k in range(m, nw):
        z[P + k] = rng.field(p)
    D = 1
    while D < m + P:
        D <<= 1
    cache = {}
    return Circuit(curve, m, P, V, D, _pack_csr(A, p, cache), _pack_csr(B, p, cache), _pack_csr(C, p, cache), z,
                   to_mont(z, p, 4))


class Reassigner:
    """Further satisfying assignments of one circuit (same matrices): fresh instance values, fresh boolean
    witnesses, every gate output recomputed from its defining row -- distinct proofs of one ProvingContext. The
    matrices are parsed once, so a batch of hundreds of assignments of a PrivateTransfer-sized circuit costs
    ~0.1 s each."""

    def __init__(self, c: Circuit):
        self.c = c
        p = FR_MODULUS[c.curve]
        Rinv = pow(1 << 256, -1, p)

        def rows(M):
            vals = [v * Rinv % p for v in limbs_to_ints(M.val)] if len(M.col) else []
            col = M.col.tolist()
            rp = M.row_ptr.tolist()
            return [[(col[k], vals[k]) for k in range(rp[i], rp[i + 1])] for i in range(c.m)]

        self.A, self.B, self.C = rows(c.A), rows(c.B), rows(c.C)

    def assign(self, seed: int) -> Circuit:
        c, A, B, C = self.c, self.A, self.B, self.C
        p = FR_MODULUS[c.curve]
        rng = XorShift(seed)
        z = [0] * c.V
        z[0] = 1
        for j in range(1, c.P):
            z[j] = _nonzero_field(rng, p)
        nw = c.V - c.P
        for k in range(min(nw, c.m)):
            v = c.P + k
            if not C[k]:  # boolean witness: v * (1 - v) = 0 (W profile: ones with the share that keeps 40 % / 25 %)
                z[v] = (1 if rng.below(65) >= 40 else 0) if c.profile == "W" else rng.below(2)
            else:         # gate with output v: (sum A)(sum B) = z[v]
                sa = sum(cf * z[i] for i, cf in A[k]) % p
                sb = sum(cf * z[i] for i, cf in B[k]) % p
                assert C[k] == [(v, 1)]
                z[v] = sa * sb % p
        for k in range(c.m, nw):
            z[c.P + k] = rng.field(p)
        return dataclasses.replace(c, z_int=z, z=to_mont(z, p, 4))


def reassign(c: Circuit, seed: int) -> Circuit:
    """One more satisfying assignment of the same circuit (see Reassigner)."""
    return Reassigner(c).assign(seed)


def make_shape(curve: int, name: str, seed: int = 0x4D414E5441_0001, profile: str = "sparse") -> Circuit:
    D, V, P = SHAPES[name]
    return make_circuit(curve, D - P, V, P, seed, profile)


def check_satisfied(c: Circuit) -> bool:
    p = FR_MODULUS[c.curve]
    Rinv = pow(1 << 256, -1, p)

    def rows(M):
        vals = [v * Rinv % p for v in limbs_to_ints(M.val)] if len(M.col) else []
        out = []
        for i in range(c.m):
            s = 0
            for k in range(M.row_ptr[i], M.row_ptr[i + 1]):
                s += vals[k] * c.z_int[M.col[k]]
            out.append(s % p)
        return out

    a, b, cc = rows(c.A), rows(c.B), rows(c.C)
    return all((x * y - w) % p == 0 for x, y, w in zip(a, b, cc))


def msm_scalars(curve: int, n: int, dist: str, seed: int = 0x4D414E5441_0003) -> np.ndarray:
    """Canonical (non-Montgomery) scalars, uint64 [n,4]. dist: 'U' uniform, 'W' witness-like
    (40% 0, 25% 1, 10% < 2^64, 25% uniform) -- SURVEY.md section 8(d) config 2."""
    p = FR_MODULUS[curve]
    rs = np.random.RandomState(seed & 0x7FFFFFFF)
    raw = rs.randint(0, 1 << 63, size=(n, 4), dtype=np.int64).astype(np.uint64)
    raw ^= rs.randint(0, 1 << 63, size=(n, 4), dtype=np.int64).astype(np.uint64) << np.uint64(1)
    # clear top bits so that every value < 2^(bits-1) < p  (uniform enough for a throughput test;
    # exact values are irrelevant, parity is checked on whatever was generated)
    raw[:, 3] &= np.uint64((1 << (FR_BITS[curve] - 1 - 192)) - 1)
    if dist == "U":
        return raw
    assert dist == "W"
    sel = rs.randint(0, 100, size=n)
    out = raw.copy()
    out[sel < 40] = 0
    one = (sel >= 40) & (sel < 65)
    out[one] = 0
    out[one, 0] = 1
    small = (sel >= 65) & (sel < 75)
    out[small, 1:] = 0
    return out
