"""ctypes binding of the CPU oracle (oracle/libmanta_oracle.so). TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by the product."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_SO = os.path.join(_ORACLE_DIR, "libmanta_oracle.so")

BN254, BLS12_381 = 0, 1
FIELD_IDS = {"bn254_fr": 0, "bn254_fq": 1, "bls381_fr": 2, "bls381_fq": 3}


def build():
    subprocess.check_call(["make", "-s", "-C", _ORACLE_DIR])


def usable_cpus():
    """CPUs this process may actually run on: the affinity mask, cut down by a cgroup CPU quota if there is one, and to
    one thread per physical core when SMT siblings are visible. The all-cores CPU baseline uses this many threads --
    `hardware_concurrency` can be far more than a container is allowed to use, and oversubscribed OpenMP barriers
    are catastrophically slow."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = max(1, min(n, int(int(q) / int(per))))
    except (OSError, ValueError):
        pass
    try:
        sib = open("/sys/devices/system/cpu/cpu0/topology/thread_siblings_list").read().strip()
        smt = len(sib.replace("-", ",").split(","))
        if smt > 1 and n == (os.cpu_count() or n):
            n = max(1, n // smt)
    except OSError:
        pass
    return n


def _load():
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")  # idle OpenMP threads sleep instead of spinning
    if not os.path.exists(_SO):
        build()
    lib = ctypes.CDLL(_SO)
    lib.mo_init()
    lib.mo_time_msm.restype = ctypes.c_double
    return lib


LIB = _load()
_vp = ctypes.c_void_p


def _p(a):
    return None if a is None else a.ctypes.data_as(_vp)


class CSRStruct(ctypes.Structure):
    _fields_ = [("row_ptr", _vp), ("col", _vp), ("val", _vp)]


class PKStruct(ctypes.Structure):
    _fields_ = [("n_vars", ctypes.c_uint64), ("n_inputs", ctypes.c_uint64), ("domain", ctypes.c_uint64),
                ("h_len", ctypes.c_uint64)] + [(k, _vp) for k in (
                    "alpha_g1", "beta_g1", "delta_g1", "beta_g2", "gamma_g2", "delta_g2", "gamma_abc_g1", "a_query",
                    "b_g1_query", "b_g2_query", "h_query", "l_query")]


def csr_struct(M):
    return CSRStruct(_p(M.row_ptr), _p(M.col), _p(M.val))


def point_limbs(curve, group):
    return LIB.mo_point_limbs(curve, group)


def generator(curve, group):
    out = np.zeros(point_limbs(curve, group), dtype=np.uint64)
    LIB.mo_generator(curve, group, _p(out))
    return out


def field_op(field, op, a, b=None):
    fid = FIELD_IDS[field]
    a = np.ascontiguousarray(a, dtype=np.uint64)
    out = np.zeros_like(a)
    n = LIB.mo_field_limbs(fid)
    cnt = a.size // n
    bb = None if b is None else np.ascontiguousarray(b, dtype=np.uint64)
    LIB.mo_field_op(fid, {"add": 0, "sub": 1, "mul": 2, "inv": 3, "from_canonical": 4, "to_canonical": 5, "neg": 6,
                          "sqr": 7}[op], _p(a), _p(bb), _p(out), ctypes.c_size_t(cnt))
    return out


def g_add(curve, group, a, b):
    out = np.zeros(point_limbs(curve, group), dtype=np.uint64)
    LIB.mo_g_add(curve, group, _p(np.ascontiguousarray(a)), _p(np.ascontiguousarray(b)), _p(out))
    return out


def g_mul(curve, group, p, k):
    out = np.zeros(point_limbs(curve, group), dtype=np.uint64)
    LIB.mo_g_mul(curve, group, _p(np.ascontiguousarray(p)), _p(np.ascontiguousarray(k, dtype=np.uint64)), _p(out))
    return out


def g_sum(curve, group, pts):
    pts = np.ascontiguousarray(pts, dtype=np.uint64)
    out = np.zeros(point_limbs(curve, group), dtype=np.uint64)
    LIB.mo_g_sum(curve, group, _p(pts), ctypes.c_size_t(pts.shape[0]), _p(out))
    return out


def on_curve(curve, group, p):
    return bool(LIB.mo_on_curve(curve, group, _p(np.ascontiguousarray(p))))


def fixed_base_mul(curve, group, base, scalars):
    scalars = np.ascontiguousarray(scalars, dtype=np.uint64)
    n = scalars.shape[0]
    out = np.zeros((n, point_limbs(curve, group)), dtype=np.uint64)
    LIB.mo_fixed_base_mul(curve, group, _p(np.ascontiguousarray(base)), _p(scalars), ctypes.c_size_t(n), _p(out))
    return out


def msm(curve, group, bases, scalars, algo=1):
    bases = np.ascontiguousarray(bases, dtype=np.uint64)
    scalars = np.ascontiguousarray(scalars, dtype=np.uint64)
    n = min(bases.shape[0], scalars.shape[0])
    out = np.zeros(point_limbs(curve, group), dtype=np.uint64)
    LIB.mo_msm(curve, group, _p(bases), _p(scalars), ctypes.c_size_t(n), algo, _p(out))
    return out


def set_threads(n):
    """1 = the single-threaded arkworks the reference ships; n > 1 = arkworks-`parallel`-style decomposition; 0 = all
    hardware threads. Returns the count in effect. Only the timed baselines use more than one."""
    return LIB.mo_set_threads(int(n))


def hardware_threads():
    return LIB.mo_hardware_threads()


def time_msm(curve, group, bases, scalars):
    bases = np.ascontiguousarray(bases, dtype=np.uint64)
    scalars = np.ascontiguousarray(scalars, dtype=np.uint64)
    n = min(bases.shape[0], scalars.shape[0])
    out = np.zeros(point_limbs(curve, group), dtype=np.uint64)
    t = LIB.mo_time_msm(curve, group, _p(bases), _p(scalars), ctypes.c_size_t(n), _p(out))
    return t, out


def ntt(curve, data, inverse=False, coset=False):
    data = np.array(data, dtype=np.uint64, copy=True)
    n = data.shape[0]
    log_n = n.bit_length() - 1
    assert 1 << log_n == n
    rc = LIB.mo_ntt(curve, _p(data), log_n, int(inverse), int(coset))
    assert rc == 0
    return data


def point_bytes(curve, group, compressed):
    return LIB.mo_point_bytes(curve, group, int(compressed))


def serialize(curve, group, p, compressed=True):
    nb = point_bytes(curve, group, compressed)
    out = ctypes.create_string_buffer(nb)
    LIB.mo_point_serialize(curve, group, int(compressed), _p(np.ascontiguousarray(p)), out)
    return out.raw


def deserialize(curve, group, data, compressed=True):
    out = np.zeros(point_limbs(curve, group), dtype=np.uint64)
    ok = LIB.mo_point_deserialize(curve, group, int(compressed), bytes(data), _p(out))
    return bool(ok), out


def witness_map(c):
    h = np.zeros((c.D, 4), dtype=np.uint64)
    A, B, C = csr_struct(c.A), csr_struct(c.B), csr_struct(c.C)
    lg = LIB.mo_witness_map(c.curve, ctypes.byref(A), ctypes.byref(B), ctypes.byref(C), ctypes.c_size_t(c.m),
                            ctypes.c_size_t(c.P), _p(c.z), _p(h))
    assert lg >= 0
    return h


class ProvingKey:
    """Host-side proving key in the ABI's memory format (affine Montgomery, infinity = zeros)."""

    def __init__(self, curve, V, P, D, h_len):
        self.curve, self.V, self.P, self.D, self.h_len = curve, V, P, D, h_len
        g1, g2 = point_limbs(curve, 1), point_limbs(curve, 2)
        z = lambda n, w: np.zeros((n, w), dtype=np.uint64)
        self.alpha_g1, self.beta_g1, self.delta_g1 = z(1, g1), z(1, g1), z(1, g1)
        self.beta_g2, self.gamma_g2, self.delta_g2 = z(1, g2), z(1, g2), z(1, g2)
        self.gamma_abc_g1 = z(P, g1)
        self.a_query, self.b_g1_query, self.b_g2_query = z(V, g1), z(V, g1), z(V, g2)
        self.h_query, self.l_query = z(h_len, g1), z(V - P, g1)

    def struct(self):
        return PKStruct(self.V, self.P, self.D, self.h_len, _p(self.alpha_g1), _p(self.beta_g1), _p(self.delta_g1),
                        _p(self.beta_g2), _p(self.gamma_g2), _p(self.delta_g2), _p(self.gamma_abc_g1),
                        _p(self.a_query), _p(self.b_g1_query), _p(self.b_g2_query), _p(self.h_query),
                        _p(self.l_query))


def groth16_setup(c, toxic_mont):
    """toxic_mont: uint64 [5,4] Montgomery (tau, alpha, beta, gamma, delta)."""
    pk = ProvingKey(c.curve, c.V, c.P, c.D, c.D - 1)
    A, B, C = csr_struct(c.A), csr_struct(c.B), csr_struct(c.C)
    toxic = np.ascontiguousarray(toxic_mont, dtype=np.uint64)
    rc = LIB.mo_groth16_setup(c.curve, ctypes.byref(A), ctypes.byref(B), ctypes.byref(C), ctypes.c_size_t(c.m),
                              ctypes.c_size_t(c.P), ctypes.c_size_t(c.V), _p(toxic), _p(pk.alpha_g1), _p(pk.beta_g1),
                              _p(pk.delta_g1), _p(pk.beta_g2), _p(pk.gamma_g2), _p(pk.delta_g2), _p(pk.gamma_abc_g1),
                              _p(pk.a_query), _p(pk.b_g1_query), _p(pk.b_g2_query), _p(pk.h_query), _p(pk.l_query))
    assert rc >= 0
    return pk


def proof_bytes(curve):
    return 2 * point_bytes(curve, 1, True) + point_bytes(curve, 2, True)


def _pk_struct(pk):
    if hasattr(pk, "struct"):
        return pk.struct()
    keep = [np.ascontiguousarray(getattr(pk, f), dtype=np.uint64) for f in (
        "alpha_g1", "beta_g1", "delta_g1", "beta_g2", "gamma_g2", "delta_g2", "gamma_abc_g1", "a_query", "b_g1_query",
        "b_g2_query", "h_query", "l_query")]
    st = PKStruct(pk.V, pk.P, pk.D, pk.h_len, *[_p(a) for a in keep])
    st._keep = keep
    return st


def groth16_prove(c, pk, r_mont, s_mont, msm_algo=1, z=None):
    out = ctypes.create_string_buffer(proof_bytes(c.curve))
    A, B, C = csr_struct(c.A), csr_struct(c.B), csr_struct(c.C)
    pks = _pk_struct(pk)
    zz = c.z if z is None else np.ascontiguousarray(z, dtype=np.uint64)
    rc = LIB.mo_groth16_prove(c.curve, ctypes.byref(pks), ctypes.byref(A), ctypes.byref(B), ctypes.byref(C),
                              ctypes.c_size_t(c.m), _p(zz), _p(np.ascontiguousarray(r_mont, dtype=np.uint64)),
                              _p(np.ascontiguousarray(s_mont, dtype=np.uint64)), msm_algo, out, None)
    assert rc == 0
    return out.raw


def groth16_verify(curve, pk, inputs_mont, proof):
    pks = _pk_struct(pk)
    inputs = np.ascontiguousarray(inputs_mont, dtype=np.uint64)
    return LIB.mo_groth16_verify(curve, ctypes.byref(pks), _p(inputs), bytes(proof))


def pairing_bytes(curve, P, Q, ark_exp=False):
    nb = 12 * (point_bytes(curve, 1, True))
    out = ctypes.create_string_buffer(nb)
    LIB.mo_pairing_bytes(curve, _p(np.ascontiguousarray(P)), _p(np.ascontiguousarray(Q)), int(ark_exp), out)
    return out.raw


def pairing_bytes_pow(curve, P, Q, mult):
    """textbook pairing value raised to `mult`, arkworks Fq12 byte order"""
    nb = 12 * (point_bytes(curve, 1, True))
    out = ctypes.create_string_buffer(nb)
    LIB.mo_pairing_bytes_pow(curve, _p(np.ascontiguousarray(P)), _p(np.ascontiguousarray(Q)), ctypes.c_uint64(mult), out)
    return out.raw


def group_ntt(curve, group, pts, inverse=False):
    """Radix2EvaluationDomain fft / ifft over a vector of 2^k group elements (affine in / out)"""
    pts = np.array(pts, dtype=np.uint64, copy=True)
    n = pts.shape[0]
    lg = n.bit_length() - 1
    assert 1 << lg == n
    assert LIB.mo_group_ntt(curve, group, _p(pts), lg, int(bool(inverse))) == 0
    return pts
