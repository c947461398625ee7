"""ctypes binding of libmantagpu.so + the Python mirror of the reference's prover interface.

Every call goes through the C ABI declared in include/mantagpu.h -- the same entry points the Rust
shim of INTEGRATION.md binds. There is no CPU path here: if the shared library (built by
`__graft_entry__.build()` / `make -C manta_rs_amd/csrc`) is absent, import raises.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MANTA_LIB") or os.path.join(_HERE, "lib", "libmantagpu.so")  # MANTA_LIB: A/B builds only

BN254, BLS12_381 = 0, 1
FQ_LIMBS = {BN254: 4, BLS12_381: 6}


class MantaGpuError(RuntimeError):
    """Mirror of the reference's opaque unit `Error` (manta-crypto/src/arkworks/groth16.rs:50-60),
    carrying the status code and the library's detail string."""

    def __init__(self, status, where=""):
        self.status = status
        detail = LIB.mg_last_error().decode() if status == 2 else ""
        super().__init__(f"{where}: {LIB.mg_strerror(status).decode()} ({status}) {detail}")


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the MI355X HIP library is the product and there is no CPU fallback. "
            "Run `python -c 'import __graft_entry__ as g; g.build()'` (or `make -C manta_rs_amd/csrc`).")
    try:  # share torch's HIP runtime if torch is (or will be) in this process
        import torch  # noqa: F401
    except Exception:  # pragma: no cover
        pass
    lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    lib.mg_strerror.restype = ctypes.c_char_p
    lib.mg_last_error.restype = ctypes.c_char_p
    lib.mg_bases_device_bytes.restype = ctypes.c_size_t
    lib.mg_ctx_domain_size.restype = ctypes.c_uint64
    lib.mg_ctx_num_variables.restype = ctypes.c_uint64
    lib.mg_ctx_num_inputs.restype = ctypes.c_uint64
    lib.mg_last_accumulate_ms.restype = ctypes.c_float
    lib.mg_last_accumulate_mhz.restype = ctypes.c_float
    lib.mg_vk_encoded_size.restype = ctypes.c_size_t
    lib.mg_vk_num_inputs.restype = ctypes.c_uint64
    lib.mg_xyzz_limbs.restype = ctypes.c_size_t
    lib.mg_partials_slot_limbs.restype = ctypes.c_size_t
    return lib


LIB = _load()
_vp = ctypes.c_void_p
_sz = ctypes.c_size_t

EXPORTS = [
    "mg_init", "mg_strerror", "mg_last_error", "mg_device_count", "mg_malloc", "mg_free", "mg_memcpy_h2d",
    "mg_memcpy_d2h", "mg_device_synchronize", "mg_host_alloc", "mg_host_free", "mg_set_kernel_timing", "mg_last_accumulate_ms", "mg_bases_create", "mg_bases_destroy", "mg_bases_device_bytes",
    "mg_msm", "mg_msm_launch", "mg_msm_finish", "mg_points_sum", "mg_fixed_base_mul", "mg_ec_elementwise", "mg_point_serialize", "mg_ntt",
    "mg_ntt_device", "mg_groth16_setup", "mg_ctx_create", "mg_ctx_create_from_bytes", "mg_ctx_set_r1cs", "mg_groth16_prove", "mg_groth16_prove_batch", "mg_witness_map", "mg_ctx_domain_size", "mg_ctx_table_bytes",
    "mg_ctx_destroy", "mg_bases_create_sharded", "mg_bases_num_shards", "mg_bases_shard", "mg_msm_launch_sharded",
    "mg_ctx_create_sharded", "mg_ctx_create_from_bytes_sharded", "mg_ctx_num_variables", "mg_ctx_num_inputs",
    "mg_ctx_num_shards", "mg_field_op", "mg_vk_create", "mg_vk_create_from_bytes", "mg_vk_encoded_size", "mg_vk_encode",
    "mg_vk_alpha_beta", "mg_vk_num_inputs", "mg_vk_destroy", "mg_groth16_verify", "mg_groth16_verify_batch", "mg_pairing_check", "mg_proof_decode", "mg_group_ntt",
    "mg_msm_result_to_device", "mg_xyzz_limbs", "mg_xyzz_sum", "mg_ctx_create_shard", "mg_partials_slot_limbs",
    "mg_groth16_partials_launch", "mg_groth16_partials_finish", "mg_groth16_assemble", "mg_blake3", "mg_ctx_create_from_bytes_checked",
    "mg_last_ntt_ms", "mg_hw_queues", "mg_last_prove_phases_ms", "mg_clock_probe", "mg_last_accumulate_mhz", "mg_ctx_create_task",
    "mg_ctx_opts_init", "mg_ctx_create_ex", "mg_ctx_create_from_bytes_ex", "mg_last_pass_host_ms",
    "mg_tuning_init", "mg_get_tuning", "mg_set_tuning", "mg_tuning_env_names",
]


def _chk(rc, where):
    if rc != 0:
        raise MantaGpuError(rc, where)


def _p(a):
    return None if a is None else a.ctypes.data_as(_vp)


def _u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


def _addr(x):
    """a device / stream address given as int, ctypes.c_void_p or DeviceBuffer -> c_void_p"""
    if x is None:
        return None
    if isinstance(x, ctypes.c_void_p):
        return x
    if hasattr(x, "ptr"):
        return _addr(x.ptr)
    return _vp(int(x))


def init(device=0):
    _chk(LIB.mg_init(int(device)), "mg_init")


def device_count():
    c = ctypes.c_int(0)
    _chk(LIB.mg_device_count(ctypes.byref(c)), "mg_device_count")
    return c.value


def set_kernel_timing(on):
    _chk(LIB.mg_set_kernel_timing(int(bool(on))), "mg_set_kernel_timing")


def last_accumulate_ms():
    return float(LIB.mg_last_accumulate_ms())


def last_accumulate_mhz():
    return float(LIB.mg_last_accumulate_mhz())


def clock_probe(iters=200000):
    """`mg_clock_probe`: (MHz the s_memtime counter ran at, wave-level v_mad_u64_u32 issued per SIMD and microsecond, probe ms)
    under an all-SIMD integer multiply-add load of two wavefronts per SIMD"""
    a, b, c = ctypes.c_double(0), ctypes.c_double(0), ctypes.c_double(0)
    _chk(LIB.mg_clock_probe(ctypes.c_uint(iters), ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)), "mg_clock_probe")
    return a.value, b.value, c.value


def last_ntt_ms():
    """with kernel timing on: (whole call, conversion in, butterfly passes, conversion out) of this thread's last NTT, ms"""
    v = (ctypes.c_float * 4)()
    _chk(LIB.mg_last_ntt_ms(v), "mg_last_ntt_ms")
    return [float(x) for x in v]


def hw_queues():
    """(normal-priority, high-priority) hardware queues the library found behind the current device's streams; (0, 0) before the
    first ProvingContext of the process or with MANTA_QUEUE_AWARE=0 (mantagpu.h mg_hw_queues)"""
    v = (ctypes.c_int * 2)()
    _chk(LIB.mg_hw_queues(v), "mg_hw_queues")
    return int(v[0]), int(v[1])


def last_prove_phases_ms():
    """with kernel timing on: the phase split of this thread's last single proof (see mantagpu.h), ms"""
    v = (ctypes.c_float * 10)()
    _chk(LIB.mg_last_prove_phases_ms(v), "mg_last_prove_phases_ms")
    names = ("upload_z", "witness_map", "msm_a", "msm_b_g1", "msm_b_g2", "msm_l", "msm_h", "part_a_upload_to_join", "g2_chain_upload_to_end",
             "host_assembly_after_gpu")
    return {n: round(float(x), 4) for n, x in zip(names, v)}


def last_pass_host_ms():
    """host side of this thread's last proving pass: ms spent enqueuing, waiting for the GPU, assembling (`mg_last_pass_host_ms`)"""
    v = (ctypes.c_float * 3)()
    _chk(LIB.mg_last_pass_host_ms(v), "mg_last_pass_host_ms")
    return {"enqueue": round(float(v[0]), 4), "wait_gpu": round(float(v[1]), 4), "assemble": round(float(v[2]), 4)}


def synchronize():
    _chk(LIB.mg_device_synchronize(), "mg_device_synchronize")


class PinnedArray:
    """A numpy array in page-locked host memory (`mg_host_alloc`): assignments kept in one are uploaded to the GPU
    without the library's staging copy. `PinnedArray.like(arr).array` is a pinned copy of `arr`."""

    def __init__(self, shape, dtype=np.uint64):
        self.nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        ptr = _vp()
        _chk(LIB.mg_host_alloc(ctypes.byref(ptr), _sz(self.nbytes)), "mg_host_alloc")
        self._ptr = ptr
        buf = (ctypes.c_uint8 * self.nbytes).from_address(ptr.value)
        self.array = np.frombuffer(buf, dtype=dtype).reshape(shape)

    @classmethod
    def like(cls, arr):
        arr = np.ascontiguousarray(arr)
        p = cls(arr.shape, arr.dtype)
        p.array[...] = arr
        return p

    def free(self):
        if self._ptr is not None and self._ptr.value:
            self.array = None
            LIB.mg_host_free(self._ptr)
            self._ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DeviceBuffer:
    """A raw HBM allocation owned by the library's HIP runtime."""

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        ptr = _vp()
        _chk(LIB.mg_malloc(ctypes.byref(ptr), _sz(self.nbytes)), "mg_malloc")
        self.ptr = ptr

    @classmethod
    def from_numpy(cls, arr):
        arr = np.ascontiguousarray(arr)
        b = cls(arr.nbytes)
        _chk(LIB.mg_memcpy_h2d(b.ptr, _p(arr), _sz(arr.nbytes)), "mg_memcpy_h2d")
        return b

    def to_numpy(self, dtype=np.uint64, shape=None):
        out = np.empty(self.nbytes // np.dtype(dtype).itemsize, dtype=dtype)
        _chk(LIB.mg_memcpy_d2h(_p(out), self.ptr, _sz(self.nbytes)), "mg_memcpy_d2h")
        return out if shape is None else out.reshape(shape)

    def offset(self, nbytes):
        return _vp(self.ptr.value + int(nbytes))

    def free(self):
        if self.ptr is not None and self.ptr.value:
            LIB.mg_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def affine_limbs(curve, group):
    return 2 * FQ_LIMBS[curve] * (2 if group == 2 else 1)


class Bases:
    """A static vector of curve points resident in HBM (the `bases: &[G::Affine]` argument of
    `VariableBaseMSM::multi_scalar_mul`, made persistent because proving-key queries never change)."""

    def __init__(self, curve, group, points, precompute_window_bits=0, on_device=False, devices=None):
        """devices: a list of HIP device indices -> the vector is range-sharded over them (`mg_bases_create_sharded`;
        a device may repeat); None -> one shard on the current device."""
        self.curve, self.group = curve, group
        h = _vp()
        if devices is not None:
            pts = _u64(points)
            assert pts.ndim == 2 and pts.shape[1] == affine_limbs(curve, group), pts.shape
            self.n = pts.shape[0]
            dv = (ctypes.c_int * len(devices))(*devices)
            _chk(LIB.mg_bases_create_sharded(curve, group, _p(pts), _sz(self.n), dv, len(devices),
                                             int(precompute_window_bits), ctypes.byref(h)), "mg_bases_create_sharded")
        elif on_device:
            ptr, n = points
            _chk(LIB.mg_bases_create(curve, group, ptr, _sz(n), 1, int(precompute_window_bits), ctypes.byref(h)),
                 "mg_bases_create")
            self.n = n
        else:
            pts = _u64(points)
            assert pts.ndim == 2 and pts.shape[1] == affine_limbs(curve, group), pts.shape
            self.n = pts.shape[0]
            _chk(LIB.mg_bases_create(curve, group, _p(pts), _sz(self.n), 0, int(precompute_window_bits),
                                     ctypes.byref(h)), "mg_bases_create")
        self.handle = h
        self.precompute_window_bits = int(precompute_window_bits)

    def device_bytes(self):
        return LIB.mg_bases_device_bytes(self.handle)

    def shards(self):
        """[(device, lo, hi)] of the range shards."""
        out = []
        for g in range(LIB.mg_bases_num_shards(self.handle)):
            dev, lo, hi = ctypes.c_int(0), _sz(0), _sz(0)
            _chk(LIB.mg_bases_shard(self.handle, g, ctypes.byref(dev), ctypes.byref(lo), ctypes.byref(hi)), "mg_bases_shard")
            out.append((dev.value, lo.value, hi.value))
        return out

    def close(self):
        if self.handle is not None:
            LIB.mg_bases_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MsmJob:
    def __init__(self, bases, handle):
        self.bases, self.handle = bases, handle

    def finish(self):
        out = np.zeros(affine_limbs(self.bases.curve, self.bases.group), dtype=np.uint64)
        _chk(LIB.mg_msm_finish(self.handle, _p(out)), "mg_msm_finish")
        self.handle = None
        return out

    def result_to_device(self, d_out_ptr, stream=None):
        """`mg_msm_result_to_device`: the window sums are folded on the GPU and the XYZZ result is written to device memory
        at d_out_ptr (xyzz_limbs u64); `stream` (raw hipStream_t as int, e.g. torch.cuda.current_stream().cuda_stream; None or
        0 = the default stream) is made to wait for it. No host synchronisation. The job must still be released with release()."""
        _chk(LIB.mg_msm_result_to_device(self.handle, _addr(d_out_ptr), _addr(stream)), "mg_msm_result_to_device")

    def release(self):
        """waits for the job and frees it without converting the result (its consumer took it on the device)"""
        _chk(LIB.mg_msm_finish(self.handle, None), "mg_msm_finish")
        self.handle = None


class VariableBaseMSM:
    """Mirror of ark_ec::msm::VariableBaseMSM (ark-ec 0.3.0; call sites in ark-groth16 create_proof,
    reached from manta-crypto/src/arkworks/groth16.rs:597)."""

    @staticmethod
    def multi_scalar_mul(bases: Bases, scalars) -> np.ndarray:
        """scalars: (n,4) uint64 canonical (`into_repr`). Returns the affine result (Montgomery limbs,
        infinity = zeros). Like arkworks, zips to the shorter of bases/scalars."""
        sc = _u64(scalars)
        out = np.zeros(affine_limbs(bases.curve, bases.group), dtype=np.uint64)
        _chk(LIB.mg_msm(bases.handle, _p(sc), _sz(sc.shape[0]), _p(out)), "mg_msm")
        return out

    @staticmethod
    def launch(bases: Bases, d_scalars, n, scalars_mont=False, window_bits=0, sparse=False) -> MsmJob:
        """Scalars already in HBM (DeviceBuffer or raw pointer); returns a job to `finish()`."""
        ptr = d_scalars.ptr if isinstance(d_scalars, DeviceBuffer) else d_scalars
        h = _vp()
        _chk(LIB.mg_msm_launch(bases.handle, ptr, _sz(n), int(bool(scalars_mont)) | (2 if sparse else 0), int(window_bits),
                               ctypes.byref(h)),
             "mg_msm_launch")
        return MsmJob(bases, h)

    @staticmethod
    def launch_sharded(bases: Bases, d_scalars_per_shard, scalars_mont=False, window_bits=0, sparse=False) -> MsmJob:
        """Sharded bases, scalars already resident: one DeviceBuffer / pointer per shard, on that shard's device."""
        ptrs = [d.ptr if isinstance(d, DeviceBuffer) else d for d in d_scalars_per_shard]
        arr = (_vp * len(ptrs))(*ptrs)
        h = _vp()
        _chk(LIB.mg_msm_launch_sharded(bases.handle, arr, int(bool(scalars_mont)) | (2 if sparse else 0), int(window_bits),
                                       ctypes.byref(h)), "mg_msm_launch_sharded")
        return MsmJob(bases, h)


def blake3(data: bytes) -> bytes:
    """`blake3::hash` (manta-parameters' checksum, lib.rs:173-177), computed by the library's host code"""
    out = ctypes.create_string_buffer(32)
    _chk(LIB.mg_blake3(bytes(data), _sz(len(data)), out), "mg_blake3")
    return out.raw


def xyzz_limbs(curve, group) -> int:
    return int(LIB.mg_xyzz_limbs(curve, group))


def xyzz_sum(curve, group, xyzz) -> np.ndarray:
    """sum of XYZZ points (arkworks Montgomery limbs X | Y | ZZ | ZZZ, ZZ = 0: infinity) -> one affine point"""
    pts = _u64(xyzz).reshape(-1, xyzz_limbs(curve, group))
    out = np.zeros(affine_limbs(curve, group), dtype=np.uint64)
    _chk(LIB.mg_xyzz_sum(curve, group, _p(pts), _sz(pts.shape[0]), _p(out)), "mg_xyzz_sum")
    return out


def points_sum(curve, group, points):
    pts = _u64(points)
    out = np.zeros(affine_limbs(curve, group), dtype=np.uint64)
    _chk(LIB.mg_points_sum(curve, group, _p(pts), _sz(pts.shape[0]), _p(out)), "mg_points_sum")
    return out


def fixed_base_mul(curve, group, base, d_scalars: DeviceBuffer, n) -> DeviceBuffer:
    out = DeviceBuffer(n * affine_limbs(curve, group) * 8)
    _chk(LIB.mg_fixed_base_mul(curve, group, _p(_u64(base)), d_scalars.ptr, _sz(n), out.ptr), "mg_fixed_base_mul")
    return out


EC_ADD_MIXED, EC_ADD, EC_DOUBLE, EC_MUL, EC_SUB_MIXED, EC_MUL_FIXED = range(6)


def group_ntt(curve, group, points, inverse=False) -> np.ndarray:
    """`Radix2EvaluationDomain::{fft, ifft}` over a vector of 2^k group elements (`mg_group_ntt`)."""
    pts = _u64(points)
    n = pts.shape[0]
    lg = n.bit_length() - 1
    assert 1 << lg == n and pts.shape[1] == affine_limbs(curve, group)
    out = np.zeros_like(pts)
    _chk(LIB.mg_group_ntt(curve, group, _p(pts), lg, int(bool(inverse)), _p(out)), "mg_group_ntt")
    return out


def ec_elementwise(curve, group, op, a, b=None) -> np.ndarray:
    """Element-wise group operation on arrays of affine points (`mg_ec_elementwise`; the ecc.rs primitive menu):
    a, b = [n, limbs] uint64 affine points (b = [n, 4] canonical scalars for EC_MUL, unused for EC_DOUBLE)."""
    a = _u64(a)
    n = a.shape[0]
    out = np.zeros_like(a)
    bb = _p(_u64(b)) if b is not None else None
    _chk(LIB.mg_ec_elementwise(curve, group, op, _p(a), bb, _sz(n), _p(out)), "mg_ec_elementwise")
    return out


FIELD_IDS = {"bn254_fr": 0, "bn254_fq": 1, "bls381_fr": 2, "bls381_fq": 3}
FIELD_OPS = {"add": 0, "sub": 1, "mul": 2, "sqr": 3, "neg": 4, "from_canonical": 5, "to_canonical": 6, "inv": 7}


def field_op(field, op, a, b=None, repr=0, lazy_a=0, lazy_b=0) -> np.ndarray:
    """Element-wise field arithmetic on the GPU (`mg_field_op`): a, b = [n, limbs] uint64 Montgomery elements of
    `field` ("bn254_fr" ...); repr 0 = the saturated Montgomery arithmetic, 1 = the MSM kernels' reduced-radix lazy
    arithmetic on the representatives a + lazy_a p, b + lazy_b p."""
    a = _u64(a)
    nl = 6 if field == "bls381_fq" else 4
    assert a.ndim == 2 and a.shape[1] == nl, a.shape
    out = np.zeros_like(a)
    bb = None if b is None else _u64(b)
    assert bb is None or bb.shape == a.shape
    _chk(LIB.mg_field_op(FIELD_IDS[field], FIELD_OPS[op], int(repr), int(lazy_a), int(lazy_b), _p(a), _p(bb), _sz(a.shape[0]),
                         _p(out)), "mg_field_op")
    return out


def point_serialize(curve, group, point, compressed=True):
    nb = FQ_LIMBS[curve] * 8 * (2 if group == 2 else 1) * (1 if compressed else 2)
    out = ctypes.create_string_buffer(nb)
    _chk(LIB.mg_point_serialize(curve, group, _p(_u64(point)), int(compressed), out), "mg_point_serialize")
    return out.raw


class Radix2EvaluationDomain:
    """Mirror of ark_poly::Radix2EvaluationDomain<Fr> (ark-poly 0.3.0)."""

    def __init__(self, curve, size):
        self.curve = curve
        self.log_size = max(0, (int(size) - 1).bit_length())
        self.size = 1 << self.log_size

    def _run(self, data, inverse, coset):
        d = np.array(data, dtype=np.uint64, copy=True)
        assert d.shape == (self.size, 4)
        _chk(LIB.mg_ntt(self.curve, _p(d), self.log_size, int(inverse), int(coset)), "mg_ntt")
        return d

    def fft(self, data):
        return self._run(data, False, False)

    def ifft(self, data):
        return self._run(data, True, False)

    def coset_fft(self, data):
        return self._run(data, False, True)

    def coset_ifft(self, data):
        return self._run(data, True, True)

    def fft_device(self, dbuf: DeviceBuffer, inverse=False, coset=False):
        _chk(LIB.mg_ntt_device(self.curve, dbuf.ptr, self.log_size, int(inverse), int(coset)), "mg_ntt_device")


# ------------------------------------------------------------------------------------------------ Groth16
class _PkView(ctypes.Structure):
    _fields_ = [("n_vars", ctypes.c_uint64), ("n_inputs", ctypes.c_uint64), ("h_len", ctypes.c_uint64)] + [
        (k, _vp) for k in ("alpha_g1", "beta_g1", "delta_g1", "beta_g2", "delta_g2", "a_query", "b_g1_query",
                           "b_g2_query", "h_query", "l_query")]


class _Csr(ctypes.Structure):
    _fields_ = [("row_ptr", _vp), ("col", _vp), ("val", _vp), ("nnz", ctypes.c_uint64)]


class R1CS:
    """Mirror of manta_crypto::arkworks::constraint::R1CS<F> as the prover sees it: the matrices of
    `cs.to_matrices()` and the full assignment z = instance || witness (Montgomery Fr)."""

    def __init__(self, curve, A, B, C, num_constraints, num_instance, z):
        self.curve, self.A, self.B, self.C = curve, A, B, C
        self.num_constraints, self.num_instance = int(num_constraints), int(num_instance)
        self.z = _u64(z)

    @classmethod
    def from_circuit(cls, c):
        return cls(c.curve, c.A, c.B, c.C, c.m, c.P, c.z)


class _PkOut(ctypes.Structure):
    _fields_ = [(n, _vp) for n in ("alpha_g1", "beta_g1", "delta_g1", "beta_g2", "gamma_g2", "delta_g2", "gamma_abc_g1",
                                   "a_query", "b_g1_query", "b_g2_query", "h_query", "l_query")]


class ProvingKey:
    """Host arrays in the C ABI's memory format (affine Montgomery limbs, infinity = zeros): the fields of
    ark_groth16::ProvingKey (incl. its verifying key) that `ProvingContext` and a verifier consume."""
    pass


def groth16_setup(r1cs: "R1CS", n_vars, toxic_mont, g1_generator, g2_generator) -> ProvingKey:
    """Mirror of `Groth16::compile` (manta-crypto/src/arkworks/groth16.rs:571-586) with the randomness made explicit
    (`mg_groth16_setup`): toxic_mont = alpha, beta, gamma, delta, tau as Montgomery Fr (5 x 4 u64), the generators as
    affine points. Every group element of the key is computed on the GPU."""
    curve, m, P, V = r1cs.curve, r1cs.num_constraints, r1cs.num_instance, int(n_vars)
    D = 1
    while D < m + P:
        D <<= 1
    w1, w2 = affine_limbs(curve, 1), affine_limbs(curve, 2)
    pk = ProvingKey()
    pk.curve, pk.V, pk.P, pk.D, pk.h_len = curve, V, P, D, D - 1
    shapes = {"alpha_g1": (1, w1), "beta_g1": (1, w1), "delta_g1": (1, w1), "beta_g2": (1, w2), "gamma_g2": (1, w2),
              "delta_g2": (1, w2), "gamma_abc_g1": (P, w1), "a_query": (V, w1), "b_g1_query": (V, w1),
              "b_g2_query": (V, w2), "h_query": (D - 1, w1), "l_query": (V - P, w1)}
    for k, sh in shapes.items():
        setattr(pk, k, np.zeros(sh, dtype=np.uint64))
    out = _PkOut(*[_p(getattr(pk, k)) for k, _ in _PkOut._fields_])
    ms = []
    for M in (r1cs.A, r1cs.B, r1cs.C):
        rp = np.ascontiguousarray(M.row_ptr, dtype=np.uint32)
        col = np.ascontiguousarray(M.col, dtype=np.uint32)
        val = _u64(M.val)
        ms.append((rp, col, val, _Csr(_p(rp), _p(col), _p(val), len(col))))
    tox = _u64(toxic_mont).reshape(5, 4)
    _chk(LIB.mg_groth16_setup(curve, ctypes.byref(ms[0][3]), ctypes.byref(ms[1][3]), ctypes.byref(ms[2][3]),
                              ctypes.c_uint64(m), ctypes.c_uint64(V), ctypes.c_uint64(P), _p(tox), _p(_u64(g1_generator)),
                              _p(_u64(g2_generator)), ctypes.byref(out)), "mg_groth16_setup")
    return pk


EXCHANGE_HOST, EXCHANGE_RCCL = 0, 1


class Tuning(ctypes.Structure):
    """`mg_tuning` (include/mantagpu.h): what a deployment decides about the library's scheduling -- process-wide through
    get_tuning / set_tuning, per context through ProvingContext(tuning=...). No field changes a result."""
    _fields_ = [("struct_size", ctypes.c_uint32), ("graph_mode", ctypes.c_int32), ("graph_mode_batch", ctypes.c_int32),
                ("prove_streams", ctypes.c_int32), ("linear_chains", ctypes.c_int32), ("coalesce_inflight", ctypes.c_int32),
                ("coalesce_gather_us", ctypes.c_int32), ("batch_inflight", ctypes.c_int32), ("queue_aware", ctypes.c_int32),
                ("msm_dedicated_queues", ctypes.c_int32), ("window_bits_narrow", ctypes.c_int32), ("window_bits_wide", ctypes.c_int32),
                ("window_bits_h", ctypes.c_int32), ("window_bits_g2", ctypes.c_int32), ("full_table_bytes", ctypes.c_int64)]

    def replace(self, **kw):
        t = Tuning.from_buffer_copy(bytes(self))
        for k, v in kw.items():
            if k not in dict(Tuning._fields_) or k == "struct_size":
                raise ValueError("mg_tuning has no field %r" % k)
            setattr(t, k, int(v))
        return t

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in Tuning._fields_ if k != "struct_size"}


GRAPH_OFF, GRAPH_SINGLE, GRAPH_SPLIT = 0, 1, 2


def tuning_defaults() -> Tuning:
    t = Tuning()
    _chk(LIB.mg_tuning_init(ctypes.byref(t)), "mg_tuning_init")
    assert t.struct_size == ctypes.sizeof(Tuning), "mg_tuning: the Python mirror is out of date"
    return t


def get_tuning() -> Tuning:
    t = Tuning()
    _chk(LIB.mg_get_tuning(ctypes.byref(t)), "mg_get_tuning")
    return t


def set_tuning(t: Tuning = None, **kw):
    """process-wide tuning for contexts created from now on: a whole struct, or keyword changes to the values in force"""
    t = (t if t is not None else get_tuning()).replace(**kw)
    _chk(LIB.mg_set_tuning(ctypes.byref(t)), "mg_set_tuning")


def tuning_env_names() -> list:
    """the environment variables the shipped library reads (mg_tuning_env_names)"""
    LIB.mg_tuning_env_names.restype = ctypes.POINTER(ctypes.c_char_p)
    arr, out, i = LIB.mg_tuning_env_names(), [], 0
    while arr[i]:
        out.append(arr[i].decode())
        i += 1
    return out


class _CtxOpts(ctypes.Structure):
    """`mg_ctx_opts` (include/mantagpu.h)"""
    _fields_ = [("struct_size", ctypes.c_uint32), ("exchange", ctypes.c_uint32), ("full_table_bytes", ctypes.c_int64),
                ("devices", ctypes.POINTER(ctypes.c_int)), ("n_devices", ctypes.c_int32), ("shard", ctypes.c_int32),
                ("n_shards", ctypes.c_int32), ("task_mask", ctypes.c_uint32), ("tuning", ctypes.POINTER(Tuning))]


def _ctx_opts(devices=None, shard=None, task_mask=None, full_table_bytes=None, exchange=None, tuning=None):
    """-> (mg_ctx_opts, keep-alive) from the keyword arguments of ProvingContext"""
    o = _CtxOpts()
    _chk(LIB.mg_ctx_opts_init(ctypes.byref(o)), "mg_ctx_opts_init")
    assert o.struct_size == ctypes.sizeof(_CtxOpts), "mg_ctx_opts: the Python mirror is out of date"
    keep = None
    if devices is not None:
        keep = (ctypes.c_int * len(devices))(*devices)
        o.devices, o.n_devices = keep, len(devices)
    if shard is not None:
        o.shard, o.n_shards = int(shard[0]), int(shard[1])
    if task_mask is not None:
        o.task_mask = int(task_mask)
    if full_table_bytes is not None:
        o.full_table_bytes = int(full_table_bytes)
    if exchange is not None:
        o.exchange = int(exchange)
    if tuning is not None:
        if isinstance(tuning, dict):
            tuning = get_tuning().replace(**tuning)
        o.tuning = ctypes.pointer(tuning)
        keep = (keep, tuning)
    return o, keep


class ProvingContext:
    """Mirror of groth16::ProvingContext<E> (manta-crypto/src/arkworks/groth16.rs:216-245): owns the
    device-resident proving key; created once, shared by every proof of the shape."""

    def __init__(self, curve, pk, devices=None, shard=None, task_mask=None, full_table_bytes=None, exchange=None, tuning=None):
        """pk: object with numpy arrays alpha_g1, beta_g1, delta_g1, beta_g2, delta_g2, a_query,
        b_g1_query, b_g2_query, h_query, l_query (affine Montgomery limbs) and ints V, P.
        devices: list of HIP device indices -> every MSM of a proof is range-sharded over them
        (`mg_ctx_create_sharded`); None -> the current device.
        shard = (g, G): this PROCESS holds slice g of G of every query on the current device (`mg_ctx_create_shard`,
        one process per GPU; see distributed.ShardedProver).
        full_table_bytes: HBM budget of the context's full tables (None = the library's default, a tenth of the device's
        HBM; 0 = bucket tables only); exchange: EXCHANGE_HOST / EXCHANGE_RCCL for a `devices` list (`mg_ctx_opts`);
        tuning: this context's `Tuning` (or a dict of field changes to the process-wide values), `mg_ctx_opts.tuning`."""
        self.curve = curve
        self._keep = [_u64(getattr(pk, k)) for k in ("alpha_g1", "beta_g1", "delta_g1", "beta_g2", "delta_g2",
                                                     "a_query", "b_g1_query", "b_g2_query", "h_query", "l_query")]
        v = _PkView(pk.V, pk.P, self._keep[8].shape[0], *[_p(a) for a in self._keep])
        h = _vp()
        if task_mask is not None and int(task_mask) == 0:  # a rank beyond the fifth owns no MSM: the struct reads 0 as "all five"
            _chk(LIB.mg_ctx_create_task(curve, ctypes.byref(v), ctypes.c_uint(0), ctypes.byref(h)), "mg_ctx_create_task")
        elif shard is not None and int(shard[1]) == 1 and devices is None and task_mask is None and full_table_bytes is None and tuning is None:
            # a world of one driven through the partials interface: the entry point that says so (no combined a | b_g1 | l table)
            _chk(LIB.mg_ctx_create_shard(curve, ctypes.byref(v), 0, 1, ctypes.byref(h)), "mg_ctx_create_shard")
        else:
            o, keep = _ctx_opts(devices, shard, task_mask, full_table_bytes, exchange, tuning)
            _chk(LIB.mg_ctx_create_ex(curve, ctypes.byref(v), ctypes.byref(o), ctypes.byref(h)), "mg_ctx_create_ex")
            del keep
        self._keep = None  # the library copied everything
        self.handle = h
        self._r1cs_ref = None

    @classmethod
    def decode(cls, curve, data: bytes, devices=None, checksum: bytes = None, full_table_bytes=None, exchange=None, tuning=None):
        """Mirror of `impl Decode for ProvingContext` (groth16.rs:268-288): arkworks `deserialize_unchecked`
        bytes of the ProvingKey -- the format of manta-parameters' proving-key files. checksum: the file's BLAKE3 digest as
        manta-parameters' data.checkfile lists it (32 bytes); a mismatch raises before anything is uploaded, whatever the
        placement (`manta_parameters::verify`, manta-parameters/src/lib.rs:173-177; `mg_ctx_create_from_bytes_ex`)."""
        self = cls.__new__(cls)
        self.curve = curve
        self._keep = None
        self._r1cs_ref = None
        h = _vp()
        if checksum is not None and len(checksum) != 32:
            raise ValueError("a BLAKE3 digest is 32 bytes")
        o, keep = _ctx_opts(devices, None, None, full_table_bytes, exchange, tuning)
        _chk(LIB.mg_ctx_create_from_bytes_ex(curve, bytes(data), _sz(len(data)), None if checksum is None else bytes(checksum),
                                             ctypes.byref(o), ctypes.byref(h)), "mg_ctx_create_from_bytes_ex")
        del keep
        self.handle = h
        return self

    @property
    def num_variables(self):
        """V: the number of Fr elements every assignment must have (from the C context, so the decode() path knows it too)."""
        return int(LIB.mg_ctx_num_variables(self.handle))

    @property
    def num_inputs(self):
        return int(LIB.mg_ctx_num_inputs(self.handle))

    @property
    def num_shards(self):
        return int(LIB.mg_ctx_num_shards(self.handle))

    def _check_assignment(self, z, k=1):
        """The C side copies k*V*32 bytes from the buffer: a short array must never reach it."""
        z = np.ascontiguousarray(z, dtype=np.uint64)
        want = k * self.num_variables * 4
        if z.size != want:
            raise ValueError(f"assignment holds {z.size} u64 words, the context's circuit needs {want} ({k} x V = {self.num_variables} x 4)")
        return z

    def set_r1cs(self, r1cs: R1CS):
        ms = []
        for M in (r1cs.A, r1cs.B, r1cs.C):
            rp = np.ascontiguousarray(M.row_ptr, dtype=np.uint32)
            col = np.ascontiguousarray(M.col, dtype=np.uint32)
            val = _u64(M.val)
            ms.append((rp, col, val, _Csr(_p(rp), _p(col), _p(val), len(col))))
        _chk(LIB.mg_ctx_set_r1cs(self.handle, ctypes.byref(ms[0][3]), ctypes.byref(ms[1][3]), ctypes.byref(ms[2][3]),
                                 ctypes.c_uint64(r1cs.num_constraints)), "mg_ctx_set_r1cs")
        import weakref
        self._r1cs_ref = weakref.ref(r1cs)  # the uploaded matrices belong to exactly this R1CS object

    @property
    def domain_size(self):
        return LIB.mg_ctx_domain_size(self.handle)

    def table_bytes(self):
        """HBM bytes of the key tables: (bucket tables, full tables)"""
        v = (ctypes.c_uint64 * 2)()
        _chk(LIB.mg_ctx_table_bytes(self.handle, v), "mg_ctx_table_bytes")
        return int(v[0]), int(v[1])

    # ---- process-per-GPU sharding: partial results on the device, gathered by a collective, assembled on the host
    @property
    def partials_slot_limbs(self):
        return int(LIB.mg_partials_slot_limbs(self.handle))

    def partials_launch(self, zs, k, d_out_ptr, stream=None):
        """`mg_groth16_partials_launch`: this shard's five partial MSM results of k proofs -> device memory at d_out_ptr
        ([k][5][slot] u64); `stream` waits for them. Returns a handle for partials_finish()."""
        zs = self._check_assignment(zs, k)
        h = _vp()
        _chk(LIB.mg_groth16_partials_launch(self.handle, ctypes.c_uint64(k), _p(zs), _addr(d_out_ptr), _addr(stream),
                                            ctypes.byref(h)), "mg_groth16_partials_launch")
        return h

    @staticmethod
    def partials_finish(job):
        _chk(LIB.mg_groth16_partials_finish(job), "mg_groth16_partials_finish")

    def assemble(self, parts, rs, ss) -> list:
        """`mg_groth16_assemble`: parts [n_parts][k][5][slot] u64 (the gathered partial results), rs / ss k blinding
        scalars each -> k proofs (bytes)."""
        rs = np.ascontiguousarray(rs, dtype=np.uint64).reshape(-1, 4)
        ss = np.ascontiguousarray(ss, dtype=np.uint64).reshape(-1, 4)
        k = rs.shape[0]
        slot = self.partials_slot_limbs
        parts = np.ascontiguousarray(parts, dtype=np.uint64).reshape(-1, k, 5, slot)
        n = PROOF_BYTES[self.curve]
        out = ctypes.create_string_buffer(n * k)
        _chk(LIB.mg_groth16_assemble(self.handle, ctypes.c_uint64(k), int(parts.shape[0]), _p(parts), _p(rs), _p(ss), out),
             "mg_groth16_assemble")
        return [out.raw[i * n:(i + 1) * n] for i in range(k)]

    def witness_map(self, z):
        h = np.zeros((self.domain_size, 4), dtype=np.uint64)
        z = self._check_assignment(z)
        _chk(LIB.mg_witness_map(self.handle, _p(z), _p(h)), "mg_witness_map")
        return h

    def close(self):
        if self.handle is not None:
            LIB.mg_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


PROOF_BYTES = {BN254: 128, BLS12_381: 192}


class Groth16:
    """Mirror of `impl ProofSystem for Groth16<E>` (manta-crypto/src/arkworks/groth16.rs:548-610), prove only."""

    @staticmethod
    def prove(context: ProvingContext, compiler: R1CS, rng) -> bytes:
        """`rng` supplies the two blinding scalars exactly where ark-groth16's create_random_proof draws
        them: r = Fr::rand(rng); s = Fr::rand(rng) -- `rng()` must return one Montgomery Fr element
        (4 x u64) per call. Returns the arkworks canonical compressed proof bytes."""
        # the matrices on the device are those of the R1CS object last uploaded: another object -- even one with the
        # same (m, P) -- is another circuit and is uploaded afresh
        if context._r1cs_ref is None or context._r1cs_ref() is not compiler:
            context.set_r1cs(compiler)
        r = _u64(rng())
        s = _u64(rng())
        return Groth16.prove_with_randomness(context, compiler.z, r, s)

    @staticmethod
    def prove_with_randomness(context: ProvingContext, z, r, s) -> bytes:
        out = ctypes.create_string_buffer(PROOF_BYTES[context.curve])
        z = context._check_assignment(z)
        r, s = _u64(r).reshape(-1), _u64(s).reshape(-1)
        if r.size != 4 or s.size != 4:
            raise ValueError("r and s are one Fr element (4 x u64) each")
        _chk(LIB.mg_groth16_prove(context.handle, _p(z), _p(r), _p(s), out), "mg_groth16_prove")
        return out.raw

    @staticmethod
    def prove_batch(context: ProvingContext, zs, rs, ss) -> list:
        """k proofs of the context's circuit in one pass of the GPU pipeline (`mg_groth16_prove_batch`): zs = k
        assignments (k x V x 4 u64), rs / ss = k blinding scalars each. Returns k proofs, proof q byte-identical to
        prove_with_randomness(context, zs[q], rs[q], ss[q])."""
        zs = np.ascontiguousarray(zs, dtype=np.uint64)
        rs = np.ascontiguousarray(rs, dtype=np.uint64).reshape(-1, 4)
        ss = np.ascontiguousarray(ss, dtype=np.uint64).reshape(-1, 4)
        k = rs.shape[0]
        if ss.shape[0] != k or k == 0:
            raise ValueError("prove_batch: zs, rs, ss must describe the same number of proofs")
        zs = context._check_assignment(zs, k)
        n = PROOF_BYTES[context.curve]
        out = ctypes.create_string_buffer(n * k)
        _chk(LIB.mg_groth16_prove_batch(context.handle, ctypes.c_uint64(k), _p(zs), _p(rs), _p(ss), out),
             "mg_groth16_prove_batch")
        return [out.raw[i * n:(i + 1) * n] for i in range(k)]


def pairing_check(curve, g1_points, g2_points) -> bool:
    """prod_i e(P_i, Q_i) == 1 (`mg_pairing_check`): the test behind `PairingEngineExt::has_same` / `same_ratio`
    (manta-crypto/src/arkworks/pairing.rs:88-109). g1_points [n, 2 * limbs], g2_points [n, 4 * limbs] affine Montgomery."""
    g1 = _u64(g1_points).reshape(-1, affine_limbs(curve, 1))
    g2 = _u64(g2_points).reshape(-1, affine_limbs(curve, 2))
    if g1.shape[0] != g2.shape[0] or g1.shape[0] == 0:
        raise ValueError("pairing_check: as many G1 as G2 points, at least one")
    ok = ctypes.c_int(0)
    _chk(LIB.mg_pairing_check(curve, _p(g1), _p(g2), ctypes.c_size_t(g1.shape[0]), ctypes.byref(ok)), "mg_pairing_check")
    return bool(ok.value)


def proof_decode(curve, proof_bytes) -> np.ndarray:
    """Mirror of `Proof::deserialize`: arkworks compressed a | b | c -> affine Montgomery limbs (a | b | c), checked."""
    out = np.zeros(2 * affine_limbs(curve, 1) + affine_limbs(curve, 2), dtype=np.uint64)
    _chk(LIB.mg_proof_decode(curve, bytes(proof_bytes), _p(out)), "mg_proof_decode")
    return out


class VerifyingContext:
    """Mirror of groth16::VerifyingContext<E> (manta-crypto/src/arkworks/groth16.rs:305-539): the prepared verifying key,
    resident on the GPU."""

    def __init__(self, curve, vk):
        """`VerifyingContext::new(&vk)`: vk = object with alpha_g1, beta_g2, gamma_g2, delta_g2, gamma_abc_g1 (affine
        Montgomery limb arrays) -- e.g. a ProvingKey."""
        self.curve = curve
        arrs = [_u64(getattr(vk, k)) for k in ("alpha_g1", "beta_g2", "gamma_g2", "delta_g2", "gamma_abc_g1")]
        h = _vp()
        _chk(LIB.mg_vk_create(curve, *[_p(a) for a in arrs], ctypes.c_uint64(arrs[4].reshape(-1, affine_limbs(curve, 1)).shape[0]),
                              ctypes.byref(h)), "mg_vk_create")
        self.handle = h

    @classmethod
    def from_proving_context_key(cls, curve, pk):
        return cls(curve, pk)

    @classmethod
    def decode(cls, curve, data: bytes):
        self = cls.__new__(cls)
        self.curve = curve
        h = _vp()
        _chk(LIB.mg_vk_create_from_bytes(curve, bytes(data), _sz(len(data)), ctypes.byref(h)), "mg_vk_create_from_bytes")
        self.handle = h
        return self

    def encode(self) -> bytes:
        out = ctypes.create_string_buffer(LIB.mg_vk_encoded_size(self.handle))
        _chk(LIB.mg_vk_encode(self.handle, out), "mg_vk_encode")
        return out.raw

    @property
    def num_inputs(self):
        return int(LIB.mg_vk_num_inputs(self.handle))

    def alpha_g1_beta_g2(self) -> bytes:
        out = ctypes.create_string_buffer(12 * FQ_LIMBS[self.curve] * 8)
        _chk(LIB.mg_vk_alpha_beta(self.handle, out), "mg_vk_alpha_beta")
        return out.raw

    def close(self):
        if getattr(self, "handle", None) is not None:
            LIB.mg_vk_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _proof_points(curve, proof):
    return proof_decode(curve, proof) if isinstance(proof, (bytes, bytearray)) else _u64(proof)


def groth16_verify(context: VerifyingContext, inputs, proof) -> bool:
    """Mirror of `Groth16::verify(context, input, proof)` (groth16.rs:603-609): inputs = [P - 1, 4] Montgomery Fr,
    proof = the proof bytes (decoded and checked first) or its a | b | c limbs."""
    inp = _u64(inputs).reshape(-1, 4)
    if inp.shape[0] != context.num_inputs - 1:
        raise ValueError(f"{inp.shape[0]} public inputs, the key takes {context.num_inputs - 1}")
    ok = ctypes.c_int(0)
    pts = _proof_points(context.curve, proof)
    _chk(LIB.mg_groth16_verify(context.handle, _p(inp), _p(pts), ctypes.byref(ok)), "mg_groth16_verify")
    return bool(ok.value)


def groth16_verify_batch(context: VerifyingContext, inputs, proofs, rand128) -> bool:
    """k proofs of one key in one pass (`mg_groth16_verify_batch`): inputs [k, P - 1, 4], proofs = k proof byte strings
    (or [k, limbs] points), rand128 [k, 2] uint64: 128 random bits per proof (not both words zero; proof i enters with k1 + lambda k2, see mantagpu.h)."""
    k = len(proofs)
    inp = _u64(inputs).reshape(k, -1, 4)
    if inp.shape[1] != context.num_inputs - 1:
        raise ValueError("public-input count does not match the key")
    pts = np.stack([_proof_points(context.curve, p) for p in proofs])
    rnd = _u64(rand128).reshape(k, 2)
    ok = ctypes.c_int(0)
    _chk(LIB.mg_groth16_verify_batch(context.handle, ctypes.c_uint64(k), _p(inp), _p(_u64(pts)), _p(rnd), ctypes.byref(ok)),
         "mg_groth16_verify_batch")
    return bool(ok.value)
