"""Python mirror of rust/capture/src/lib.rs: the container of one captured proof -- (A, B, C, z, r, s, proving key, proof
bytes) as they cross the C ABI on the prove path. `tests/golden/arkworks/*.bin` are meant to be written by the Rust capture
(arkworks on the CPU, the reference itself); the same writer here lets the oracle produce files in the identical format so
that the consumer (tests/test_gpu_reference_fixture.py) is exercised in this container, where no Rust toolchain exists."""
import struct
from types import SimpleNamespace

import numpy as np

MAGIC = b"MGFX0001"
SECTIONS = "magic curve m num_instance num_variables a b c z r s proving_key proof"


def _put_matrix(M):
    row_ptr = np.ascontiguousarray(M.row_ptr, dtype="<u4")
    col = np.ascontiguousarray(M.col, dtype="<u4")
    val = np.ascontiguousarray(M.val, dtype="<u8").reshape(-1, 4)
    assert row_ptr[-1] == len(col) == val.shape[0]
    return struct.pack("<Q", len(col)) + row_ptr.tobytes() + col.tobytes() + val.tobytes()


def encode(curve, A, B, C, m, P, z, r, s, pk_bytes, proof_bytes) -> bytes:
    z = np.ascontiguousarray(z, dtype="<u8").reshape(-1, 4)
    out = [MAGIC, struct.pack("<IIQQQ", curve, 0, m, P, z.shape[0]), _put_matrix(A), _put_matrix(B), _put_matrix(C), z.tobytes(),
           np.ascontiguousarray(r, dtype="<u8").reshape(4).tobytes(), np.ascontiguousarray(s, dtype="<u8").reshape(4).tobytes(),
           struct.pack("<Q", len(pk_bytes)), bytes(pk_bytes), struct.pack("<Q", len(proof_bytes)), bytes(proof_bytes)]
    return b"".join(out)


def decode(data: bytes):
    if data[:8] != MAGIC:
        raise ValueError("not a captured-proof fixture")
    at = 8
    curve, _, m, P, V = struct.unpack_from("<IIQQQ", data, at)
    at += struct.calcsize("<IIQQQ")
    mats = []
    for _ in range(3):
        (nnz,) = struct.unpack_from("<Q", data, at)
        at += 8
        row_ptr = np.frombuffer(data, dtype="<u4", count=m + 1, offset=at).astype(np.uint32)
        at += 4 * (m + 1)
        col = np.frombuffer(data, dtype="<u4", count=nnz, offset=at).astype(np.uint32)
        at += 4 * nnz
        val = np.frombuffer(data, dtype="<u8", count=4 * nnz, offset=at).astype(np.uint64).reshape(nnz, 4)
        at += 32 * nnz
        if row_ptr[0] != 0 or row_ptr[-1] != nnz:
            raise ValueError("inconsistent matrix section")
        mats.append(SimpleNamespace(row_ptr=row_ptr, col=col, val=val))
    z = np.frombuffer(data, dtype="<u8", count=4 * V, offset=at).astype(np.uint64).reshape(V, 4)
    at += 32 * V
    r = np.frombuffer(data, dtype="<u8", count=4, offset=at).astype(np.uint64)
    s = np.frombuffer(data, dtype="<u8", count=4, offset=at + 32).astype(np.uint64)
    at += 64
    (n,) = struct.unpack_from("<Q", data, at)
    pk = data[at + 8:at + 8 + n]
    at += 8 + n
    (n,) = struct.unpack_from("<Q", data, at)
    proof = data[at + 8:at + 8 + n]
    at += 8 + n
    if at != len(data) or len(pk) == 0 or len(proof) == 0:
        raise ValueError("truncated or oversized fixture")
    return SimpleNamespace(curve=curve, m=m, P=P, V=V, A=mats[0], B=mats[1], C=mats[2], z=z, r=r, s=s, pk_bytes=pk, proof=proof)


def pk_bytes(O, curve, pk) -> bytes:
    """arkworks 0.3 `ProvingKey::serialize_unchecked` layout of an oracle-side key"""
    ser1 = lambda p: O.serialize(curve, 1, p, compressed=False)  # noqa: E731
    ser2 = lambda p: O.serialize(curve, 2, p, compressed=False)  # noqa: E731
    vec = lambda pts, f: struct.pack("<Q", len(pts)) + b"".join(f(p) for p in pts)  # noqa: E731
    return (ser1(pk.alpha_g1[0]) + ser2(pk.beta_g2[0]) + ser2(pk.gamma_g2[0]) + ser2(pk.delta_g2[0]) +
            vec(pk.gamma_abc_g1, ser1) + ser1(pk.beta_g1[0]) + ser1(pk.delta_g1[0]) + vec(pk.a_query, ser1) +
            vec(pk.b_g1_query, ser1) + vec(pk.b_g2_query, ser2) + vec(pk.h_query, ser1) + vec(pk.l_query, ser1))
