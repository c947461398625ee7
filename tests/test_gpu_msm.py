"""GPU parity: Pippenger MSM (HIP, through the C ABI) vs the CPU oracle, bit-exact.
Mirrors how ark-groth16 calls VariableBaseMSM::multi_scalar_mul."""
import os

import numpy as np
import pytest

import helpers as H
import oracle_lib as O
from manta_rs_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu

CASES = [(0, 1), (1, 1), (0, 2), (1, 2)]


@pytest.mark.parametrize("curve,group", CASES)
@pytest.mark.parametrize("n", [1, 2, 31, 32, 100, 1000])
def test_msm_matches_oracle_uniform(gpu, curve, group, n):
    pts = H.random_points(curve, group, n, seed=100 + n)
    sc = synth.msm_scalars(curve, n, "U", seed=5 + n)
    want = O.msm(curve, group, pts, sc, algo=1)
    b = gpu.Bases(curve, group, pts)
    got = gpu.VariableBaseMSM.multi_scalar_mul(b, sc)
    assert (got == want).all()


@pytest.mark.parametrize("curve,group", CASES)
def test_msm_witness_like_and_infinity(gpu, curve, group):
    """0/1-heavy scalars (booleans dominate real witnesses), infinity bases (variables absent from B),
    repeated bases (P+P inside a bucket) and P + (-P)."""
    n = 600
    pts = H.random_points(curve, group, n, seed=77)
    pts[5] = 0
    pts[17] = 0          # infinity entries
    pts[40] = pts[41]    # equal points
    pts[50] = pts[51]
    sc = synth.msm_scalars(curve, n, "W", seed=9)
    sc[40] = sc[41] = np.array([1, 0, 0, 0], dtype=np.uint64)  # forces P+P in bucket "1"
    p = synth.FR_MODULUS[curve]
    sc[50] = synth.ints_to_limbs([5], 4)[0]
    sc[51] = synth.ints_to_limbs([p - 5], 4)[0]                # [5]P + [-5]P = 0
    want = O.msm(curve, group, pts, sc, algo=0)
    assert (want == O.msm(curve, group, pts, sc, algo=1)).all()
    got = gpu.VariableBaseMSM.multi_scalar_mul(gpu.Bases(curve, group, pts), sc)
    assert (got == want).all()


@pytest.mark.parametrize("curve,group", [(0, 1), (1, 1), (0, 2)])
@pytest.mark.parametrize("c", [7, 11])
def test_msm_precomputed_tables(gpu, curve, group, c):
    n = 700
    pts = H.random_points(curve, group, n, seed=31)
    pts[3] = 0
    sc = synth.msm_scalars(curve, n, "W", seed=3)
    want = O.msm(curve, group, pts, sc, algo=1)
    got = gpu.VariableBaseMSM.multi_scalar_mul(gpu.Bases(curve, group, pts, precompute_window_bits=c), sc)
    assert (got == want).all()


@pytest.mark.parametrize("curve,group,pre", [(1, 1, 17), (1, 1, 15), (1, 1, 5), (1, 2, 17), (1, 1, 0), (0, 1, 11), (0, 1, 0), (0, 2, 7)])
def test_msm_scalars_around_half_the_group_order(gpu, curve, group, pre):
    """The digit kernel recodes k > r / 2 as -(r - k), which is what lets a window width that divides the scalar's bit length
    (BLS12-381: 255 = 15 x 17) do with one window fewer: the scalars where that folding switches, the largest ones, and the ones
    whose top window is full, mixed into uniform ones -- against the oracle, table widths that divide 255 and ones that do not."""
    n = 400
    r = synth.FR_MODULUS[curve]
    pts = H.random_points(curve, group, n, seed=55)
    ks = synth.limbs_to_ints(synth.msm_scalars(curve, n, "U", seed=56))
    bits = r.bit_length()
    edge = [0, 1, 2, (r - 1) // 2 - 1, (r - 1) // 2, (r + 1) // 2, (r + 1) // 2 + 1, r - 2, r - 1, (1 << (bits - 1)) - 1, 1 << (bits - 1),
            (1 << (bits - 1)) + 1, (1 << (bits - 2)) - 1, 1 << (bits - 2), r - (1 << (bits - 2)), r >> 1, (r >> 1) ^ ((1 << 200) - 1)]
    edge = [e % r for e in edge]
    for j, e in enumerate(edge):
        ks[7 * j + 3] = e
    sc = synth.ints_to_limbs(ks, 4)
    want = O.msm(curve, group, pts, sc, algo=1)
    got = gpu.VariableBaseMSM.multi_scalar_mul(gpu.Bases(curve, group, pts, precompute_window_bits=pre), sc)
    assert (got == want).all()


@pytest.mark.parametrize("curve,group,c", [(0, 1, 8), (0, 1, 3), (0, 2, 6), (1, 1, 5), (1, 1, 7), (1, 2, 4), (0, 1, 2)])
def test_msm_full_tables(gpu, curve, group, c):
    """precompute_window_bits = -c: every multiple m 2^(cw) P of every window is tabulated, a signed digit addresses its
    summand, the MSM is one plain sum (no buckets, no sort, no bucket reduce). Against the oracle: uniform and witness-like
    scalars, infinity bases, repeated bases, P + (-P), the scalars around r / 2 and the largest ones, a digit equal to 2^(c-1)."""
    n = 700
    r = synth.FR_MODULUS[curve]
    pts = H.random_points(curve, group, n, seed=61)
    pts[3] = 0
    pts[44] = 0
    pts[40] = pts[41]
    b = gpu.Bases(curve, group, pts, precompute_window_bits=-c)
    for dist in ("U", "W"):
        ks = synth.limbs_to_ints(synth.msm_scalars(curve, n, dist, seed=62))
        edge = [0, 1, (r - 1) // 2, (r + 1) // 2, r - 1, r - 2, 1 << (c - 1), (1 << (c - 1)) + 1, (1 << c) - 1, 1 << c,
                ((1 << (c - 1)) << c) | (1 << (c - 1)), r - (1 << (c - 1))]
        for j, e in enumerate(edge):
            ks[11 * j + 5] = e % r
        ks[40], ks[41] = 5, r - 5
        sc = synth.ints_to_limbs(ks, 4)
        want = O.msm(curve, group, pts, sc, algo=1)
        assert (gpu.VariableBaseMSM.multi_scalar_mul(b, sc) == want).all(), dist
        dsc = gpu.DeviceBuffer.from_numpy(sc)
        assert (gpu.VariableBaseMSM.launch(b, dsc, n, sparse=True).finish() == want).all(), dist
    z = np.zeros((n, 4), dtype=np.uint64)
    assert not gpu.VariableBaseMSM.multi_scalar_mul(b, z).any()


def test_msm_all_zero_and_all_one(gpu):
    n = 300
    pts = H.random_points(0, 1, n, seed=2)
    b = gpu.Bases(0, 1, pts)
    z = np.zeros((n, 4), dtype=np.uint64)
    assert not gpu.VariableBaseMSM.multi_scalar_mul(b, z).any()          # result = infinity
    one = z.copy()
    one[:, 0] = 1
    assert (gpu.VariableBaseMSM.multi_scalar_mul(b, one) == O.g_sum(0, 1, pts)).all()


def test_msm_zips_to_shorter(gpu):
    pts = H.random_points(0, 1, 50, seed=4)
    sc = synth.msm_scalars(0, 80, "U", seed=4)
    got = gpu.VariableBaseMSM.multi_scalar_mul(gpu.Bases(0, 1, pts), sc)
    assert (got == O.msm(0, 1, pts, sc[:50], algo=1)).all()


@pytest.mark.parametrize("curve", [0, 1])
def test_msm_medium_closed_form(gpu, curve):
    """n = 2^15: bases in arithmetic progression P_i = [s0 + i*s1]G built on the GPU, so the expected
    result is [sum k_i (s0 + i s1) mod r]G -- one scalar multiplication."""
    n = 1 << 15
    p = synth.FR_MODULUS[curve]
    s0, s1 = 0x1234567, 0x89abcdef1
    ks = synth.ints_to_limbs([(s0 + i * s1) % p for i in range(n)], 4)
    G = O.generator(curve, 1)
    dks = gpu.DeviceBuffer.from_numpy(ks)
    dpts = gpu.fixed_base_mul(curve, 1, G, dks, n)
    pts = dpts.to_numpy(shape=(n, gpu.affine_limbs(curve, 1)))
    assert (pts[12345] == O.g_mul(curve, 1, G, ks[12345])).all()
    sc = synth.msm_scalars(curve, n, "W", seed=21)
    sci = synth.limbs_to_ints(sc)
    expect_k = sum(k * ((s0 + i * s1) % p) for i, k in enumerate(sci)) % p
    want = O.g_mul(curve, 1, G, synth.ints_to_limbs([expect_k], 4)[0])
    for pre in (0, 13):
        b = gpu.Bases(curve, 1, (dpts.ptr, n), precompute_window_bits=pre, on_device=True)
        dsc = gpu.DeviceBuffer.from_numpy(sc)
        got = gpu.VariableBaseMSM.launch(b, dsc, n).finish()
        assert (got == want).all(), pre


@pytest.mark.parametrize("curve,pre", [(1, 16), (1, 17), (0, 16), (1, 0)])
def test_msm_full_size_2_20_closed_form(gpu, curve, pre):
    """BASELINE size (n = 2^20), property check that needs no O(n) oracle run: bases P_i = [s0 + i s1]G built
    by the library's fixed-base batch multiply, so sum k_i P_i = [sum k_i (s0 + i s1) mod r] G. Run for uniform
    and witness-like scalars; also linearity: MSM(k) + MSM(k') = MSM(k + k')."""
    n = 1 << 20
    p = synth.FR_MODULUS[curve]
    s0, s1 = 0x1234567, 0x89abcdef1
    kb = np.zeros((n, 4), dtype=np.uint64)
    idx = np.arange(n, dtype=np.uint64)
    kb[:, 0] = np.uint64(s0) + idx * np.uint64(s1)          # < 2^64 for n = 2^20: no reduction needed
    G = O.generator(curve, 1)
    dpts = gpu.fixed_base_mul(curve, 1, G, gpu.DeviceBuffer.from_numpy(kb), n)
    b = gpu.Bases(curve, 1, (dpts.ptr, n), precompute_window_bits=pre, on_device=True)
    base_k = [s0 + i * s1 for i in range(n)]

    def expect(sc):
        t = sum(k * bk for k, bk in zip(synth.limbs_to_ints(sc), base_k)) % p
        return O.g_mul(curve, 1, G, synth.ints_to_limbs([t], 4)[0])

    res = {}
    for dist in ("U", "W"):
        sc = synth.msm_scalars(curve, n, dist, seed=33)
        got = gpu.VariableBaseMSM.launch(b, gpu.DeviceBuffer.from_numpy(sc), n).finish()
        assert (got == expect(sc)).all(), dist
        res[dist] = (sc, got)
    # linearity: scalars U + W (mod r) -- sum of the two results
    su = synth.limbs_to_ints(res["U"][0])
    sw = synth.limbs_to_ints(res["W"][0])
    ssum = synth.ints_to_limbs([(a + c) % p for a, c in zip(su, sw)], 4)
    got = gpu.VariableBaseMSM.launch(b, gpu.DeviceBuffer.from_numpy(ssum), n).finish()
    assert (got == O.g_add(curve, 1, res["U"][1], res["W"][1])).all()


@pytest.mark.parametrize("curve,group", CASES)
@pytest.mark.parametrize("pre", [0, 9])
def test_msm_sparse_hint_compacts_zero_digits(gpu, curve, group, pre):
    """MG_SCALARS_SPARSE: the digit kernel drops zero digits before the sort (what the prover does for the three
    witness MSMs). Same result as the oracle for witness-like, all-zero, single-nonzero and uniform scalars, with
    plain bases and precomputed tables, and more bases than scalars."""
    n = 3000
    pts = H.random_points(curve, group, n, seed=41)
    pts[7] = 0
    b = gpu.Bases(curve, group, pts, precompute_window_bits=pre)
    cases = {"W": synth.msm_scalars(curve, n, "W", seed=12), "U": synth.msm_scalars(curve, n, "U", seed=13)}
    z = np.zeros((n, 4), dtype=np.uint64)
    cases["zero"] = z
    one = z.copy()
    one[1234, 0] = 5
    cases["single"] = one
    for name, sc in cases.items():
        d = gpu.DeviceBuffer.from_numpy(sc)
        got = gpu.VariableBaseMSM.launch(b, d, n, sparse=True).finish()
        assert (got == O.msm(curve, group, pts, sc, algo=1)).all(), name
    m = 1777  # fewer scalars than bases: zip to the shorter
    sc = cases["W"][:m]
    got = gpu.VariableBaseMSM.launch(b, gpu.DeviceBuffer.from_numpy(sc), m, sparse=True).finish()
    assert (got == O.msm(curve, group, pts[:m], sc, algo=1)).all()


_FRONT_SCRIPT = r'''
import os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import numpy as np
import oracle_lib as O
from manta_rs_amd import api as gpu, synth
gpu.init(0)
curve, n = {curve}, 1 << 15
p = synth.FR_MODULUS[curve]
s0, s1 = 0x1234567, 0x89abcdef1
ks = synth.ints_to_limbs([(s0 + i * s1) % p for i in range(n)], 4)
G = O.generator(curve, {group})
dpts = gpu.fixed_base_mul(curve, {group}, G, gpu.DeviceBuffer.from_numpy(ks), n)
for dist in ("U", "W"):
    sc = synth.msm_scalars(curve, n, dist, seed=77)
    t = sum(k * ((s0 + i * s1) % p) for i, k in enumerate(synth.limbs_to_ints(sc))) % p
    want = O.g_mul(curve, {group}, G, synth.ints_to_limbs([t], 4)[0])
    for pre in {pres}:
        b = gpu.Bases(curve, {group}, (dpts.ptr, n), precompute_window_bits=pre, on_device=True)
        got = gpu.VariableBaseMSM.launch(b, gpu.DeviceBuffer.from_numpy(sc), n, sparse=(dist == "W")).finish()
        assert (got == want).all(), (dist, pre)
        b.close()
print("front levels ok")
'''


@pytest.mark.parametrize("curve,group,pres,env", [
    (1, 1, (16, 20), {"MANTA_RED_S": "3"}),                                              # one / two-three front levels, side stream
    (1, 1, (14, 18), {"MANTA_RED_S": "2", "MANTA_RED_MIN": "1024", "MANTA_RED_SIDE": "0", "MANTA_RED_SP": "2"}),  # many levels, inline
    (0, 1, (16,), {"MANTA_RED_S": "3", "MANTA_RED_S0": "3"}),
    (0, 2, (15,), {"MANTA_RED_S": "3", "MANTA_RED_MIN": "2048"}),                        # G2: the additions are calls
    (0, 1, (9, 13), {"MANTA_RED_MIN": "128"}),                                            # the lowest threshold the knob takes
    (1, 1, (12, 16), {"MANTA_RED_MIN": "1024"}),
    (0, 1, (13, 16), {"MANTA_RED_MIN": "16384"}),                                         # the default threshold, stated
])
def test_msm_reduce_front_levels(gpu, curve, group, pres, env):
    """The optional work-efficient front levels of the bucket reduce (serial_reduce / serial_reduce_coop, msm_impl.h; off by
    default, knobs are read once per process, hence the child process): closed-form check at n = 2^15 for window widths
    whose bucket windows are long enough to take one to several levels, uniform and witness-like scalars."""
    import subprocess
    import sys
    code = _FRONT_SCRIPT.format(root=ROOT, curve=curve, group=group, pres=pres)
    out = subprocess.run([sys.executable, "-c", code], env=H.knob_env(env), capture_output=True, text=True, timeout=900)  # (the diagnosis twin)
    assert out.returncode == 0 and "front levels ok" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


@pytest.mark.parametrize("curve,group,n,pre", [(0, 1, 300, 5), (0, 1, 4000, 9), (0, 1, 3000, -6), (0, 2, 500, -4), (1, 1, 20000, 13), (1, 1, 1 << 15, 16), (0, 2, 900, 6),
                                                (1, 2, 2500, 8)])
def test_msm_result_folded_on_the_device(gpu, curve, group, n, pre):
    """`mg_msm_result_to_device`: the host fold of mg_msm_finish done by one kernel behind the MSM -- every staging layout
    (one tile per window, (X, sumS) pairs, the general two-level one) -- leaves an XYZZ point in device memory that
    normalises to the same affine point; plain bases are refused (their Horner chain is a host job)."""
    pts = H.random_points(curve, group, min(n, 1500), seed=51)
    pts = np.concatenate([pts] * (-(-n // pts.shape[0])))[:n]
    sc = synth.msm_scalars(curve, n, "W", seed=52)
    b = gpu.Bases(curve, group, pts, precompute_window_bits=pre)
    want = gpu.VariableBaseMSM.multi_scalar_mul(b, sc)
    d = gpu.DeviceBuffer.from_numpy(sc)
    out = gpu.DeviceBuffer(gpu.xyzz_limbs(curve, group) * 8)
    job = gpu.VariableBaseMSM.launch(b, d, n, sparse=True)
    job.result_to_device(out.ptr)
    job.release()
    got = gpu.xyzz_sum(curve, group, out.to_numpy(shape=(1, gpu.xyzz_limbs(curve, group))))
    assert (got == want).all()
    plain = gpu.VariableBaseMSM.launch(gpu.Bases(curve, group, pts), d, n)
    with pytest.raises(gpu.MantaGpuError):
        plain.result_to_device(out.ptr)
    assert (plain.finish() == want).all()


@pytest.mark.parametrize("min_items", [128, 1024, 16384])
def test_front_levels_inside_captured_passes(gpu, min_items):
    """VERDICT r5 item 5. Round 5 saw "the pass fails" with the front levels inside the captured graphs of a batched pass and
    MANTA_RED_MIN=1024, deleted the knob and kept the restriction. Root cause (round 6, profiles/r06_front_levels_in_graph.txt): the
    engine's ONE side stream joined the forked capture from several branches, the runtime's per-stream lists of parallel capture
    streams became cyclic, and hipStreamEndCapture recursed until the stack was gone. The side stream no longer joins any capture;
    with that the front levels run INSIDE the slot's graphs wherever a window is long enough, and passes of 8 proofs (diagnosis twin)
    are the oracle's -- eager, eager, capture, replay, replay -- for thresholds 128 / 1 024 (front levels in every MSM of the
    pass) and 16 384 (the default: none qualifies), in the forked topology and in the split one."""
    import subprocess
    import sys
    for extra in ({}, {"MANTA_GRAPH_BATCH": "split"}):
        env = H.knob_env(dict({"MANTA_RED_MIN": str(min_items)}, **extra), strip_prefix="MANTA_")
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "diag_front_in_graph.py"), "8", "to_public"], env=env, capture_output=True,
                             text=True, timeout=900)
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
        ok = [ln for ln in out.stdout.splitlines() if ln.endswith("8 proofs == oracle")]
        assert len(ok) == 5 and "libmantagpu_diag.so" in out.stdout, out.stdout[-2000:]
