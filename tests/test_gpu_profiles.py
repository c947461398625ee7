"""Whole-proof parity on witnesses of STATED density.

The reference proves the assignment `Transfer::known_constraints` synthesises (manta-accounting/src/transfer/mod.rs:667-673,
driven by manta-pay/src/test/payment.rs:222-273): Poseidon states and in-circuit curve coordinates are dense field elements,
bits are booleans. No such witness can be captured here, so synth offers three profiles with the density measured on the
resulting z (synth.histogram): "sparse" (rounds 1-3: 69 % zeros / 23 % ones), "W" (40 / 25 / 10 / 25) and
"dense" (config 1: a multiplication chain, no trivial scalar). The pair-count dependent paths of the MSMs -- digit compaction,
the chunk length derived on the device, merge sizing, full tables against bucket tables -- see 4x (W) and 11x (dense) the
digit pairs of the sparse profile; every case below is byte-compared with the CPU oracle through the C ABI."""
import numpy as np
import pytest

import helpers as H
import oracle_lib as O
from manta_rs_amd import keygen, synth

pytestmark = pytest.mark.gpu


def _fast_oracle():
    O.set_threads(O.usable_cpus())  # the dense PrivateTransfer proof is ~4 s on one core


@pytest.fixture(scope="module")
def pt_keys():
    """one key per profile for the PrivateTransfer shape (the matrices differ per profile, so do the keys)"""
    out = {}
    for prof in ("W", "dense"):
        c = synth.make_shape(0, "private_transfer", profile=prof)
        out[prof] = (c, keygen.generate(c, synth.from_mont(H.toxic(0, seed=40 + len(prof)), synth.FR_MODULUS[0])))
    return out


def test_profiles_have_the_stated_density():
    for shape in ("to_private", "private_transfer"):
        w = synth.histogram(synth.make_shape(0, shape, profile="W").z_int)
        assert abs(w["zero"] - 0.40) < 2e-3 and abs(w["one"] - 0.25) < 2e-3 and abs(w["small"] - 0.10) < 2e-3 and abs(w["dense"] - 0.25) < 2e-3
        d = synth.histogram(synth.make_shape(0, shape, profile="dense").z_int)
        assert d["zero"] == 0 and d["small"] == 0 and d["one"] * d["n"] == 1  # z_0 = 1 is the only trivial scalar
    s = synth.histogram(synth.make_shape(0, "private_transfer").z_int)
    assert s["zero"] > 0.6  # what rounds 1-3 measured on


@pytest.mark.parametrize("prof", ["W", "dense"])
def test_private_transfer_single_proof_on_full_and_bucket_tables(gpu, pt_keys, prof):
    """single proofs: on the context's full tables (default budget) and with none (`full_table_bytes = 0`: the bucket
    tables, sort + bucket reduce on the chain); eager runs, graph capture and replay all give the oracle's bytes"""
    _fast_oracle()
    c, pk = pt_keys[prof]
    assert synth.check_satisfied(c)
    rs = H.rand_fr_mont(0, 2, seed=77)
    want = O.groth16_prove(c, pk, rs[0], rs[1])
    assert O.groth16_verify(0, pk, c.z[1:c.P], want) == 1
    for budget in (None, 0):
        ctx = gpu.ProvingContext(0, pk, full_table_bytes=budget)
        ctx.set_r1cs(gpu.R1CS.from_circuit(c))
        tb = ctx.table_bytes()
        assert (tb[1] == 0) == (budget == 0)
        for _ in range(4):  # two eager runs size the buffers, the third captures, the fourth replays
            assert gpu.Groth16.prove_with_randomness(ctx, c.z, rs[0], rs[1]) == want, (prof, budget)
        ctx.close()


@pytest.mark.parametrize("prof", ["W", "dense"])
def test_private_transfer_batch_of_32_distinct_assignments(gpu, pt_keys, prof):
    """one pass of 32 proofs, 32 distinct assignments of the profile (Reassigner keeps the density); every member pairing-checked,
    five byte-compared with the oracle; a pass of 5 coalesced-size proofs (narrow tables) as well"""
    _fast_oracle()
    c, pk = pt_keys[prof]
    R = synth.Reassigner(c)
    k = 32
    cs = [c] + [R.assign(0x4D414E5441_2000 + q) for q in range(1, k)]
    h = synth.histogram(cs[7].z_int)
    if prof == "W":
        assert abs(h["zero"] - 0.40) < 0.01 and abs(h["one"] - 0.25) < 0.01 and abs(h["small"] - 0.10) < 0.01
    else:
        assert h["zero"] == 0
    rs = H.rand_fr_mont(0, 2 * k, seed=91)
    ctx = gpu.ProvingContext(0, pk)
    ctx.set_r1cs(gpu.R1CS.from_circuit(c))
    zs = np.stack([x.z for x in cs])
    for rep in range(3):
        got = gpu.Groth16.prove_batch(ctx, zs, rs[:k], rs[k:])
        if rep == 0:
            for q in range(k):
                assert O.groth16_verify(0, pk, cs[q].z[1:c.P], got[q]) == 1, q
            for q in (0, 1, 13, 30, 31):
                assert got[q] == O.groth16_prove(cs[q], pk, rs[q], rs[k + q]), q
            first = got
        assert got == first
    got5 = gpu.Groth16.prove_batch(ctx, zs[3:8], rs[3:8], rs[k + 3:k + 8])
    assert got5 == first[3:8]
    ctx.close()


@pytest.mark.parametrize("prof", ["W", "dense"])
def test_private_transfer_sharded_8_ways(gpu, pt_keys, prof):
    """BASELINE configs[3] on the profile: every MSM range-sharded over 8 device entries (device 0 eight times: one GPU per
    box), partial points summed; bytes equal the oracle's and the single-device context's"""
    _fast_oracle()
    c, pk = pt_keys[prof]
    rs = H.rand_fr_mont(0, 4, seed=93)
    ctx = gpu.ProvingContext(0, pk, devices=[0] * 8)
    ctx.set_r1cs(gpu.R1CS.from_circuit(c))
    assert ctx.num_shards == 8
    p0 = gpu.Groth16.prove_with_randomness(ctx, c.z, rs[0], rs[1])
    assert p0 == O.groth16_prove(c, pk, rs[0], rs[1])
    got = gpu.Groth16.prove_batch(ctx, np.stack([c.z, c.z]), rs[0:4:2], rs[1:4:2])
    assert got[0] == p0 and got[1] == O.groth16_prove(c, pk, rs[2], rs[3])
    ctx.close()


@pytest.mark.parametrize("prof", ["W", "dense"])
def test_bls12_381_2_15_circuit(gpu, prof):
    """the 2^15 BLS12-381 circuit (D = V = 2^15, P = 16: the small sibling of BASELINE configs[2]) on the profile: single
    proof and a batch of 3 against the oracle"""
    _fast_oracle()
    curve, D, P = 1, 1 << 15, 16
    c = synth.make_circuit(curve, D - P, D, P, seed=0x4D414E5441_0315, profile=prof)
    pk = keygen.generate(c, synth.from_mont(H.toxic(curve, seed=52), synth.FR_MODULUS[curve]))
    ctx = gpu.ProvingContext(curve, pk)
    ctx.set_r1cs(gpu.R1CS.from_circuit(c))
    rs = H.rand_fr_mont(curve, 6, seed=95)
    want = O.groth16_prove(c, pk, rs[0], rs[1])
    for _ in range(4):
        assert gpu.Groth16.prove_with_randomness(ctx, c.z, rs[0], rs[1]) == want
    assert O.groth16_verify(curve, pk, c.z[1:c.P], want) == 1
    R = synth.Reassigner(c)
    cs = [c, R.assign(5), R.assign(6)]
    got = gpu.Groth16.prove_batch(ctx, np.stack([x.z for x in cs]), rs[0:6:2], rs[1:6:2])
    assert got[0] == want
    for q in (1, 2):
        assert got[q] == O.groth16_prove(cs[q], pk, rs[2 * q], rs[2 * q + 1])
    ctx.close()


# ---- tuning: every field of mg_tuning and every environment variable the shipped library reads ---------------
_TUNING_CASES = {  # one non-default value per variable of mg_tuning_env_names (MANTA_RCCL_LIB is a path, not a schedule)
    "MANTA_GRAPH": ("split", "off"), "MANTA_GRAPH_BATCH": ("off",), "MANTA_PROVE_STREAMS": ("3",), "MANTA_Z3_LINEAR": ("0",),
    "MANTA_COALESCE": ("0",), "MANTA_COALESCE_GATHER_US": ("0",), "MANTA_BATCH_INFLIGHT": ("1",), "MANTA_QUEUE_AWARE": ("0",),
    "MANTA_MSM_DEDICATED_QUEUES": ("0", "2"), "MANTA_PROVE_C": ("7",), "MANTA_PROVE_CW": ("9",), "MANTA_PROVE_CH": ("10",),
    "MANTA_PROVE_CG2": ("8",), "MANTA_FULL_TABLE_GB": ("0.5",),
}
_TUNING_SCRIPT = '''
import os, sys, threading
sys.path.insert(0, r"{root}"); sys.path.insert(0, os.path.join(r"{root}", "tests"))
import numpy as np
import helpers as H
from manta_rs_amd import api as gpu, synth, keygen
gpu.init(0)
assert gpu.LIB_PATH.endswith("libmantagpu.so"), gpu.LIB_PATH     # the SHIPPED library
print("TUNING", sorted(gpu.get_tuning().as_dict().items()))
c = synth.make_shape(0, "to_public", profile="W")
pk = keygen.generate(c, synth.from_mont(H.toxic(0, seed=6), synth.FR_MODULUS[0]))
ctx = gpu.ProvingContext(0, pk)
ctx.set_r1cs(gpu.R1CS.from_circuit(c))
rs = H.rand_fr_mont(0, 4, seed=98)
rs[2][:] = 0
z2 = synth.Reassigner(c).assign(0x5EED).z
for i in range(5):
    print("PROOF", gpu.Groth16.prove_with_randomness(ctx, z2 if i & 1 else c.z, rs[2 * (i & 1)], rs[2 * (i & 1) + 1]).hex())
for rep in range(4):
    got = gpu.Groth16.prove_batch(ctx, np.stack([c.z, z2] * 4), np.stack([rs[0], rs[2]] * 4), np.stack([rs[1], rs[3]] * 4))
    print("BATCH", " ".join(g.hex() for g in got))
big = gpu.Groth16.prove_batch(ctx, np.stack([c.z, z2] * 40), np.stack([rs[0], rs[2]] * 40), np.stack([rs[1], rs[3]] * 40))  # three passes
print("BIG", len(big), len(set(big[0::2])), len(set(big[1::2])), big[0].hex(), big[1].hex())
def worker(t):
    for i in range(10):
        j = (i + t) & 1
        print("CONC %d %s" % (j, gpu.Groth16.prove_with_randomness(ctx, z2 if j else c.z, rs[2 * j], rs[2 * j + 1]).hex()), flush=True)
ts = [threading.Thread(target=worker, args=(t,)) for t in range(3)]
[t.start() for t in ts]
[t.join() for t in ts]
# a stand-alone MSM, three in flight (MANTA_MSM_DEDICATED_QUEUES)
pts = H.random_points(0, 1, 1500, seed=3)
pts = np.concatenate([pts] * 4)
sc = synth.msm_scalars(0, pts.shape[0], "U", seed=4)
b = gpu.Bases(0, 1, pts, precompute_window_bits=11)
d = gpu.DeviceBuffer.from_numpy(sc)
jobs = [gpu.VariableBaseMSM.launch(b, d, pts.shape[0]) for _ in range(3)]
print("MSM", " ".join(bytes(j.finish()).hex() for j in jobs))
'''


def _tuning_truth():
    c = synth.make_shape(0, "to_public", profile="W")
    pk = keygen.generate(c, synth.from_mont(H.toxic(0, seed=6), synth.FR_MODULUS[0]))
    rs = H.rand_fr_mont(0, 4, seed=98)
    rs[2][:] = 0
    z2 = synth.Reassigner(c).assign(0x5EED).z
    _fast_oracle()
    two = [O.groth16_prove(c, pk, rs[0], rs[1]).hex(), O.groth16_prove(c, pk, rs[2], rs[3], z=z2).hex()]
    pts = np.concatenate([H.random_points(0, 1, 1500, seed=3)] * 4)
    msm = bytes(O.msm(0, 1, pts, synth.msm_scalars(0, pts.shape[0], "U", seed=4), algo=1)).hex()
    return c, pk, rs, z2, two, msm


def _check_tuning_child(out, two, msm, what):
    assert out.returncode == 0, (what, out.stdout[-2000:] + out.stderr[-2000:])
    lines = out.stdout.split("\n")
    assert [ln.split()[1] for ln in lines if ln.startswith("PROOF")] == [two[i & 1] for i in range(5)], what
    batches = [ln[6:] for ln in lines if ln.startswith("BATCH")]
    assert len(batches) == 4 and all(b == " ".join(two * 4) for b in batches), what
    big = [ln.split() for ln in lines if ln.startswith("BIG")]
    assert big == [["BIG", "80", "1", "1", two[0], two[1]]], what
    conc = [ln.split() for ln in lines if ln.startswith("CONC")]
    assert len(conc) == 30 and all(two[int(j)] == h for _, j, h in conc), what
    assert [ln for ln in lines if ln.startswith("MSM")] == ["MSM " + " ".join([msm] * 3)], what
    return [ln for ln in lines if ln.startswith("TUNING")][0]


def test_every_environment_variable_of_the_shipped_library_leaves_results_unchanged(gpu):
    """the shipped library reads the variables mg_tuning_env_names() lists and no other (tests/test_host.py checks
    the binary). Each one, set to a non-default value in a process of its own (the table is read once), must leave every result what
    the oracle says: single proofs (eager, captured, replayed; r = 0; two assignments), passes of 8, a batch of 80 streamed as three
    passes, single calls from three host threads at once (coalesced), and three stand-alone MSMs in flight. The child also prints
    the tuning in force, so a variable that the table silently ignored fails here."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    names = [n for n in gpu.tuning_env_names() if n != "MANTA_RCCL_LIB"]
    assert sorted(names) == sorted(_TUNING_CASES), "a variable was added to the library's table without a case here"
    c, pk, rs, z2, two, msm = _tuning_truth()
    base = {k: v for k, v in os.environ.items() if not k.startswith("MANTA_")}
    default = _check_tuning_child(subprocess.run([sys.executable, "-c", _TUNING_SCRIPT.format(root=root)], env=base, capture_output=True, text=True,
                                                 timeout=900), two, msm, "defaults")
    for name in names:
        for value in _TUNING_CASES[name]:
            out = subprocess.run([sys.executable, "-c", _TUNING_SCRIPT.format(root=root)], env=dict(base, **{name: value}), capture_output=True,
                                 text=True, timeout=900)
            assert _check_tuning_child(out, two, msm, (name, value)) != default, (name, value, "the variable did not reach the tuning")


def test_per_context_tuning_through_the_abi_leaves_results_unchanged(gpu):
    """mg_ctx_opts.tuning: contexts of ONE process under different tuning structs (what a Rust host does instead of exporting
    variables) -- every field at a non-default value, the process-wide values untouched -- give the oracle's bytes; a struct with
    an out-of-range field is refused and creates nothing."""
    c, pk, rs, z2, two, _ = _tuning_truth()
    before = gpu.get_tuning().as_dict()
    cases = [dict(graph_mode=gpu.GRAPH_SPLIT), dict(graph_mode=gpu.GRAPH_OFF), dict(graph_mode_batch=gpu.GRAPH_OFF), dict(prove_streams=3),
             dict(prove_streams=4, linear_chains=0), dict(linear_chains=1), dict(coalesce_inflight=0), dict(coalesce_inflight=4, coalesce_gather_us=0),
             dict(batch_inflight=1), dict(queue_aware=0), dict(window_bits_narrow=7), dict(window_bits_wide=9, window_bits_h=10, window_bits_g2=8),
             dict(full_table_bytes=500_000_000), dict(full_table_bytes=0)]
    assert set(k for cs in cases for k in cs) | {"msm_dedicated_queues"} == set(before), "a field of mg_tuning has no case here"
    r1cs = gpu.R1CS.from_circuit(c)
    for cs in cases:
        ctx = gpu.ProvingContext(0, pk, tuning=cs)
        ctx.set_r1cs(r1cs)
        for i in range(5):
            assert gpu.Groth16.prove_with_randomness(ctx, z2 if i & 1 else c.z, rs[2 * (i & 1)], rs[2 * (i & 1) + 1]).hex() == two[i & 1], (cs, i)
        for rep in range(3):
            got = gpu.Groth16.prove_batch(ctx, np.stack([c.z, z2] * 4), np.stack([rs[0], rs[2]] * 4), np.stack([rs[1], rs[3]] * 4))
            assert [g.hex() for g in got] == two * 4, (cs, rep)
        if cs.get("full_table_bytes") == 0:
            assert ctx.table_bytes()[1] == 0
        ctx.close()
    assert gpu.get_tuning().as_dict() == before  # per-context tuning does not leak into the process
    with pytest.raises(gpu.MantaGpuError):
        gpu.ProvingContext(0, pk, tuning=gpu.get_tuning().replace(prove_streams=2))
