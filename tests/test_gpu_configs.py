"""GPU parity at the sizes BASELINE.json's configs name (each through the C ABI, each against the CPU oracle):
  configs[2]  2^20 Fr NTT + 2^20 G1/G2 MSM -> one BLS12-381 proof            test_config2_*
  configs[3]  PrivateTransfer proof with every MSM range-sharded over devices test_sharded_*   (device 0 listed G times:
              the box has one GPU; the code path -- per-device engines, slices, partial-point sum -- is the multi-GPU one)
  configs[4]  batch of 256 PrivateTransfer proofs, 256 distinct assignments   test_config4_*
and the ToPublic shape (manta-benchmark/benches/to_public.rs), plus regressions for the boundary's failure modes."""
import threading

import numpy as np
import pytest

import helpers as H
import oracle_lib as O
from manta_rs_amd import keygen, synth

pytestmark = pytest.mark.gpu


# ------------------------------------------------------------------------------------------------ configs[2]
@pytest.mark.parametrize("pre", [16, 0])
def test_config2_g2_msm_2_20_closed_form(gpu, pre):
    """BLS12-381 G2 MSM at n = 2^20 (SURVEY a-8 at BASELINE size): bases Q_i = [s0 + i s1] G2 from the library's
    fixed-base multiply, expectation [sum k_i (s0 + i s1) mod r] G2 from ONE oracle scalar multiplication; uniform and
    witness-like scalars, with the c = 16 window tables the 2^20 prover uses and with plain bases."""
    curve, n = 1, 1 << 20
    p = synth.FR_MODULUS[curve]
    s0, s1 = 0x7654321, 0x1fedcba987
    kb = np.zeros((n, 4), dtype=np.uint64)
    kb[:, 0] = np.uint64(s0) + np.arange(n, dtype=np.uint64) * np.uint64(s1)
    G2 = O.generator(curve, 2)
    dpts = gpu.fixed_base_mul(curve, 2, G2, gpu.DeviceBuffer.from_numpy(kb), n)
    b = gpu.Bases(curve, 2, (dpts.ptr, n), precompute_window_bits=pre, on_device=True)
    base_k = [s0 + i * s1 for i in range(n)]
    for dist in ("U", "W"):
        sc = synth.msm_scalars(curve, n, dist, seed=0x4D414E54 + (dist == "W"))
        got = gpu.VariableBaseMSM.launch(b, gpu.DeviceBuffer.from_numpy(sc), n, sparse=(dist == "W")).finish()
        t = sum(k * bk for k, bk in zip(synth.limbs_to_ints(sc), base_k)) % p
        assert (got == O.g_mul(curve, 2, G2, synth.ints_to_limbs([t], 4)[0])).all(), dist
    # a 2^12 prefix against the oracle's own Pippenger (different algorithm, same group element)
    m = 1 << 12
    host = dpts.to_numpy(shape=(n, 24))[:m]
    sc = synth.msm_scalars(curve, m, "U", seed=77)
    assert (gpu.VariableBaseMSM.launch(b, gpu.DeviceBuffer.from_numpy(sc), m).finish() == O.msm(curve, 2, host, sc)).all()


@pytest.mark.parametrize("curve,profile", [(1, "sparse"), (1, "W"), (0, "W")])
def test_config2_proof_at_2_20(gpu, curve, profile):
    """configs[2] end to end: D = V = 2^20, P = 16 synthetic circuit over BLS12-381 -- 7 NTTs of 2^20, 3 SpMVs, four G1
    MSMs and one G2 MSM of ~2^20 terms each. The proof bytes equal the CPU oracle's (all host cores: tens of seconds) and the
    proof satisfies the pairing equation; replayed through the captured hipGraphs it stays the same bytes. `W` is the witness
    bench.py's config2 leg proves (same seeds: 40 / 25 / 10 / 25 split -- the zero-digit compaction, the pair-count-dependent chunk
    length and the 13-bit G2 windows at full size: VERDICT r4 item 5); `sparse` the generator of rounds 1-3. Round 6: the same
    proof over BN254, manta-pay's own curve (SURVEY.md 8(d) config 3 names both; manta-pay/src/config/mod.rs:40,79)."""
    lg, P = 20, 16
    D = 1 << lg
    c = synth.make_circuit(curve, D - P, D, P, seed=0x4D414E5441_0301, profile=profile)
    assert (c.D, c.V) == (D, D)
    O.set_threads(O.usable_cpus())
    p = synth.FR_MODULUS[curve]
    rng = synth.XorShift(0x4D414E5441_0302)
    pk = keygen.generate(c, [rng.field(p) for _ in range(5)])
    ctx = gpu.ProvingContext(curve, pk)
    ctx.set_r1cs(gpu.R1CS.from_circuit(c))
    assert ctx.domain_size == D and ctx.num_variables == D
    rs = synth.to_mont([rng.field(p), rng.field(p)], p, 4)
    proof = gpu.Groth16.prove_with_randomness(ctx, c.z, rs[0], rs[1])
    assert O.groth16_verify(curve, pk, c.z[1:c.P], proof) == 1
    for _ in range(3):  # eager, capture, replay
        assert gpu.Groth16.prove_with_randomness(ctx, c.z, rs[0], rs[1]) == proof
    assert proof == O.groth16_prove(c, pk, rs[0], rs[1])


# ------------------------------------------------------------------------------------------------ shapes
def test_prove_real_shape_to_public(gpu):
    """Shape-exact ToPublic circuit (D = 2^15, V = 27 945, P = 19; manta-benchmark/benches/to_public.rs:26-42):
    bit-exact vs the oracle, pairing-verified, fuzzed input rejected."""
    c = synth.make_shape(0, "to_public")
    assert (c.D, c.V, c.P) == (1 << 15, 27945, 19)
    pk = keygen.generate(c, synth.from_mont(H.toxic(0, seed=6), synth.FR_MODULUS[0]))
    ctx = gpu.ProvingContext(0, pk)
    ctx.set_r1cs(gpu.R1CS.from_circuit(c))
    rs = H.rand_fr_mont(0, 4, seed=222)
    proof = gpu.Groth16.prove_with_randomness(ctx, c.z, rs[0], rs[1])
    assert proof == O.groth16_prove(c, pk, rs[0], rs[1])
    assert O.groth16_verify(0, pk, c.z[1:c.P], proof) == 1
    bad = c.z[1:c.P].copy()
    bad[0] = rs[2]
    assert O.groth16_verify(0, pk, bad, proof) == 0
    got = gpu.Groth16.prove_batch(ctx, np.stack([c.z] * 2), rs[0:4:2], rs[1:4:2])
    assert got[0] == proof and got[1] == O.groth16_prove(c, pk, rs[2], rs[3])


# ------------------------------------------------------------------------------------------------ configs[4]
def test_config4_batch_256_private_transfer(gpu):
    """configs[4]: 256 PrivateTransfer-shape proofs in one mg_groth16_prove_batch pass -- 256 DISTINCT satisfying
    assignments and (r, s) pairs (the signer's batch, manta-accounting/src/wallet/signer/functions.rs:748-800). All
    256 satisfy the pairing equation against their own public inputs, 8 spread over the batch are byte-compared with
    the CPU oracle, and a proof checked against its neighbour's inputs is rejected."""
    curve, k = 0, 256
    c0 = synth.make_shape(curve, "private_transfer")
    pk = keygen.generate(c0, synth.from_mont(H.toxic(curve, seed=16), synth.FR_MODULUS[curve]))
    ctx = gpu.ProvingContext(curve, pk)
    ctx.set_r1cs(gpu.R1CS.from_circuit(c0))
    R = synth.Reassigner(c0)
    cs = [R.assign(0x4D414E5441_1000 + q) for q in range(k)]
    assert synth.check_satisfied(cs[17]) and not (cs[17].z == cs[18].z).all()
    rs = H.rand_fr_mont(curve, 2 * k, seed=90)
    zs = gpu.PinnedArray.like(np.stack([x.z for x in cs]))
    got = gpu.Groth16.prove_batch(ctx, zs.array, rs[:k], rs[k:])
    assert len(got) == k and len(set(got)) == k
    for q in range(k):
        assert O.groth16_verify(curve, pk, cs[q].z[1:c0.P], got[q]) == 1, q
    assert O.groth16_verify(curve, pk, cs[1].z[1:c0.P], got[0]) == 0
    for q in (0, 1, 63, 64, 127, 128, 200, 255):
        assert got[q] == O.groth16_prove(cs[q], pk, rs[q], rs[k + q]), q
    # member q of the batch is byte-identical to the single call
    assert gpu.Groth16.prove_with_randomness(ctx, cs[5].z, rs[5], rs[k + 5]) == got[5]
    zs.free()


# ------------------------------------------------------------------------------------------------ configs[3]: sharding
@pytest.mark.parametrize("curve,group", [(1, 1), (0, 1), (1, 2), (0, 2)])
@pytest.mark.parametrize("shards,pre", [(2, 0), (3, 9)])
def test_sharded_msm_matches_oracle(gpu, curve, group, shards, pre):
    """mg_bases_create_sharded / mg_msm / mg_msm_launch_sharded: contiguous range shards, one Pippenger pass per
    shard, partial points added at finish. Infinity bases, ragged split (n not divisible), fewer scalars than bases."""
    n = 2999
    pts = H.random_points(curve, group, n, seed=300 + shards)
    pts[5] = 0
    pts[n - 1] = 0
    devs = [0] * shards
    b = gpu.Bases(curve, group, pts, precompute_window_bits=pre, devices=devs)
    sh = b.shards()
    assert [s[0] for s in sh] == devs and sh[0][1] == 0 and sh[-1][2] == n
    assert all(sh[g][2] == sh[g + 1][1] for g in range(shards - 1))
    sc = synth.msm_scalars(curve, n, "W", seed=301)
    want = O.msm(curve, group, pts, sc)
    assert (gpu.VariableBaseMSM.multi_scalar_mul(b, sc) == want).all()
    m = sh[1][1] + 3  # scalars end inside shard 1: zip to the shorter side, later shards idle
    assert (gpu.VariableBaseMSM.multi_scalar_mul(b, sc[:m]) == O.msm(curve, group, pts[:m], sc[:m])).all()
    bufs = [gpu.DeviceBuffer.from_numpy(sc[lo:hi]) for (_, lo, hi) in sh]
    assert (gpu.VariableBaseMSM.launch_sharded(b, bufs, sparse=True).finish() == want).all()


def test_sharded_msm_2_20_closed_form(gpu):
    """The 2^20 BLS12-381 G1 MSM of configs[1] split into 4 range shards of 2^18 (what every rank of the 4-GPU run
    computes, here all on device 0): the sum of the four partial points is the closed-form point."""
    curve, n, G = 1, 1 << 20, 4
    p = synth.FR_MODULUS[curve]
    s0, s1 = 0x1234567, 0x89abcdef1
    kb = np.zeros((n, 4), dtype=np.uint64)
    kb[:, 0] = np.uint64(s0) + np.arange(n, dtype=np.uint64) * np.uint64(s1)
    Gen = O.generator(curve, 1)
    host = gpu.fixed_base_mul(curve, 1, Gen, gpu.DeviceBuffer.from_numpy(kb), n).to_numpy(shape=(n, 12))
    b = gpu.Bases(curve, 1, host, precompute_window_bits=14, devices=[0] * G)
    sc = synth.msm_scalars(curve, n, "U", seed=35)
    bufs = [gpu.DeviceBuffer.from_numpy(sc[lo:hi]) for (_, lo, hi) in b.shards()]
    got = gpu.VariableBaseMSM.launch_sharded(b, bufs).finish()
    t = sum(k * (s0 + i * s1) for i, k in enumerate(synth.limbs_to_ints(sc))) % p
    assert (got == O.g_mul(curve, 1, Gen, synth.ints_to_limbs([t], 4)[0])).all()


@pytest.mark.parametrize("curve,shards", [(0, 2), (0, 3), (1, 2)])
def test_sharded_context_proofs_equal_single_device(gpu, curve, shards):
    """mg_ctx_create_sharded: every query of the key split into `shards` contiguous slices (all on device 0 here),
    every proof = shards x 5 partial MSMs + host sum. Same bytes as the oracle, single and batched, with infinity
    entries in the queries, r = 0, and through the graph replay."""
    c = synth.make_circuit(curve, 1500, 1100, 9, seed=400 + shards)
    pk = O.groth16_setup(c, H.toxic(curve, seed=17))
    ctx = gpu.ProvingContext(curve, pk, devices=[0] * shards)
    assert ctx.num_shards == shards
    ctx.set_r1cs(gpu.R1CS.from_circuit(c))
    rs = H.rand_fr_mont(curve, 8, seed=91)
    want = [O.groth16_prove(c, pk, rs[2 * q], rs[2 * q + 1]) for q in range(4)]
    for _ in range(4):  # eager, eager, capture, replay -- on every shard
        assert gpu.Groth16.prove_with_randomness(ctx, c.z, rs[0], rs[1]) == want[0]
    assert O.groth16_verify(curve, pk, c.z[1:c.P], want[0]) == 1
    assert gpu.Groth16.prove_batch(ctx, np.stack([c.z] * 4), rs[0:8:2], rs[1:8:2]) == want
    zero = np.zeros(4, dtype=np.uint64)
    assert gpu.Groth16.prove_with_randomness(ctx, c.z, zero, rs[1]) == O.groth16_prove(c, pk, zero, rs[1])


def test_sharded_private_transfer_proof(gpu):
    """configs[3]: the PrivateTransfer-shape proof with its five MSMs range-sharded 8 ways (the 8-GPU layout,
    all shards on device 0): byte-identical to the oracle and pairing-verified; also from the key's wire format."""
    curve = 0
    c = synth.make_shape(curve, "private_transfer")
    pk = keygen.generate(c, synth.from_mont(H.toxic(curve, seed=8), synth.FR_MODULUS[curve]))
    ctx = gpu.ProvingContext(curve, pk, devices=[0] * 8)
    ctx.set_r1cs(gpu.R1CS.from_circuit(c))
    rs = H.rand_fr_mont(curve, 2, seed=321)
    want = O.groth16_prove(c, pk, rs[0], rs[1])
    for _ in range(4):
        assert gpu.Groth16.prove_with_randomness(ctx, c.z, rs[0], rs[1]) == want
    assert O.groth16_verify(curve, pk, c.z[1:c.P], want) == 1


# ------------------------------------------------------------------------------------------------ boundary regressions
def test_set_r1cs_is_all_or_nothing(gpu):
    """A rejected mg_ctx_set_r1cs leaves the context exactly as it was (round-1 advisor finding): a good A followed
    by a bad B or C, a non-monotone row_ptr, row_ptr[0] != 0 -- each returns INVALID_ARGUMENT and the previous circuit
    still proves the oracle's bytes."""
    import copy
    curve = 0
    c = synth.make_circuit(curve, 400, 300, 5, seed=501)
    pk = O.groth16_setup(c, H.toxic(curve, seed=18))
    ctx = gpu.ProvingContext(curve, pk)
    good = gpu.R1CS.from_circuit(c)
    ctx.set_r1cs(good)
    rs = H.rand_fr_mont(curve, 2, seed=92)
    want = O.groth16_prove(c, pk, rs[0], rs[1])
    for _ in range(3):
        assert gpu.Groth16.prove_with_randomness(ctx, c.z, rs[0], rs[1]) == want

    def broken(which, how):
        b = copy.deepcopy(c)
        M = getattr(b, which)
        if how == "col":
            M.col[len(M.col) // 2] = c.V + 7
        elif how == "monotone":
            M.row_ptr[10], M.row_ptr[11] = M.row_ptr[11] + 5, M.row_ptr[10]
        elif how == "first":
            M.row_ptr[0] = 1
        elif how == "last":
            M.row_ptr[-1] -= 1
        return gpu.R1CS.from_circuit(b)

    for which in ("B", "C", "A"):
        for how in ("col", "monotone", "first", "last"):
            with pytest.raises(gpu.MantaGpuError) as e:
                ctx.set_r1cs(broken(which, how))
            assert e.value.status == 1, (which, how)
            assert gpu.Groth16.prove_with_randomness(ctx, c.z, rs[0], rs[1]) == want, (which, how)
    # key generation validates the same way (its host loop walks row_ptr too)
    import dataclasses
    with pytest.raises(gpu.MantaGpuError):
        keygen.generate(dataclasses.replace(c, A=broken("A", "monotone").A),
                        synth.from_mont(H.toxic(curve), synth.FR_MODULUS[curve]))


def test_python_mirror_rejects_short_buffers_and_tracks_the_circuit(gpu):
    """api.py never lets a short assignment reach the C side, and `Groth16.prove` re-uploads the matrices when handed
    a different R1CS object with the same (m, P) (round-1 advisor finding)."""
    curve = 0
    c1 = synth.make_circuit(curve, 300, 260, 4, seed=601)
    c2 = synth.make_circuit(curve, 300, 260, 4, seed=602)  # same m, V, P -- another circuit
    assert c1.A.col.shape != c2.A.col.shape or not (c1.A.col == c2.A.col).all()
    tox = H.toxic(curve, seed=19)
    pk1, pk2 = O.groth16_setup(c1, tox), O.groth16_setup(c2, tox)
    rs = H.rand_fr_mont(curve, 2, seed=93)
    ctx1 = gpu.ProvingContext(curve, pk1)
    ctx1.set_r1cs(gpu.R1CS.from_circuit(c1))
    with pytest.raises(ValueError):
        gpu.Groth16.prove_with_randomness(ctx1, c1.z[:-1], rs[0], rs[1])
    with pytest.raises(ValueError):
        gpu.Groth16.prove_batch(ctx1, np.stack([c1.z] * 3)[:, :-2], rs[:1].repeat(3, 0), rs[1:].repeat(3, 0))
    with pytest.raises(ValueError):
        ctx1.witness_map(c1.z[:10])
    it = iter([rs[0], rs[1]])
    r1 = gpu.R1CS.from_circuit(c1)
    assert gpu.Groth16.prove(ctx1, r1, lambda: next(it)) == O.groth16_prove(c1, pk1, rs[0], rs[1])
    # the same context object asked to prove c2's R1CS: the matrices must be c2's, so with c2's key the bytes are the
    # oracle's (with the stale matrices of c1 they would not be)
    ctx2 = gpu.ProvingContext(curve, pk2)
    ctx2.set_r1cs(r1)
    it = iter([rs[0], rs[1]])
    assert gpu.Groth16.prove(ctx2, gpu.R1CS.from_circuit(c2), lambda: next(it)) == O.groth16_prove(c2, pk2, rs[0], rs[1])


def test_set_r1cs_while_proofs_are_in_flight(gpu):
    """mantagpu.h: prove is re-entrant and set_r1cs waits for the passes in flight. Four threads prove in a loop while
    the main thread swaps the circuit back and forth; every proof is the oracle's for the circuit of one of the two
    generations (same variables, same key shape), never a mixture, and nothing crashes."""
    curve = 0
    c1 = synth.make_circuit(curve, 250, 300, 5, seed=701)   # D = 256
    c2 = synth.make_circuit(curve, 900, 300, 5, seed=702)   # D = 1024
    pk = O.groth16_setup(c2, H.toxic(curve, seed=20))
    ctx = gpu.ProvingContext(curve, pk)
    r1, r2 = gpu.R1CS.from_circuit(c1), gpu.R1CS.from_circuit(c2)
    rs = H.rand_fr_mont(curve, 2, seed=94)
    ok = {O.groth16_prove(c1, pk, rs[0], rs[1], z=c2.z), O.groth16_prove(c2, pk, rs[0], rs[1])}
    ctx.set_r1cs(r2)
    stop, bad = threading.Event(), []

    def work():
        while not stop.is_set():
            got = gpu.Groth16.prove_with_randomness(ctx, c2.z, rs[0], rs[1])
            if got not in ok:
                bad.append(got)
    ts = [threading.Thread(target=work) for _ in range(4)]
    [t.start() for t in ts]
    for i in range(12):
        ctx.set_r1cs(r1 if i % 2 == 0 else r2)
    stop.set()
    [t.join() for t in ts]
    assert not bad
    ctx.set_r1cs(r2)
    assert gpu.Groth16.prove_with_randomness(ctx, c2.z, rs[0], rs[1]) == O.groth16_prove(c2, pk, rs[0], rs[1])


def test_idle_slot_cache_is_bounded(gpu):
    """Proof slots are cached per batch size; the cache is capped (6 idle slots per context), so a service that
    varies k does not pin HBM for every size it ever used. 12 batch sizes, then the first again -- all the oracle's bytes."""
    curve = 0
    c = synth.make_circuit(curve, 300, 260, 4, seed=801)
    pk = O.groth16_setup(c, H.toxic(curve, seed=21))
    ctx = gpu.ProvingContext(curve, pk)
    ctx.set_r1cs(gpu.R1CS.from_circuit(c))
    rs = H.rand_fr_mont(curve, 2, seed=95)
    want = O.groth16_prove(c, pk, rs[0], rs[1])
    for k in list(range(1, 13)) + [1, 2]:
        got = gpu.Groth16.prove_batch(ctx, np.stack([c.z] * k), np.stack([rs[0]] * k), np.stack([rs[1]] * k))
        assert got == [want] * k
