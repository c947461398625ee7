"""GPU parity: NTT, witness map and whole Groth16 proofs vs the CPU oracle, bit-exact."""
import numpy as np
import pytest

import helpers as H
import oracle_lib as O
from manta_rs_amd import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("curve", [0, 1])
@pytest.mark.parametrize("log_n", [0, 1, 2, 5, 10, 13])
def test_ntt_matches_oracle(gpu, curve, log_n):
    n = 1 << log_n
    x = H.rand_fr_mont(curve, n, seed=log_n + 1)
    dom = gpu.Radix2EvaluationDomain(curve, n)
    for inverse in (False, True):
        for coset in (False, True):
            want = O.ntt(curve, x, inverse=inverse, coset=coset)
            got = dom._run(x, inverse, coset)
            assert (got == want).all(), (inverse, coset)
    assert (dom.ifft(dom.fft(x)) == x).all()
    assert (dom.coset_ifft(dom.coset_fft(x)) == x).all()


@pytest.mark.parametrize("curve", [0, 1])
@pytest.mark.parametrize("m,V,P", [(100, 70, 5), (1000, 700, 13), (3000, 3500, 27)])
def test_prove_matches_oracle_and_verifies(gpu, curve, m, V, P):
    c = synth.make_circuit(curve, m, V, P, seed=m)
    assert synth.check_satisfied(c)
    pk = O.groth16_setup(c, H.toxic(curve))
    ctx = gpu.ProvingContext(curve, pk)
    r1cs = gpu.R1CS.from_circuit(c)
    ctx.set_r1cs(r1cs)
    assert ctx.domain_size == c.D
    assert (ctx.witness_map(c.z) == O.witness_map(c)).all()
    rs = H.rand_fr_mont(curve, 2, seed=99)
    it = iter(rs)
    proof = gpu.Groth16.prove(ctx, r1cs, lambda: next(it))
    want = O.groth16_prove(c, pk, rs[0], rs[1], msm_algo=1)
    assert proof == want
    assert O.groth16_verify(curve, pk, c.z[1:c.P], proof) == 1
    # r = 0 skips g1_b (App. B.1) and must still agree
    zero = np.zeros(4, dtype=np.uint64)
    assert gpu.Groth16.prove_with_randomness(ctx, c.z, zero, rs[1]) == O.groth16_prove(c, pk, zero, rs[1])
    # an unsatisfying witness is not an error: same bytes as the reference algorithm, proof rejected
    z_bad = c.z.copy()
    z_bad[c.P + 1] = rs[0]
    bad = gpu.Groth16.prove_with_randomness(ctx, z_bad, rs[0], rs[1])
    assert bad == O.groth16_prove(c, pk, rs[0], rs[1], z=z_bad)


@pytest.mark.parametrize("curve", [0, 1])
def test_gpu_keygen_matches_oracle_setup(gpu, curve):
    """manta_rs_amd.keygen (host scalars + GPU fixed-base batch multiply) reproduces the oracle's toy setup
    point-for-point, for G1 and G2 queries incl. infinity entries."""
    from manta_rs_amd import keygen
    c = synth.make_circuit(curve, 300, 200, 7, seed=77)
    tox = H.toxic(curve, seed=21)
    want = O.groth16_setup(c, tox)
    got = keygen.generate(c, synth.from_mont(tox, synth.FR_MODULUS[curve]))
    for f in ("alpha_g1", "beta_g1", "delta_g1", "beta_g2", "gamma_g2", "delta_g2", "gamma_abc_g1", "a_query",
              "b_g1_query", "b_g2_query", "h_query", "l_query"):
        assert (getattr(got, f) == getattr(want, f)).all(), f
    assert any(not row.any() for row in got.b_g1_query)  # infinity entries present


def test_prove_real_shape_to_private(gpu):
    """Shape-exact ToPrivate circuit (D=2^14, V=8253, P=13): proof bytes equal the oracle's and the
    proof verifies."""
    from manta_rs_amd import keygen
    c = synth.make_shape(0, "to_private")
    assert (c.D, c.V, c.P) == (1 << 14, 8253, 13)
    pk = keygen.generate(c, synth.from_mont(H.toxic(0, seed=5), synth.FR_MODULUS[0]))
    ctx = gpu.ProvingContext(0, pk)
    ctx.set_r1cs(gpu.R1CS.from_circuit(c))
    rs = H.rand_fr_mont(0, 2, seed=123)
    proof = gpu.Groth16.prove_with_randomness(ctx, c.z, rs[0], rs[1])
    assert proof == O.groth16_prove(c, pk, rs[0], rs[1])
    assert O.groth16_verify(0, pk, c.z[1:c.P], proof) == 1


_TABLES_SCRIPT = '''
import os, sys
sys.path.insert(0, r"{root}"); sys.path.insert(0, os.path.join(r"{root}", "tests"))
import numpy as np
import helpers as H
from manta_rs_amd import api as gpu, synth, keygen
gpu.init(0)
c = synth.make_shape(0, "to_public")
pk = keygen.generate(c, synth.from_mont(H.toxic(0, seed=5), synth.FR_MODULUS[0]))
ctx = gpu.ProvingContext(0, pk)
ctx.set_r1cs(gpu.R1CS.from_circuit(c))
rs = H.rand_fr_mont(0, 4, seed=99)
bucket, full = ctx.table_bytes()
print("TABLES", bucket, full)
print("PROOF", gpu.Groth16.prove_with_randomness(ctx, c.z, rs[0], rs[1]).hex())
print("PROOF", gpu.Groth16.prove_with_randomness(ctx, c.z, rs[2], rs[3]).hex())
'''


def test_single_proofs_on_full_and_on_bucket_tables(gpu):
    """A proof-sized key keeps FULL tables of its queries (every multiple of every window: passes of one proof are plain sums,
    no sort, no bucket reduce) next to the bucket tables; MANTA_FULL_TABLE_GB (GB per context, overriding mg_ctx_opts.full_table_bytes) bounds them, 0 leaves them out -- the path a
    context takes when HBM is short. The knob is read once per process, hence the children: the ToPublic shape with the
    default budget, with a small one (narrower windows, some queries without) and with none must give the oracle's bytes."""
    import os
    import subprocess
    import sys
    from manta_rs_amd import keygen
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    c = synth.make_shape(0, "to_public")
    pk = keygen.generate(c, synth.from_mont(H.toxic(0, seed=5), synth.FR_MODULUS[0]))
    rs = H.rand_fr_mont(0, 4, seed=99)
    want = [O.groth16_prove(c, pk, rs[0], rs[1]).hex(), O.groth16_prove(c, pk, rs[2], rs[3]).hex()]
    sizes = {}
    for gb in ("", "10", "0"):
        env = dict(os.environ)
        env.pop("MANTA_FULL_TABLE_GB", None)
        if gb:
            env["MANTA_FULL_TABLE_GB"] = gb
        out = subprocess.run([sys.executable, "-c", _TABLES_SCRIPT.format(root=root)], env=env, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
        lines = out.stdout.split("\n")
        assert [ln.split()[1] for ln in lines if ln.startswith("PROOF")] == want, gb
        sizes[gb] = [int(x) for x in [ln for ln in lines if ln.startswith("TABLES")][0].split()[1:]]
    assert sizes["0"][1] == 0 and sizes["0"][0] > 0
    assert 0 < sizes["10"][1] <= 10e9 and sizes[""][1] > sizes["10"][1]
    assert sizes[""][0] == sizes["0"][0]


_Z3_SCRIPT = '''
import os, sys
sys.path.insert(0, r"{root}"); sys.path.insert(0, os.path.join(r"{root}", "tests"))
import helpers as H
from manta_rs_amd import api as gpu, synth, keygen
gpu.init(0)
c = synth.make_shape(0, "to_public", profile="W")
pk = keygen.generate(c, synth.from_mont(H.toxic(0, seed=6), synth.FR_MODULUS[0]))
ctx = gpu.ProvingContext(0, pk)
ctx.set_r1cs(gpu.R1CS.from_circuit(c))
rs = H.rand_fr_mont(0, 4, seed=98)
rs[2][:] = 0  # r = 0: b_g1 is not used
for i in range(5):  # eager, eager, capture, replay, replay
    print("PROOF", gpu.Groth16.prove_with_randomness(ctx, c.z, rs[2 * (i & 1)], rs[2 * (i & 1) + 1]).hex())
'''


def test_single_proof_host_fold_variants(gpu):
    """Single proofs run the a | b_g1 | l queries as ONE MSM over a concatenated full table. Round 4: one digit launch per
    query leaves the pairs grouped by query (no sort), and the MSM's end-of-chain token lets the host compute s A + r B1 and
    finish A and the G2 element while the h chain is still running. MANTA_Z3_SORT=1 / MANTA_Z3_EARLY=0 restore the single
    launch + radix pass / the wait for all of part A; MANTA_Z3=0 the three separate MSMs. Knobs are read once per process,
    hence the children: every variant must give the oracle's bytes, eager and replayed, with r != 0 and r = 0."""
    import os
    import subprocess
    import sys
    from manta_rs_amd import keygen
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    c = synth.make_shape(0, "to_public", profile="W")
    pk = keygen.generate(c, synth.from_mont(H.toxic(0, seed=6), synth.FR_MODULUS[0]))
    rs = H.rand_fr_mont(0, 4, seed=98)
    rs[2][:] = 0
    O.set_threads(O.usable_cpus())
    two = [O.groth16_prove(c, pk, rs[0], rs[1]).hex(), O.groth16_prove(c, pk, rs[2], rs[3]).hex()]
    want = [two[i & 1] for i in range(5)]
    for knobs in ({}, {"MANTA_Z3_EARLY": "0"}, {"MANTA_Z3_SORT": "1"}, {"MANTA_Z3_EARLY": "0", "MANTA_Z3_SORT": "1"}, {"MANTA_Z3": "0"}):
        env = H.knob_env(knobs, strip_prefix="MANTA_Z3")  # (A/B switches: the diagnosis twin)
        out = subprocess.run([sys.executable, "-c", _Z3_SCRIPT.format(root=root)], env=env, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
        assert [ln.split()[1] for ln in out.stdout.split("\n") if ln.startswith("PROOF")] == want, knobs


_R5_SCRIPT = '''
import os, sys
sys.path.insert(0, r"{root}"); sys.path.insert(0, os.path.join(r"{root}", "tests"))
import numpy as np
import helpers as H
from manta_rs_amd import api as gpu, synth, keygen
gpu.init(0)
c = synth.make_shape(0, "to_public", profile="W")
pk = keygen.generate(c, synth.from_mont(H.toxic(0, seed=6), synth.FR_MODULUS[0]))
ctx = gpu.ProvingContext(0, pk)
ctx.set_r1cs(gpu.R1CS.from_circuit(c))
rs = H.rand_fr_mont(0, 4, seed=98)
rs[2][:] = 0  # r = 0: b_g1 is not used
z2 = synth.Reassigner(c).assign(0x5EED).z
for i in range(5):  # eager, eager, capture, replay, replay; two assignments alternate
    print("PROOF", gpu.Groth16.prove_with_randomness(ctx, z2 if i & 1 else c.z, rs[2 * (i & 1)], rs[2 * (i & 1) + 1]).hex())
for rep in range(4):  # a pass of 8 (two assignments): eager, eager, capture, replay
    got = gpu.Groth16.prove_batch(ctx, np.stack([c.z, z2] * 4), np.stack([rs[0], rs[2]] * 4), np.stack([rs[1], rs[3]] * 4))
    print("BATCH", " ".join(g.hex() for g in got))
import threading
def worker(t):  # two host threads at once: single proofs beside another pass (slots of their own: eager, capture, replay)
    for i in range(10):
        j = (i + t) & 1
        print("CONC %d %s" % (j, gpu.Groth16.prove_with_randomness(ctx, z2 if j else c.z, rs[2 * j], rs[2 * j + 1]).hex()), flush=True)
ts = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
[t.start() for t in ts]
[t.join() for t in ts]
'''


def test_round5_knobs_do_not_change_results(gpu):
    """Every environment knob added in round 5 selects between code paths that must ALL give the oracle's bytes (two
    knobs of earlier rounds could change results; none may): three linear graphs or the forked one for a lone single proof, the
    launch order of the three, fused / unfused witness-map passes (both halves), twiddles in LDS, the in-workgroup sum of
    single-key MSMs (G1, G2, both, none), the two-pass sort of batched dense MSMs, the graph topology of batched passes, the
    queue-aware stream sets (off; linear graphs for lone proofs only; the combined MSM's stream forced to either priority). Single
    proofs (eager and replayed, r = 0 included, two assignments alternating on one slot), passes of 8, and single proofs from
    two host threads at once; knobs are read once per process, hence the children."""
    import os
    import subprocess
    import sys
    from manta_rs_amd import keygen
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    c = synth.make_shape(0, "to_public", profile="W")
    pk = keygen.generate(c, synth.from_mont(H.toxic(0, seed=6), synth.FR_MODULUS[0]))
    rs = H.rand_fr_mont(0, 4, seed=98)
    rs[2][:] = 0
    z2 = synth.Reassigner(c).assign(0x5EED).z
    O.set_threads(O.usable_cpus())
    two = [O.groth16_prove(c, pk, rs[0], rs[1]).hex(), O.groth16_prove(c, pk, rs[2], rs[3], z=z2).hex()]
    want = [two[i & 1] for i in range(5)]
    want_batch = " ".join(two * 4)
    variants = ({}, {"MANTA_Z3_LINEAR": "0"}, {"MANTA_Z3_ORDER": "zba"}, {"MANTA_NTT_FUSE": "0"}, {"MANTA_NTT_FUSE": "1"}, {"MANTA_NTT_FUSE": "2"},
                {"MANTA_NTT_TWL": "0", "MANTA_NTT_FUSE": "0"}, {"MANTA_ACC_SINGLE": "0"}, {"MANTA_ACC_SINGLE": "2"}, {"MANTA_ACC_SINGLE": "3"},
                {"MANTA_SORT_LOW": "0"}, {"MANTA_Z3_LINEAR": "1"}, {"MANTA_Z3_LINEAR": "2"}, {"MANTA_Z3_HIGH": "1"}, {"MANTA_Z3_HIGH": "0"})
    # (MANTA_GRAPH_BATCH split / off and MANTA_QUEUE_AWARE=0 moved to the tuning sweep of tests/test_gpu_profiles.py in round 6: they are
    # variables of the shipped library's table now)
    for knobs in variants:
        env = H.knob_env(knobs, strip_prefix="MANTA_")  # tuning-table names: the shipped library; A/B switches: the diagnosis twin
        out = subprocess.run([sys.executable, "-c", _R5_SCRIPT.format(root=root)], env=env, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, (knobs, out.stdout[-2000:] + out.stderr[-2000:])
        lines = out.stdout.split("\n")
        got = [ln.split()[1] for ln in lines if ln.startswith("PROOF")]
        assert got == want, H.proof_diff(got, want, knobs, out.stdout, "round5_knobs_single")
        batches = [ln[6:] for ln in lines if ln.startswith("BATCH")]
        assert len(batches) == 4, knobs
        for b in batches:
            assert b == want_batch, H.proof_diff(b.split(), want_batch.split(), knobs, out.stdout, "round5_knobs_batch")
        conc = [ln.split() for ln in lines if ln.startswith("CONC")]
        assert len(conc) == 20, knobs
        assert all(two[int(j)] == h for _, j, h in conc), H.proof_diff([h for _, j, h in conc], [two[int(j)] for _, j, h in conc], knobs, out.stdout,
                                                                         "round5_knobs_threads")


def test_captured_graphs_survive_other_contexts(gpu):
    """A context's captured graphs (part A forked, the G2 chain linear; passes of 1, 2, 3 and 8 proofs) must keep giving the
    oracle's bytes while OTHER contexts of the process are created, load a circuit (window tables built on the default stream,
    GBs allocated), prove eagerly and capture graphs of their own. Round 4 found that a part-A graph captured as ONE linear
    chain (MANTA_PROVE_STREAMS=1, a measurement knob) gives a wrong C element exactly there -- profiles/r04_z3_helper_thread.txt;
    the default topology must not. Every element of every proof is compared (A | B | C)."""
    c = synth.make_circuit(0, 700, 500, 9, seed=11)
    pk = O.groth16_setup(c, H.toxic(0))
    ctx = gpu.ProvingContext(0, pk)
    r1cs = gpu.R1CS.from_circuit(c)
    ctx.set_r1cs(r1cs)
    K = 8
    rs = H.rand_fr_mont(0, 2 * K, seed=5)
    rs[1][:] = 0  # one member with r = 0
    truth = [O.groth16_prove(c, pk, rs[q], rs[K + q]) for q in range(K)]
    zs = np.stack([c.z] * K)

    def check(tag):
        assert gpu.Groth16.prove_with_randomness(ctx, c.z, rs[0], rs[K]) == truth[0], tag
        assert gpu.Groth16.prove_with_randomness(ctx, c.z, rs[1], rs[K + 1]) == truth[1], tag
        for k in (2, 3, K):
            got = gpu.Groth16.prove_batch(ctx, zs[:k], rs[0:k], rs[K:K + k])
            for q in range(k):
                assert got[q] == truth[q], (tag, k, q, [got[q][a:b] == truth[q][a:b] for a, b in ((0, 32), (32, 96), (96, 128))])

    for i in range(4):  # eager, eager, capture + replay, replay -- for every pass size
        check("run %d" % i)
    c2 = gpu.ProvingContext(0, pk)
    check("after a second context was created")
    c2.set_r1cs(r1cs)
    check("after its set_r1cs")
    for i in range(3):
        assert gpu.Groth16.prove_with_randomness(c2, c.z, rs[0], rs[K]) == truth[0]
        check("after its proof %d" % (i + 1))
    c3 = gpu.ProvingContext(0, pk, full_table_bytes=0)
    c3.set_r1cs(r1cs)
    check("after a third context without full tables")
    got = gpu.Groth16.prove_batch(c3, zs[:3], rs[0:3], rs[K:K + 3])
    assert [got[q] for q in range(3)] == truth[:3]
    check("after its batched pass")
    c2.close()
    check("after the second context was closed")
    c3.close()


def _linear_child(env_extra):
    """tools/diag_linear.py --child in a process of its own (the topology knobs are read once per process): the first context proves
    after every step of a second context's life, each proof split A | B | C against the oracle. -> (number of wrong proofs, output)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if not k.startswith(("MANTA_", "MG_DIAG", "DIAG_"))}
    env.update(env_extra)
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "diag_linear.py"), "--child"], env=env, capture_output=True, text=True,
                         timeout=900)
    res = [ln for ln in out.stdout.split("\n") if ln.startswith("RESULT nbad=")]
    assert out.returncode == 0 and res, out.stdout[-3000:] + out.stderr[-2000:]
    return int(res[-1].split("=")[1]), out.stdout


def test_linear_graphs_are_right_and_the_guard_is_sensitive(gpu):
    """Round 5 root cause of round 4's wrong-C mode (profiles/r05_linear_graph_defect.txt): a hipMemsetAsync NODE inside a LINEAR
    captured graph (one stream: the runtime pre-builds the graph's AQL packets, its own fill kernel included --
    DEBUG_CLR_GRAPH_PACKET_CAPTURE) replays with a wrong fill pattern after other work has gone through the runtime; the a | b | c
    vectors of the witness map then start from garbage instead of zeros and C is wrong. The library no longer has a memset node in
    any captured graph (spmv3 writes its zero rows, zero_ranges the MSM's counters and buckets).
    NEGATIVE CONTROL first: the diagnosis twin of the library with the memset node put back (MG_DIAG_MEMSET=1) and part A linear must
    FAIL this very sequence at this very circuit size -- otherwise the sequence pins nothing. Then the same
    topologies without the node must be right: linear part A (diagnosis twin), six linear graphs (MANTA_GRAPH=split, shipped
    library), and the shipped default with MANTA_PROVE_STREAMS=1 in the environment (ignored outside -DMG_DIAG builds)."""
    import os
    from manta_rs_amd import api
    diag = os.path.join(os.path.dirname(api.LIB_PATH), "libmantagpu_diag.so")
    assert os.path.exists(diag), "manta_rs_amd/csrc/Makefile builds the diagnosis twin next to the library"
    nbad, out = _linear_child({"MANTA_LIB": diag, "MANTA_PROVE_STREAMS": "1", "MG_DIAG_MEMSET": "1"})
    assert nbad > 0, "the negative control passed: the guard sequence is not sensitive to the defect any more\n" + out
    assert "ok ok BAD" in out and "BAD ok" not in out and "BAD BAD" not in out, out  # C alone goes wrong: the h term
    for env in ({"MANTA_LIB": diag, "MANTA_PROVE_STREAMS": "1"}, {"MANTA_LIB": diag, "MANTA_GRAPH": "split"}, {"MANTA_GRAPH": "split"},
                {"MANTA_PROVE_STREAMS": "1"}, {"MANTA_GRAPH": "split", "DIAG_FULL_TABLE_BYTES": "0"}):
        nbad, out = _linear_child(env)
        assert nbad == 0, (env, out)


def test_prove_real_shape_private_transfer(gpu):
    """Shape-exact PrivateTransfer circuit (D=2^16, V=35175, P=27): bit-exact vs the oracle, pairing-verified,
    and -- like manta-pay/src/test/transfer.rs:346-417 -- a fuzzed public input must invalidate the proof."""
    from manta_rs_amd import keygen
    c = synth.make_shape(0, "private_transfer")
    assert (c.D, c.V, c.P) == (1 << 16, 35175, 27)
    pk = keygen.generate(c, synth.from_mont(H.toxic(0, seed=8), synth.FR_MODULUS[0]))
    ctx = gpu.ProvingContext(0, pk)
    ctx.set_r1cs(gpu.R1CS.from_circuit(c))
    rs = H.rand_fr_mont(0, 2, seed=321)
    proof = gpu.Groth16.prove_with_randomness(ctx, c.z, rs[0], rs[1])
    assert proof == O.groth16_prove(c, pk, rs[0], rs[1])
    assert O.groth16_verify(0, pk, c.z[1:c.P], proof) == 1
    bad = c.z[1:c.P].copy()
    bad[3] = rs[0]
    assert O.groth16_verify(0, pk, bad, proof) == 0
    # concurrent proofs on one context (the reference's signer shares a ProvingContext across threads)
    import threading
    outs = [None] * 8

    def work(i):
        outs[i] = gpu.Groth16.prove_with_randomness(ctx, c.z, rs[0], rs[1])
    ts = [threading.Thread(target=work, args=(i,)) for i in range(8)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert all(o == proof for o in outs)


@pytest.mark.parametrize("curve", [0, 1])
def test_ntt_full_size_roundtrip_and_linearity(gpu, curve):
    """2^20 (BASELINE size): ifft(fft(x)) = x, coset round trip, linearity fft(x + y) = fft(x) + fft(y); a 2^16
    slice is also compared with the oracle directly."""
    n = 1 << 20
    rs = np.random.RandomState(5)
    p = synth.FR_MODULUS[curve]

    def rand():
        a = rs.randint(0, 1 << 62, size=(n, 4), dtype=np.int64).astype(np.uint64)
        a[:, 3] &= np.uint64((1 << 60) - 1)  # < p
        return a
    x, y = rand(), rand()
    dom = gpu.Radix2EvaluationDomain(curve, n)
    fx = dom.fft(x)
    assert (dom.ifft(fx) == x).all()
    assert (dom.coset_ifft(dom.coset_fft(x)) == x).all()
    fr = "bn254_fr" if curve == 0 else "bls381_fr"
    assert (O.field_op(fr, "add", fx, dom.fft(y)) == dom.fft(O.field_op(fr, "add", x, y))).all()
    m = 1 << 16
    assert (gpu.Radix2EvaluationDomain(curve, m).coset_fft(x[:m]) == O.ntt(curve, x[:m], coset=True)).all()


def _pk_bytes(curve, pk):
    import fixture_io
    return fixture_io.pk_bytes(O, curve, pk)


@pytest.mark.parametrize("curve", [0, 1])
def test_proving_context_decode_wire_format(gpu, curve):
    """ProvingContext::decode (groth16.rs:268-288): a key in arkworks' serialize_unchecked byte format, incl.
    infinity entries, gives the same proofs as the same key passed as limb arrays; size matches App. C."""
    c = synth.make_circuit(curve, 400, 300, 9, seed=12)
    pk = O.groth16_setup(c, H.toxic(curve, seed=3))
    data = _pk_bytes(curve, pk)
    if curve == 0:
        # 64 + 3*128 + (8 + 64 P) + 64 + 64 + (8 + 64 V)*2 + (8 + 128 V) + (8 + 64 H) + (8 + 64 (V - P)) = 624 + 320 V + 64 H
        assert len(data) == 624 + 320 * c.V + 64 * (c.D - 1)
    ctx = gpu.ProvingContext.decode(curve, data)
    r1cs = gpu.R1CS.from_circuit(c)
    ctx.set_r1cs(r1cs)
    rs = H.rand_fr_mont(curve, 2, seed=4)
    proof = gpu.Groth16.prove_with_randomness(ctx, c.z, rs[0], rs[1])
    assert proof == O.groth16_prove(c, pk, rs[0], rs[1])
    assert O.groth16_verify(curve, pk, c.z[1:c.P], proof) == 1
    with pytest.raises(gpu.MantaGpuError):
        gpu.ProvingContext.decode(curve, data[:-5])          # truncated
    bad = bytearray(data)
    bad[31 if curve == 0 else 47] = 0x3f                    # x coordinate >= p (non-canonical)
    with pytest.raises(gpu.MantaGpuError):
        gpu.ProvingContext.decode(curve, bytes(bad))
    # manta-parameters' integrity check in front of the loader (`manta_parameters::verify`, lib.rs:173-177): the right
    # BLAKE3 digest loads the key, a wrong one is refused with MG_ERROR_CHECKSUM before anything is uploaded
    ok = gpu.ProvingContext.decode(curve, data, checksum=gpu.blake3(data))
    ok.set_r1cs(r1cs)
    assert gpu.Groth16.prove_with_randomness(ok, c.z, rs[0], rs[1]) == proof
    with pytest.raises(gpu.MantaGpuError) as e:
        gpu.ProvingContext.decode(curve, data, checksum=gpu.blake3(data[:-1]))
    assert e.value.status == 6


def test_error_codes_not_exceptions(gpu):
    """The C ABI reports failures as status codes (any non-zero -> the reference's opaque `Error`): domain
    larger than the field's two-adicity (PolynomialDegreeTooLarge), prove before the R1CS is set, bad arguments."""
    import ctypes
    with pytest.raises(gpu.MantaGpuError) as e:  # BN254 Fr has two-adicity 28: a 2^29 domain does not exist
        gpu._chk(gpu.LIB.mg_ntt_device(0, ctypes.c_void_p(8), 29, 0, 0), "mg_ntt_device")
    assert e.value.status == 4
    c = synth.make_circuit(0, 50, 40, 3, seed=1)
    pk = O.groth16_setup(c, H.toxic(0))
    ctx = gpu.ProvingContext(0, pk)
    rs = H.rand_fr_mont(0, 2, seed=1)
    with pytest.raises(gpu.MantaGpuError) as e:
        gpu.Groth16.prove_with_randomness(ctx, c.z, rs[0], rs[1])   # no R1CS yet
    assert e.value.status == 5
    with pytest.raises(gpu.MantaGpuError) as e:
        gpu.Bases(0, 3, np.ones((4, 8), dtype=np.uint64))           # no such group
    assert e.value.status == 1
    bad = synth.make_circuit(0, 50, 40, 3, seed=1)
    bad.A.col = bad.A.col.copy()
    bad.A.col[0] = 1000                                             # column index out of range
    with pytest.raises(gpu.MantaGpuError) as e:
        ctx.set_r1cs(gpu.R1CS.from_circuit(bad))
    assert e.value.status == 1
    ctx.set_r1cs(gpu.R1CS.from_circuit(c))
    assert gpu.Groth16.prove_with_randomness(ctx, c.z, rs[0], rs[1]) == O.groth16_prove(c, pk, rs[0], rs[1])


def test_context_reused_across_shapes_and_many_proofs(gpu):
    """One ProvingContext, the R1CS replaced by a different (larger-domain) one after the proof slots have
    captured their hipGraphs; then a run of proofs with fresh randomness each -- all byte-identical to the oracle."""
    curve = 0
    c1 = synth.make_circuit(curve, 200, 300, 5, seed=41)      # D = 256
    c2 = synth.make_circuit(curve, 900, 300, 5, seed=42)      # D = 1024, same V and P -> same key shape
    pk1 = O.groth16_setup(c1, H.toxic(curve, seed=9))
    ctx = gpu.ProvingContext(curve, pk1)
    rs = H.rand_fr_mont(curve, 12, seed=77)
    r1 = gpu.R1CS.from_circuit(c1)
    ctx.set_r1cs(r1)
    for i in range(4):  # eager, eager, capture, replay
        got = gpu.Groth16.prove_with_randomness(ctx, c1.z, rs[2 * i], rs[2 * i + 1])
        assert got == O.groth16_prove(c1, pk1, rs[2 * i], rs[2 * i + 1])
    # a different circuit over the same variables needs its own key; reuse the context object only for the R1CS swap
    pk2 = O.groth16_setup(c2, H.toxic(curve, seed=9))
    ctx2 = gpu.ProvingContext(curve, pk2)
    ctx2.set_r1cs(gpu.R1CS.from_circuit(c1))                  # wrong (smaller) circuit first ...
    gpu.Groth16.prove_with_randomness(ctx2, c1.z, rs[0], rs[1])
    gpu.Groth16.prove_with_randomness(ctx2, c1.z, rs[0], rs[1])
    gpu.Groth16.prove_with_randomness(ctx2, c1.z, rs[0], rs[1])
    ctx2.set_r1cs(gpu.R1CS.from_circuit(c2))                  # ... then the right one: slots must be rebuilt
    for i in range(4):
        got = gpu.Groth16.prove_with_randomness(ctx2, c2.z, rs[2 * i], rs[2 * i + 1])
        assert got == O.groth16_prove(c2, pk2, rs[2 * i], rs[2 * i + 1])
        assert O.groth16_verify(curve, pk2, c2.z[1:c2.P], got) == 1


def test_contexts_created_and_dropped_between_captured_proofs(gpu):
    """Regression: a ProvingContext that has captured its proof graph is dropped, then another context captures
    and replays its own -- the HIP runtime used to crash in hipGraphLaunch when the first context's capture
    stream had been destroyed (proof-slot streams are pooled for the life of the process since)."""
    import gc
    curve = 0
    c = synth.make_circuit(curve, 200, 300, 5, seed=51)
    pk = O.groth16_setup(c, H.toxic(curve, seed=10))
    rs = H.rand_fr_mont(curve, 2, seed=78)
    want = O.groth16_prove(c, pk, rs[0], rs[1])
    for _ in range(6):
        ctx = gpu.ProvingContext(curve, pk)
        ctx.set_r1cs(gpu.R1CS.from_circuit(c))
        for _ in range(4):  # eager, eager, capture, replay
            assert gpu.Groth16.prove_with_randomness(ctx, c.z, rs[0], rs[1]) == want
        del ctx
        gc.collect()


@pytest.mark.parametrize("curve,k", [(0, 1), (0, 3), (0, 8), (1, 2)])
def test_prove_batch_matches_single_proofs_and_oracle(gpu, curve, k):
    """mg_groth16_prove_batch: k different assignments of one circuit, different (r, s) each, one pass of the GPU
    pipeline -- every proof byte-identical to the oracle's (and so to mg_groth16_prove)."""
    c0 = synth.make_circuit(curve, 300, 260, 4, seed=60)
    pk = O.groth16_setup(c0, H.toxic(curve, seed=11))
    ctx = gpu.ProvingContext(curve, pk)
    ctx.set_r1cs(gpu.R1CS.from_circuit(c0))
    zs = [synth.reassign(c0, seed=200 + q) for q in range(k)]
    rs = H.rand_fr_mont(curve, 2 * k, seed=79)
    for rep in range(4):  # eager, eager, capture, replay
        got = gpu.Groth16.prove_batch(ctx, np.stack([z.z for z in zs]), rs[:k], rs[k:])
        assert len(got) == k
        for q in range(k):
            want = O.groth16_prove(zs[q], pk, rs[q], rs[k + q])
            assert got[q] == want, (rep, q)
            assert O.groth16_verify(curve, pk, zs[q].z[1:c0.P], got[q]) == 1
    # interleave with single proofs on the same context (different slot pool)
    assert gpu.Groth16.prove_with_randomness(ctx, zs[0].z, rs[0], rs[k]) == O.groth16_prove(zs[0], pk, rs[0], rs[k])


def test_prove_batch_longer_than_a_pass_is_streamed(gpu):
    """k = 70 > 32: the library streams the batch as passes of 32, 32 and 6 with three in flight on their own slots;
    every proof is still the bytes of the single-proof path, and a sample is checked against the oracle."""
    curve = 0
    c0 = synth.make_circuit(curve, 300, 260, 4, seed=61)
    pk = O.groth16_setup(c0, H.toxic(curve, seed=15))
    ctx = gpu.ProvingContext(curve, pk)
    ctx.set_r1cs(gpu.R1CS.from_circuit(c0))
    k = 70
    zs = [synth.reassign(c0, seed=300 + q) for q in range(k)]
    rs = H.rand_fr_mont(curve, 2 * k, seed=81)
    zk = np.stack([z.z for z in zs])
    single = [gpu.Groth16.prove_with_randomness(ctx, zs[q].z, rs[q], rs[k + q]) for q in range(k)]
    for rep in range(4):
        got = gpu.Groth16.prove_batch(ctx, zk, rs[:k], rs[k:])
        assert got == single, rep
    for q in (0, 31, 32, 63, 64, 69):
        assert single[q] == O.groth16_prove(zs[q], pk, rs[q], rs[k + q]), q


def test_concurrent_single_proofs_are_coalesced_and_stay_byte_identical(gpu):
    """Six host threads proving one transfer at a time on ONE context (the reference's simulation does exactly this,
    manta-pay/src/bin/simulation.rs:36-38): the library groups whatever calls are waiting into batched passes (sizes
    rounded up to a power of two). Every proof must still be the bytes of the oracle for its own (z, r, s), whichever
    calls it shared a pass with; one member has r = 0."""
    import threading
    curve = 0
    c0 = synth.make_circuit(curve, 300, 260, 4, seed=62)
    pk = O.groth16_setup(c0, H.toxic(curve, seed=16))
    ctx = gpu.ProvingContext(curve, pk)
    ctx.set_r1cs(gpu.R1CS.from_circuit(c0))
    n_threads, per = 6, 12
    n = n_threads * per
    zs = [synth.reassign(c0, seed=400 + q) for q in range(n)]
    rs = H.rand_fr_mont(curve, 2 * n, seed=82).copy()
    rs[5] = 0
    want = [O.groth16_prove(zs[q], pk, rs[q], rs[n + q]) for q in range(n)]
    got = [None] * n
    errs = []

    def worker(t):
        try:
            for j in range(per):
                q = t * per + j
                got[q] = gpu.Groth16.prove_with_randomness(ctx, zs[q].z, rs[q], rs[n + q])
        except Exception as e:  # pragma: no cover
            errs.append(e)
    for rep in range(3):  # slots of every batch size go through eager, eager, capture, replay
        ts = [threading.Thread(target=worker, args=(t,)) for t in range(n_threads)]
        [t.start() for t in ts]
        [t.join() for t in ts]
        assert not errs, errs
        assert got == want, rep


def test_prove_batch_real_shape_with_r_zero_member(gpu):
    """ToPrivate-shape circuit, batch of 4 with one member's r = 0 (that member skips g1_b like create_proof)."""
    curve = 0
    c = synth.make_shape(curve, "to_private")
    pk = O.groth16_setup(c, H.toxic(curve, seed=12))
    ctx = gpu.ProvingContext(curve, pk)
    ctx.set_r1cs(gpu.R1CS.from_circuit(c))
    k = 4
    rs = H.rand_fr_mont(curve, 2 * k, seed=80).copy()
    rs[2] = 0
    zs = np.stack([c.z] * k)
    got = gpu.Groth16.prove_batch(ctx, zs, rs[:k], rs[k:])
    for q in range(k):
        assert got[q] == O.groth16_prove(c, pk, rs[q], rs[k + q]), q
    with pytest.raises(gpu.MantaGpuError):
        gpu.Groth16.prove_batch(ctx, np.stack([c.z] * 2000), np.zeros((2000, 4), np.uint64), np.zeros((2000, 4), np.uint64))


def test_prove_bls12_381_at_circuit_size(gpu):
    """BLS12-381 (the curve of BASELINE's synthetic configs) at a real circuit size -- D = 2^15, V = 2^15 - 200:
    the 14-limb reduced-radix G1/G2 kernels, precomputed c = 8 tables, hipGraph replay and a batch of 3, all
    byte-identical to the oracle and pairing-verified. (D = V = 2^20 is `tools/config3_bls_2_20.py`.)"""
    from manta_rs_amd import keygen
    curve = 1
    D = 1 << 15
    c = synth.make_circuit(curve, D - 16, D - 200, 16, seed=0x4D414E5441_0005)
    assert c.D == D
    pk = keygen.generate(c, synth.from_mont(H.toxic(curve, seed=13), synth.FR_MODULUS[curve]))
    ctx = gpu.ProvingContext(curve, pk)
    ctx.set_r1cs(gpu.R1CS.from_circuit(c))
    rs = H.rand_fr_mont(curve, 6, seed=555)
    want = O.groth16_prove(c, pk, rs[0], rs[1])
    for _ in range(4):  # eager, eager, capture, replay
        assert gpu.Groth16.prove_with_randomness(ctx, c.z, rs[0], rs[1]) == want
    assert O.groth16_verify(curve, pk, c.z[1:c.P], want) == 1
    got = gpu.Groth16.prove_batch(ctx, np.stack([c.z] * 3), rs[0:6:2], rs[1:6:2])
    assert got[0] == want
    for q in (1, 2):
        assert got[q] == O.groth16_prove(c, pk, rs[2 * q], rs[2 * q + 1])
        assert O.groth16_verify(curve, pk, c.z[1:c.P], got[q]) == 1


def test_page_locked_assignment_is_uploaded_in_place(gpu):
    """An assignment kept in mg_host_alloc memory skips the library's staging copy; the proofs are the same
    bytes, single and batched, and an ordinary buffer still works afterwards."""
    curve = 0
    c = synth.make_circuit(curve, 500, 400, 6, seed=71)
    pk = O.groth16_setup(c, H.toxic(curve, seed=14))
    ctx = gpu.ProvingContext(curve, pk)
    ctx.set_r1cs(gpu.R1CS.from_circuit(c))
    rs = H.rand_fr_mont(curve, 8, seed=81)
    want = [O.groth16_prove(c, pk, rs[2 * q], rs[2 * q + 1]) for q in range(4)]
    zp = gpu.PinnedArray.like(c.z)
    for _ in range(4):
        assert gpu.Groth16.prove_with_randomness(ctx, zp.array, rs[0], rs[1]) == want[0]
    assert gpu.Groth16.prove_with_randomness(ctx, c.z, rs[2], rs[3]) == want[1]
    zk = gpu.PinnedArray.like(np.stack([c.z] * 4))
    for _ in range(4):
        assert gpu.Groth16.prove_batch(ctx, zk.array, rs[0:8:2], rs[1:8:2]) == want
    zp.free()
    zk.free()


@pytest.mark.parametrize("curve", [0, 1])
def test_groth16_setup_with_custom_generators(gpu, curve):
    """mg_groth16_setup with the group generators the reference would draw from its RNG (ark-groth16 0.3 samples
    random g1 / g2 in generate_random_parameters): a key over non-standard generators must still prove and verify,
    and with the standard generators it is the oracle's key point for point (test_gpu_keygen_matches_oracle_setup)."""
    from manta_rs_amd import keygen
    c = synth.make_circuit(curve, 300, 220, 5, seed=91)
    r = synth.FR_MODULUS[curve]
    k1 = synth.ints_to_limbs([0x1234567 % r], 4)[0]
    k2 = synth.ints_to_limbs([0x7654321 % r], 4)[0]
    g1 = O.g_mul(curve, 1, keygen.generator(curve, 1), k1)
    g2 = O.g_mul(curve, 2, keygen.generator(curve, 2), k2)
    toxic = synth.from_mont(H.toxic(curve, seed=15), r)
    pk = keygen.generate(c, toxic, g1, g2)
    assert not (pk.a_query == keygen.generate(c, toxic).a_query).all()
    ctx = gpu.ProvingContext(curve, pk)
    ctx.set_r1cs(gpu.R1CS.from_circuit(c))
    rs = H.rand_fr_mont(curve, 2, seed=82)
    proof = gpu.Groth16.prove_with_randomness(ctx, c.z, rs[0], rs[1])
    assert proof == O.groth16_prove(c, pk, rs[0], rs[1])
    assert O.groth16_verify(curve, pk, c.z[1:c.P], proof) == 1
    # error codes: tau inside the evaluation domain, gamma = 0
    bad = list(toxic)
    bad[0] = 1  # tau = 1 = w^0
    with pytest.raises(gpu.MantaGpuError):
        keygen.generate(c, bad)
    bad = list(toxic)
    bad[3] = 0  # gamma
    with pytest.raises(gpu.MantaGpuError):
        keygen.generate(c, bad)


def test_in_library_rccl_exchange_without_a_process_group(gpu):
    """mg_ctx_opts.exchange = MG_EXCHANGE_RCCL in a process without torch.distributed / any process group: the library finds
    librccl.so.1 itself (the copy the process already maps, else dlopen), builds a one-rank clique over the device list [0] with
    ncclCommInitAll and runs the partial-point exchange inside
    `mg_groth16_prove` -- bytes equal the oracle's and the host-exchange context's (caller
    manta-accounting/src/transfer/mod.rs:695-715 -> groth16.rs:589-600)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = r"""
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np
import oracle_lib as O, helpers as H
from manta_rs_amd import api, synth
import torch.distributed as dist
assert not dist.is_initialized()
api.init(0)
curve = api.BN254
c = synth.make_circuit(curve, 700, 500, 9, seed=11, profile="W")
pk = O.groth16_setup(c, H.toxic(curve))
r1cs = api.R1CS.from_circuit(c)
rs = H.rand_fr_mont(curve, 8, seed=5)
rc = api.ProvingContext(curve, pk, devices=[0], exchange=api.EXCHANGE_RCCL)
rc.set_r1cs(r1cs)
for i in range(4):
    assert api.Groth16.prove_with_randomness(rc, c.z, rs[2 * i], rs[2 * i + 1]) == O.groth16_prove(c, pk, rs[2 * i], rs[2 * i + 1]), i
zs = np.stack([c.z] * 3)
got = api.Groth16.prove_batch(rc, zs, rs[0:3], rs[3:6])
assert got == [O.groth16_prove(c, pk, rs[q], rs[3 + q]) for q in range(3)]
rc.close()
print("in-library rccl ok")
""" % (root, root)
    out = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=900, cwd=root)
    assert out.returncode == 0 and "in-library rccl ok" in out.stdout, out.stdout[-3000:] + out.stderr[-3000:]


def test_three_contexts_on_one_device_within_a_stated_budget(gpu):
    """A signer holds three contexts at once -- `MultiProvingContext {to_private, private_transfer, to_public}`,
    manta-accounting/src/transfer/canonical.rs:561-588: with `mg_ctx_opts.full_table_bytes` each states what it may spend on
    its full tables. Three contexts of 12 GB each on one device: every one stays inside its budget, keeps
    full tables (narrower windows than the default budget would buy), and proves the oracle's bytes; budget 0 = none."""
    from manta_rs_amd import keygen
    budget = 12 << 30
    ctxs = []
    for i, shape in enumerate(("to_private", "private_transfer", "to_public")):
        c = synth.make_shape(0, shape, profile="W")
        pk = keygen.generate(c, synth.from_mont(H.toxic(0, seed=60 + i), synth.FR_MODULUS[0]))
        ctx = gpu.ProvingContext(0, pk, full_table_bytes=budget)
        ctx.set_r1cs(gpu.R1CS.from_circuit(c))
        tb = ctx.table_bytes()
        assert 0 < tb[1] <= budget, (shape, tb)
        ctxs.append((c, pk, ctx))
    O.set_threads(O.usable_cpus())
    rs = H.rand_fr_mont(0, 2, seed=61)
    for c, pk, ctx in ctxs:  # all three alive at once
        want = O.groth16_prove(c, pk, rs[0], rs[1])
        for _ in range(3):
            assert gpu.Groth16.prove_with_randomness(ctx, c.z, rs[0], rs[1]) == want
    c, pk, _ = ctxs[0]
    none = gpu.ProvingContext(0, pk, full_table_bytes=0)
    none.set_r1cs(gpu.R1CS.from_circuit(c))
    assert none.table_bytes()[1] == 0 and none.table_bytes()[0] == ctxs[0][2].table_bytes()[0]
    assert gpu.Groth16.prove_with_randomness(none, c.z, rs[0], rs[1]) == O.groth16_prove(c, pk, rs[0], rs[1])
    none.close()
    for _, _, ctx in ctxs:
        ctx.close()


def test_checksum_is_never_skipped_by_a_device_list(gpu):
    """(advisor r3) `ProvingContext.decode(..., devices=[...], checksum=...)` used to fall through to the unchecked sharded
    loader: the digest is now verified in front of every placement (`mg_ctx_create_from_bytes_ex`)."""
    c = synth.make_circuit(0, 100, 70, 5, seed=3)
    pk = O.groth16_setup(c, H.toxic(0))
    data = _pk_bytes(0, pk)
    good = gpu.blake3(data)
    for devices in (None, [0], [0, 0]):
        ctx = gpu.ProvingContext.decode(0, data, devices=devices, checksum=good)
        ctx.close()
        with pytest.raises(gpu.MantaGpuError) as ei:
            gpu.ProvingContext.decode(0, data, devices=devices, checksum=bytes(32))
        assert ei.value.status == 6
    with pytest.raises(ValueError):
        gpu.ProvingContext.decode(0, data, devices=[0], checksum=b"short")


@pytest.mark.parametrize("curve", [0, 1])
def test_witness_map_matches_oracle_across_domain_sizes(gpu, curve):
    """h = R1CStoQAP::witness_map(z) against the oracle for EVERY domain size 2^6 .. 2^17 (W-profile circuits): each size takes
    its own split of the stages into passes, tile shapes, register / LDS kernels and column groups -- a first version of the
    round-4 register kernel was wrong from 2^15 up and passed the public-transform tests at 2^13 (this sweep caught it). The
    context only needs a key of the right lengths: the generators repeated."""
    O.set_threads(O.usable_cpus())
    g1, g2 = O.generator(curve, 1), O.generator(curve, 2)
    for lg in range(6, 18):
        D, P = 1 << lg, 5
        c = synth.make_circuit(curve, D - P, max(D // 2, 40), P, seed=lg, profile="W")

        class K:
            pass
        k = K()
        k.V, k.P = c.V, c.P
        for name, n, g in (("alpha_g1", 1, g1), ("beta_g1", 1, g1), ("delta_g1", 1, g1), ("beta_g2", 1, g2), ("delta_g2", 1, g2),
                           ("a_query", c.V, g1), ("b_g1_query", c.V, g1), ("b_g2_query", c.V, g2), ("h_query", D - 1, g1),
                           ("l_query", c.V - c.P, g1)):
            setattr(k, name, np.tile(np.asarray(g, dtype=np.uint64).reshape(1, -1), (n, 1)))
        ctx = gpu.ProvingContext(curve, k, full_table_bytes=0)
        ctx.set_r1cs(gpu.R1CS.from_circuit(c))
        assert (ctx.witness_map(c.z) == O.witness_map(c)).all(), lg
        ctx.close()


def test_hardware_queues_are_probed_and_slots_get_their_own(gpu):
    """Round 5: the library measures which of its streams share a hardware queue (csrc/queues.hip: a kernel on one stream waits,
    bounded, for a word a kernel on the other writes) and gives every single-proof slot three streams on three different queues.
    After the first ProvingContext the probe must have found the runtime's queues (GPU_MAX_HW_QUEUES, default 4, per priority
    level; at least three high-priority ones or the sets are not used), and proofs from four threads at once -- a lone slot,
    slots beside other passes, coalesced passes -- are the oracle's."""
    import threading
    from manta_rs_amd import keygen
    c = synth.make_circuit(0, 700, 500, 9, seed=77)
    pk = keygen.generate(c, synth.from_mont(H.toxic(0, seed=8), synth.FR_MODULUS[0]))
    ctx = gpu.ProvingContext(0, pk)
    ctx.set_r1cs(gpu.R1CS.from_circuit(c))
    normal, high = gpu.hw_queues()
    if (normal, high) == (0, 0) or high < 3:
        # a GPU shared with other processes (or GPU_MAX_HW_QUEUES < 3): the probe's bounded waits can report queues as shared; the
        # library then falls back to plain pooled streams -- the parity half below still runs
        import warnings
        warnings.warn("hardware-queue probe found (%d, %d) queues: stream sets not used on this box" % (normal, high))
    else:
        assert 1 <= normal <= 16 and 3 <= high <= 16, (normal, high)
    rs = H.rand_fr_mont(0, 2, seed=99)
    want = O.groth16_prove(c, pk, rs[0], rs[1])
    bad = []

    def worker():
        for _ in range(12):
            if gpu.Groth16.prove_with_randomness(ctx, c.z, rs[0], rs[1]) != want:
                bad.append(1)
    ts = [threading.Thread(target=worker) for _ in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not bad
    ctx.close()
