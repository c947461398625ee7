"""dev tool: hammer TWO ProvingContexts (W and dense witness profiles: different pair counts, both on z3 slots) from several host
threads each and check every proof's bytes against the context's own first (oracle-checked) proofs.
usage: python tools/stress_two_contexts.py [threads_per_context] [proofs_per_thread]"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from manta_rs_amd import api, synth, keygen
import oracle_lib as O
api.init(0)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 2
N = int(sys.argv[2]) if len(sys.argv) > 2 else 300
curve = 0
p = synth.FR_MODULUS[curve]
O.set_threads(O.usable_cpus())
ctxs = []
for prof in ("W", "dense"):
    c = synth.make_shape(curve, "private_transfer", profile=prof)
    rng = synth.XorShift(5)
    pk = keygen.generate(c, [rng.field(p) for _ in range(5)])
    ctx = api.ProvingContext(curve, pk)
    ctx.set_r1cs(api.R1CS.from_circuit(c))
    rs = synth.to_mont([rng.field(p) for _ in range(8)], p, 4)
    want = [api.Groth16.prove_with_randomness(ctx, c.z, rs[2 * k], rs[2 * k + 1]) for k in range(4)]
    assert want[0] == O.groth16_prove(c, pk, rs[0], rs[1]), prof
    ctxs.append((ctx, c, rs, want))
bad = []
def worker(ci, tid):
    ctx, c, rs, want = ctxs[ci]
    try:
        for i in range(N):
            k = (tid + i) % 4
            if api.Groth16.prove_with_randomness(ctx, c.z, rs[2 * k], rs[2 * k + 1]) != want[k]:
                bad.append((ci, tid, i))
    except Exception as e:  # noqa
        bad.append((ci, tid, repr(e)))
for tpc in sorted({1, T}):
    t0 = time.perf_counter()
    th = [threading.Thread(target=worker, args=(ci, t)) for ci in range(2) for t in range(tpc)]
    [t.start() for t in th]
    [t.join() for t in th]
    dt = time.perf_counter() - t0
    print(f"contexts=2 threads_per_context={tpc} proofs={2*tpc*N} {2*tpc*N/dt:.1f} proofs/s nbad={len(bad)} {bad[:4]}", flush=True)
sys.exit(1 if bad else 0)
