"""dev tool: hammer one ProvingContext from several host threads and check every proof's bytes.
usage: python tools/stress_prove.py [threads] [proofs_per_thread] [shape]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from manta_rs_amd import api, synth, keygen
api.init(0)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 4
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
shape = sys.argv[3] if len(sys.argv) > 3 else "private_transfer"
curve = 0
p = synth.FR_MODULUS[curve]
c = synth.make_shape(curve, shape)
rng = synth.XorShift(5)
pk = keygen.generate(c, [rng.field(p) for _ in range(5)])
ctx = api.ProvingContext(curve, pk)
ctx.set_r1cs(api.R1CS.from_circuit(c))
rs = synth.to_mont([rng.field(p) for _ in range(2 * 4)], p, 4)
want = [api.Groth16.prove_with_randomness(ctx, c.z, rs[2 * k], rs[2 * k + 1]) for k in range(4)]
assert len(set(want)) == 4
bad = []
def worker(tid):
    try:
        for i in range(N):
            k = (tid + i) % 4
            got = api.Groth16.prove_with_randomness(ctx, c.z, rs[2 * k], rs[2 * k + 1])
            if got != want[k]:
                bad.append((tid, i, "mismatch"))
    except Exception as e:  # noqa
        bad.append((tid, -1, repr(e)))
t0 = time.perf_counter()
th = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
[t.start() for t in th]
[t.join() for t in th]
dt = time.perf_counter() - t0
print(f"threads={T} proofs={T*N} {T*N/dt:.1f} proofs/s bad={bad[:5]} nbad={len(bad)}")
sys.exit(1 if bad else 0)
