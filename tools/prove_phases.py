"""dev tool: phase split (HIP events, plain launches) of single proofs of one shape, next to the graph-replayed time.
usage: python tools/prove_phases.py [shape]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from manta_rs_amd import api, synth, keygen
api.init(0)
shape = sys.argv[1] if len(sys.argv) > 1 else "private_transfer"
curve = 0
p = synth.FR_MODULUS[curve]
c = synth.make_shape(curve, shape)
rng = synth.XorShift(5)
pk = keygen.generate(c, [rng.field(p) for _ in range(5)])
ctx = api.ProvingContext(curve, pk)
ctx.set_r1cs(api.R1CS.from_circuit(c))
rs = synth.to_mont([rng.field(p) for _ in range(2)], p, 4)
for _ in range(3):
    api.Groth16.prove_with_randomness(ctx, c.z, rs[0], rs[1])
t = time.perf_counter()
for _ in range(20):
    api.Groth16.prove_with_randomness(ctx, c.z, rs[0], rs[1])
print(f"{shape}: {(time.perf_counter()-t)/20*1e3:.3f} ms/proof sequential (graph replay)")
api.set_kernel_timing(True)
acc = {}
for i in range(8):
    t = time.perf_counter()
    api.Groth16.prove_with_randomness(ctx, c.z, rs[0], rs[1])
    dt = (time.perf_counter() - t) * 1e3
    ph = api.last_prove_phases_ms()
    ph["wall"] = dt
    if i >= 2:
        for k, v in ph.items():
            acc.setdefault(k, []).append(v)
api.set_kernel_timing(False)
print({k: round(sorted(v)[len(v) // 2], 3) for k, v in acc.items()})
