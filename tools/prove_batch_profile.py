"""dev tool: a few batched PrivateTransfer-shape proving passes (for rocprofv3 --kernel-trace --stats).
usage: python tools/prove_batch_profile.py [k] [passes]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from manta_rs_amd import api, synth, keygen
api.init(0)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N = int(sys.argv[2]) if len(sys.argv) > 2 else 6
curve = 0
p = synth.FR_MODULUS[curve]
c = synth.make_shape(curve, "private_transfer", profile=os.environ.get("PROFILE", "sparse"))
rng = synth.XorShift(5)
pk = keygen.generate(c, [rng.field(p) for _ in range(5)])
ctx = api.ProvingContext(curve, pk)
ctx.set_r1cs(api.R1CS.from_circuit(c))
rs = synth.to_mont([rng.field(p) for _ in range(2 * K)], p, 4)
zs = np.ascontiguousarray(np.stack([c.z] * K))
for _ in range(3):
    api.Groth16.prove_batch(ctx, zs, rs[:K], rs[K:])
t = time.perf_counter()
for _ in range(N):
    api.Groth16.prove_batch(ctx, zs, rs[:K], rs[K:])
dt = (time.perf_counter() - t) / N
print(f"k={K}: {dt*1e3:.3f} ms per pass, {dt/K*1e3:.4f} ms per proof, {K/dt:.0f} proofs/s (one host thread)")
