"""dev tool: mg_groth16_verify_batch of 256 PrivateTransfer-shape proofs (for rocprofv3 --kernel-trace; prints ms per call)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from manta_rs_amd import api, synth, keygen
api.init(0)
curve = 0
p = synth.FR_MODULUS[curve]
c = synth.make_shape(curve, "private_transfer")
rng = synth.XorShift(5)
pk = keygen.generate(c, [rng.field(p) for _ in range(5)])
ctx = api.ProvingContext(curve, pk)
ctx.set_r1cs(api.R1CS.from_circuit(c))
k = int(sys.argv[2]) if len(sys.argv) > 2 else 256
rs = synth.to_mont([rng.field(p) for _ in range(2 * k)], p, 4)
proofs = api.Groth16.prove_batch(ctx, np.stack([c.z] * k), rs[:k], rs[k:])
pts = [api.proof_decode(curve, x) for x in proofs]
vctx = api.VerifyingContext(curve, pk)
inputs = np.stack([c.z[1:c.P]] * k)
rnd = np.random.RandomState(7).randint(1, 1 << 62, size=(k, 2)).astype(np.uint64)
for _ in range(3):
    assert api.groth16_verify_batch(vctx, inputs, pts, rnd)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10
t = time.perf_counter()
for _ in range(N):
    api.groth16_verify_batch(vctx, inputs, pts, rnd)
print(f"verify_batch({k}): {(time.perf_counter()-t)/N*1e3:.3f} ms")
