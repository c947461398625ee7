"""dev tool: run a few PrivateTransfer-shape proofs (for rocprofv3 --kernel-trace --stats)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from manta_rs_amd import api, synth, keygen
api.init(0)
shape = sys.argv[1] if len(sys.argv) > 1 else "private_transfer"
curve = 0
p = synth.FR_MODULUS[curve]
c = synth.make_shape(curve, shape, profile=os.environ.get("PROFILE", "sparse"))
rng = synth.XorShift(5)
pk = keygen.generate(c, [rng.field(p) for _ in range(5)])
ctx = api.ProvingContext(curve, pk)
ctx.set_r1cs(api.R1CS.from_circuit(c))
rs = synth.to_mont([rng.field(p) for _ in range(2)], p, 4)
for _ in range(3):
    api.Groth16.prove_with_randomness(ctx, c.z, rs[0], rs[1])
t = time.perf_counter()
N = int(os.environ.get("PROVE_N", "10"))
for _ in range(N):
    api.Groth16.prove_with_randomness(ctx, c.z, rs[0], rs[1])
print(f"{shape}: {(time.perf_counter()-t)/N*1e3:.3f} ms/proof sequential")
