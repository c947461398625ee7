"""dev tool: single verifications of a PrivateTransfer-shape proof (for rocprofv3 --kernel-trace; prints ms per verification).
CURVE=1: BLS12-381, a small circuit with the same number of public inputs (the verifier's work depends on nothing else)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from manta_rs_amd import api, synth, keygen
api.init(0)
curve = int(os.environ.get("CURVE", "0"))
p = synth.FR_MODULUS[curve]
c = synth.make_shape(curve, "private_transfer") if curve == 0 else synth.make_circuit(curve, 4096, 3000, 27, seed=5)
rng = synth.XorShift(5)
pk = keygen.generate(c, [rng.field(p) for _ in range(5)])
ctx = api.ProvingContext(curve, pk)
ctx.set_r1cs(api.R1CS.from_circuit(c))
rs = synth.to_mont([rng.field(p) for _ in range(2)], p, 4)
proof = api.proof_decode(curve, api.Groth16.prove_with_randomness(ctx, c.z, rs[0], rs[1]))
vctx = api.VerifyingContext(curve, pk)
inputs = c.z[1:c.P]
for _ in range(3):
    assert api.groth16_verify(vctx, inputs, proof)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
t = time.perf_counter()
for _ in range(N):
    api.groth16_verify(vctx, inputs, proof)
print(f"verify: {(time.perf_counter()-t)/N*1e3:.3f} ms")
