"""dev tool: a few single Groth16 verifications + one batch of 64 (for rocprofv3 --kernel-trace --stats)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
ps = bench.ProveSetup("to_private")
api = ps.api
proofs = ps.run(64, 1, 32)
vctx = api.VerifyingContext(ps.curve, ps.pk)
inputs = ps.c.z[1:ps.c.P]
pts = [api.proof_decode(ps.curve, p) for p in proofs]
assert api.groth16_verify(vctx, inputs, pts[0])
t = time.perf_counter()
for i in range(5): assert api.groth16_verify(vctx, inputs, pts[i])
print(f"single verify: {(time.perf_counter()-t)/5*1e3:.2f} ms")
import numpy as np
t = time.perf_counter()
assert api.groth16_verify_batch(vctx, np.stack([inputs] * 64), pts, np.arange(1, 129, dtype=np.uint64).reshape(64, 2))
print(f"batch of 64: {(time.perf_counter()-t)*1e3:.2f} ms")
