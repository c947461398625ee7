"""dev tool (round 6): the work-efficient front levels of the bucket reduce INSIDE the captured graphs of a batched
pass (diagnosis twin) with the threshold lowered so that the pass's MSMs qualify (MANTA_RED_MIN): what failed in rounds 3-5?
Runs passes of K proofs -- eager, eager, capture, replay, replay -- and reports status, the library's error text and the first
differing proof against the oracle.   python tools/diag_front_in_graph.py [K=32] [shape=private_transfer]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import helpers as H, oracle_lib as O
from manta_rs_amd import api as gpu, synth, keygen
K = int(sys.argv[1]) if len(sys.argv) > 1 else 32
shape = sys.argv[2] if len(sys.argv) > 2 else "private_transfer"
gpu.init(0)
print("library", os.path.basename(gpu.LIB_PATH), {k: v for k, v in os.environ.items() if k.startswith("MANTA_") and k != "MANTA_LIB"}, flush=True)
c = synth.make_shape(0, shape, profile="W")
pk = keygen.generate(c, synth.from_mont(H.toxic(0, seed=6), synth.FR_MODULUS[0]))
rs = H.rand_fr_mont(0, 2, seed=98)
O.set_threads(O.usable_cpus())
want = O.groth16_prove(c, pk, rs[0], rs[1])
ctx = gpu.ProvingContext(0, pk)
ctx.set_r1cs(gpu.R1CS.from_circuit(c))
zs, r, s = np.stack([c.z] * K), np.stack([rs[0]] * K), np.stack([rs[1]] * K)
for rep, what in enumerate(("eager", "eager", "capture", "replay", "replay")):
    try:
        got = gpu.Groth16.prove_batch(ctx, zs, r, s)
    except gpu.MantaGpuError as e:
        print("pass %d (%s): ERROR %s" % (rep, what, e), flush=True)
        continue
    bad = [q for q in range(K) if got[q] != want]
    if bad:
        q = bad[0]
        parts = [("A", 0, 32), ("B", 32, 96), ("C", 96, 128)]
        print("pass %d (%s): %d of %d proofs differ; first %d: elements %s" % (rep, what, len(bad), K, q, [n for n, lo, hi in parts if got[q][lo:hi] != want[lo:hi]]), flush=True)
    else:
        print("pass %d (%s): %d proofs == oracle" % (rep, what, K), flush=True)
