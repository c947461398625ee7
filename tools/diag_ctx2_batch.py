"""dev tool (round 4): do the DEFAULT graphs (part A forked, G2 linear) of single AND batched passes survive the creation of a second
context? Every proof is split into A | B | C against the oracle. usage: python tools/diag_ctx2_batch.py [shape|small]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O, helpers as H
from manta_rs_amd import api, synth, keygen
api.init(0)
O.set_threads(O.usable_cpus())
curve = api.BN254
which = sys.argv[1] if len(sys.argv) > 1 else "small"
if which == "small":
    c = synth.make_circuit(curve, 700, 500, 9, seed=11); pk = O.groth16_setup(c, H.toxic(curve))
else:
    c = synth.make_shape(curve, which, profile="W"); pk = keygen.generate(c, synth.from_mont(H.toxic(curve, seed=6), synth.FR_MODULUS[curve]))
ctx = api.ProvingContext(curve, pk)
r1cs = api.R1CS.from_circuit(c)
ctx.set_r1cs(r1cs)
K = 8
rs = H.rand_fr_mont(curve, 2 * K, seed=5)
truth = [O.groth16_prove(c, pk, rs[q], rs[K + q]) for q in range(K)]
zs = np.stack([c.z] * K)
n = len(truth[0]); g1 = n // 4
def parts(p, t):
    return "".join("ok " if p[a:b] == t[a:b] else "BAD " for a, b in ((0, g1), (g1, 3 * g1), (3 * g1, n)))
def check(tag):
    one = parts(api.Groth16.prove_with_randomness(ctx, c.z, rs[0], rs[K]), truth[0])
    out = []
    for k in (2, 3, K):
        got = api.Groth16.prove_batch(ctx, zs[:k], rs[0:k], rs[K:K + k])
        out.append(f"k={k}: " + ("ok" if all(got[q] == truth[q] for q in range(k)) else "BAD[" + " | ".join(parts(got[q], truth[q]).strip() for q in range(k)) + "]"))
    print(f"{tag:<26} single: {one}| " + "  ".join(out), flush=True)
for i in range(4):
    check(f"run {i}")
c2 = api.ProvingContext(curve, pk); check("after create(2nd)")
c2.set_r1cs(r1cs); check("after set_r1cs(2nd)")
for i in range(3):
    api.Groth16.prove_with_randomness(c2, c.z, rs[0], rs[K]); check(f"after 2nd prove {i + 1}")
c3 = api.ProvingContext(curve, pk, full_table_bytes=0); c3.set_r1cs(r1cs); check("after 3rd ctx (no full)")
api.Groth16.prove_batch(c3, zs[:3], rs[0:3], rs[K:K + 3]); check("after 3rd prove_batch")
