"""dev tool (round 4): which element of a single proof goes wrong, and after which step of a second (sharded) context's life, when the
combined MSM runs outside the captured graph (A/B builds only). usage: DIAG=<mode> python tools/diag_z3.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, torch.distributed as dist
import oracle_lib as O, helpers as H
from manta_rs_amd import api, synth, distributed
torch.cuda.set_device(0)
api.init(0)
mode = os.environ.get("DIAG", "full")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
curve = api.BN254
c = synth.make_circuit(curve, 700, 500, 9, seed=11)
pk = O.groth16_setup(c, H.toxic(curve))
ctx = api.ProvingContext(curve, pk)
r1cs = api.R1CS.from_circuit(c)
ctx.set_r1cs(r1cs)
rs = H.rand_fr_mont(curve, 8, seed=5)
rs0 = rs.copy(); rs0[0][:] = 0  # r = 0: b_g1 unused
truth = O.groth16_prove(c, pk, rs[0], rs[1]); truth0 = O.groth16_prove(c, pk, rs0[0], rs0[1])
def parts(p, t):
    return "".join("ok " if p[a:b] == t[a:b] else "BAD " for a, b in ((0, 32), (32, 96), (96, 128)))  # A B C
def check(tag):
    a = parts(api.Groth16.prove_with_randomness(ctx, c.z, rs[0], rs[1]), truth)
    b = parts(api.Groth16.prove_with_randomness(ctx, c.z, rs0[0], rs0[1]), truth0)
    print(f"{tag:<28} ctx: {a} | r=0: {b}", flush=True)
for i in range(4):
    check(f"run {i}")
if mode == "second_plain":      # a second PLAIN context of the same key instead of the sharded prover
    c2 = api.ProvingContext(curve, pk); check("after create(plain)")
    c2.set_r1cs(r1cs); check("after set_r1cs(plain)")
    api.Groth16.prove_with_randomness(c2, c.z, rs[0], rs[1]); check("after prove(plain) 1")
    api.Groth16.prove_with_randomness(c2, c.z, rs[0], rs[1]); check("after prove(plain) 2")
    api.Groth16.prove_with_randomness(c2, c.z, rs[0], rs[1]); check("after prove(plain) 3")
else:
    sp = distributed.ShardedProver(curve, pk, force_collective=True, max_batch=3); check("after create(sp)")
    sp.set_r1cs(r1cs); check("after set_r1cs(sp)")
    for i in range(3):
        ok = parts(sp.prove(c.z, rs[0], rs[1]), truth); check(f"after sp.prove {i} ({ok.strip()})")
