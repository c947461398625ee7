"""dev tool: bench.py's verification leg alone (a proving context alive next to it, as in the bench)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from manta_rs_amd import api
api.init(0)
ps = bench.ProveSetup("private_transfer", "W")
proofs = [api.Groth16.prove_with_randomness(ps.ctx, ps.c.z, ps.rs[i][0], ps.rs[i][1]) for i in range(8)]
for _ in range(3):
    r = bench.verify_bench(ps, proofs)
    print(r["single_ms"], r["batch_ms"])
