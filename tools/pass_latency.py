"""dev tool: latency of ONE batched pass of k proofs (PrivateTransfer shape), one pass in flight, and the rate with two callers;
where concurrent single calls end up when they are coalesced into passes of 2-8 (tools/single_threads_sweep.py)."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
ps = bench.ProveSetup("private_transfer")
api = ps.api
KS = [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4, 6, 8, 16, 32]
for k in KS:
    zs = ps.zk(k) if k > 1 else ps.z1_pin.array
    sel = list(range(k))
    def one():
        if k == 1:
            return api.Groth16.prove_with_randomness(ps.ctx, zs, ps.rs[0][0], ps.rs[0][1])
        return api.Groth16.prove_batch(ps.ctx, zs, ps.rs[sel, 0], ps.rs[sel, 1])
    for _ in range(6): one()
    t = time.perf_counter(); n = 40
    for _ in range(n): one()
    dt1 = (time.perf_counter() - t) / n
    def loop(cnt):
        for _ in range(cnt): one()
    ts = [threading.Thread(target=loop, args=(n,)) for _ in range(2)]
    t = time.perf_counter(); [x.start() for x in ts]; [x.join() for x in ts]; dt2 = (time.perf_counter() - t) / (2 * n)
    print(f"k={k:2d}: one pass in flight {dt1*1e3:6.2f} ms/pass = {k/dt1:7.0f} proofs/s | two callers {k/dt2:7.0f} proofs/s", flush=True)
