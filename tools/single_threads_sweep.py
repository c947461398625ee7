"""dev tool: mg_groth16_prove from several host threads on ONE context (PrivateTransfer shape), raw ctypes calls in a tight
loop (a few microseconds of Python per call, GIL released inside the library): how well concurrent single calls are
coalesced into batched passes (MANTA_COALESCE = passes in flight, 0 = off)."""
import ctypes, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
ps = bench.ProveSetup("private_transfer")
api = ps.api
LIB = api.LIB
z = ps.z1_pin.array
zp = z.ctypes.data_as(ctypes.c_void_p)
rs = ps.rs
want = api.Groth16.prove_with_randomness(ps.ctx, z, rs[0][0], rs[0][1])
r0, s0 = rs[0][0].ctypes.data_as(ctypes.c_void_p), rs[0][1].ctypes.data_as(ctypes.c_void_p)
for th in (1, 2, 3, 4, 6, 8, 12):
    n = 1200
    outs = [ctypes.create_string_buffer(len(want)) for _ in range(th)]
    def worker(t, count):
        for _ in range(count):
            rc = LIB.mg_groth16_prove(ps.ctx.handle, zp, r0, s0, outs[t])
            assert rc == 0
    for phase, count in (("warm", 60), ("timed", n // th)):
        ts = [threading.Thread(target=worker, args=(t, count)) for t in range(th)]
        t0 = time.perf_counter(); [t.start() for t in ts]; [t.join() for t in ts]; dt = time.perf_counter() - t0
    assert all(o.raw == want for o in outs)
    print(f"single proofs, host threads={th:2d}: {(n // th) * th / dt:8.1f} proofs/s", flush=True)
