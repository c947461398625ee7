"""dev tool: PrivateTransfer-shape proofs on the three witness profiles -- sequential latency (graph replay), per-phase split
(HIP events, plain launches), batches of 256 distinct assignments, one byte comparison with the oracle per profile.
usage: python tools/profile_proofs.py [profiles=sparse,W,dense] [shape]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from manta_rs_amd import api, synth, keygen
profs = (sys.argv[1] if len(sys.argv) > 1 else "sparse,W,dense").split(",")
shape = sys.argv[2] if len(sys.argv) > 2 else "private_transfer"
check = os.environ.get("CHECK", "1") != "0"
api.init(0)
curve = 0
p = synth.FR_MODULUS[curve]
for prof in profs:
    c = synth.make_shape(curve, shape, profile=prof)
    rng = synth.XorShift(5)
    pk = keygen.generate(c, [rng.field(p) for _ in range(5)])
    t0 = time.perf_counter()
    ctx = api.ProvingContext(curve, pk)
    ctx.set_r1cs(api.R1CS.from_circuit(c))
    rs = synth.to_mont([rng.field(p) for _ in range(64)], p, 4).reshape(32, 2, 4)
    print(f"== {prof}: z {synth.histogram(c.z_int)} tables {ctx.table_bytes()} setup {time.perf_counter()-t0:.2f}s", flush=True)
    first = api.Groth16.prove_with_randomness(ctx, c.z, rs[0][0], rs[0][1])
    if check:
        import oracle_lib as O
        O.set_threads(O.usable_cpus())
        t = time.perf_counter()
        assert first == O.groth16_prove(c, pk, rs[0][0], rs[0][1]), "bytes differ from the oracle"
        print(f"   bytes == oracle ({time.perf_counter()-t:.2f}s on {O.usable_cpus()} threads)", flush=True)
    zpin = api.PinnedArray.like(c.z)
    for _ in range(4):
        api.Groth16.prove_with_randomness(ctx, zpin.array, rs[0][0], rs[0][1])
    n = 100
    reps = []
    for rep in range(5):
        t = time.perf_counter()
        for i in range(n):
            api.Groth16.prove_with_randomness(ctx, zpin.array, rs[i % 32][0], rs[i % 32][1])
        reps.append((time.perf_counter() - t) / n * 1e3)
    print("   host side of the last pass (ms):", api.last_pass_host_ms(), flush=True)
    print(f"   sequential {sorted(reps)[2]:.3f} ms/proof (median of 5 x {n}; min {min(reps):.3f} max {max(reps):.3f})", flush=True)
    if os.environ.get("SEQ_ONLY"):
        ctx.close(); zpin.free()
        continue
    api.set_kernel_timing(True)
    acc = {}
    for i in range(8):
        api.Groth16.prove_with_randomness(ctx, zpin.array, rs[0][0], rs[0][1])
        ph = api.last_prove_phases_ms()
        if i >= 2:
            for k, v in ph.items():
                acc.setdefault(k, []).append(v)
    api.set_kernel_timing(False)
    print("   phases", {k: round(sorted(v)[len(v) // 2], 3) for k, v in acc.items()}, flush=True)
    K = 256
    R = synth.Reassigner(c)
    zs = np.stack([c.z] + [R.assign(100 + q).z for q in range(1, 32)] * 8 + [c.z] * 7)[:K]
    zK = api.PinnedArray.like(zs)
    sel = [q % 32 for q in range(K)]
    for _ in range(2):
        api.Groth16.prove_batch(ctx, zK.array, rs[sel, 0], rs[sel, 1])
    t = time.perf_counter()
    nb = 3
    for _ in range(nb):
        api.Groth16.prove_batch(ctx, zK.array, rs[sel, 0], rs[sel, 1])
    dt = time.perf_counter() - t
    print(f"   batched {nb*K/dt:.0f} proofs/s ({dt/(nb*K)*1e3:.3f} ms/proof), one caller, 32 distinct assignments x 8", flush=True)
    for k in (2, 4, 8, 32):
        for _ in range(3):
            api.Groth16.prove_batch(ctx, zK.array[:k], rs[sel[:k], 0], rs[sel[:k], 1])
        t = time.perf_counter()
        for _ in range(10):
            api.Groth16.prove_batch(ctx, zK.array[:k], rs[sel[:k], 0], rs[sel[:k], 1])
        print(f"   pass of {k}: {(time.perf_counter()-t)/10*1e3:.3f} ms", flush=True)
    ctx.close(); zpin.free(); zK.free()
