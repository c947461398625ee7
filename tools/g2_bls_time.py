"""dev tool: 2^20 BLS12-381 G2 MSM (c = 17 tables) on uniform and witness-like scalars, closed-form checked; ms per MSM."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import oracle_lib as O
from manta_rs_amd import api, synth
api.init(0)
curve, n = 1, 1 << int(os.environ.get("LOGN", "20"))
p = synth.FR_MODULUS[curve]
s0, s1 = 0x7654321, 0x1fedcba987
kb = np.zeros((n, 4), dtype=np.uint64)
kb[:, 0] = np.uint64(s0) + np.arange(n, dtype=np.uint64) * np.uint64(s1)
G2 = O.generator(curve, 2)
dpts = api.fixed_base_mul(curve, 2, G2, api.DeviceBuffer.from_numpy(kb), n)
b = api.Bases(curve, 2, (dpts.ptr, n), precompute_window_bits=17, on_device=True)
base_k = [s0 + i * s1 for i in range(n)]
for dist in ("U", "W"):
    sc = synth.msm_scalars(curve, n, dist, seed=0x4D414E54 + (dist == "W"))
    d = api.DeviceBuffer.from_numpy(sc)
    got = api.VariableBaseMSM.launch(b, d, n, sparse=(dist == "W")).finish()
    t = sum(k * bk for k, bk in zip(synth.limbs_to_ints(sc), base_k)) % p
    ok = bool((got == O.g_mul(curve, 2, G2, synth.ints_to_limbs([t], 4)[0])).all())
    api.VariableBaseMSM.launch(b, d, n, sparse=(dist == "W")).finish()
    t0 = time.perf_counter()
    for _ in range(5):
        api.VariableBaseMSM.launch(b, d, n, sparse=(dist == "W")).finish()
    print(f"2^{n.bit_length()-1} BLS12-381 G2 MSM, {dist}: {(time.perf_counter()-t0)/5*1e3:.3f} ms, closed form {'OK' if ok else 'MISMATCH'}", flush=True)
