"""dev tool: 2^20 BLS12-381 G1 MSM over PLAIN bases (no precomputed multiples: what multi_scalar_mul(bases, scalars) literally
takes) for several window widths, one at a time and three in flight. usage: python tools/plain_sweep.py [c ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from manta_rs_amd import api, synth
api.init(0)
n = 1 << 20
q = synth.FQ_MODULUS[1]
G = synth.to_mont([0x17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb,
                   0x08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1], q, 6).reshape(-1)
kb = np.zeros((n, 4), dtype=np.uint64)
kb[:, 0] = np.uint64(12345) + np.arange(n, dtype=np.uint64) * np.uint64(977)
dp = api.fixed_base_mul(1, 1, G, api.DeviceBuffer.from_numpy(kb), n)
b = api.Bases(1, 1, (dp.ptr, n), precompute_window_bits=0, on_device=True)
d = api.DeviceBuffer.from_numpy(synth.msm_scalars(1, n, "U", seed=5))
ref = None
for c in [int(a) for a in sys.argv[1:]] or [0, 14, 16]:
    r = api.VariableBaseMSM.launch(b, d, n, window_bits=c).finish()
    if ref is None:
        ref = r
    assert (r == ref).all(), c
    for _ in range(2):
        api.VariableBaseMSM.launch(b, d, n, window_bits=c).finish()
    t = time.perf_counter(); k = 8
    for _ in range(k):
        api.VariableBaseMSM.launch(b, d, n, window_bits=c).finish()
    t1 = (time.perf_counter() - t) / k
    pend = []; t = time.perf_counter(); k = 12
    for _ in range(k):
        pend.append(api.VariableBaseMSM.launch(b, d, n, window_bits=c))
        if len(pend) == 3:
            pend.pop(0).finish()
    while pend:
        pend.pop(0).finish()
    t3 = (time.perf_counter() - t) / k
    print(f"plain bases, window_bits={c or 'default'}: one at a time {t1*1e3:6.2f} ms = {n/t1/1e6:6.1f} Mscalar/s | three in flight {t3*1e3:6.2f} ms = {n/t3/1e6:6.1f} Mscalar/s", flush=True)
