"""A/B timing of accumulate-kernel build variants (dev tool). Usage: MANTA_LIB=... MANTA_MSM_L=.. python tools/acc_variants.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from manta_rs_amd import api, synth
api.init(0)
curve, lg = 1, 20
n = 1 << lg
q = synth.FQ_MODULUS[curve]
G = synth.to_mont([0x17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb, 0x08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1], q, 6).reshape(-1)
ks = np.zeros((n, 4), dtype=np.uint64); ks[:, 0] = np.arange(n, dtype=np.uint64) * np.uint64(0x9E3779B1) + np.uint64(12345)
dpts = api.fixed_base_mul(curve, 1, G, api.DeviceBuffer.from_numpy(ks), n)
sc = synth.msm_scalars(curve, n, "U"); dsc = api.DeviceBuffer.from_numpy(sc)
pre = int(os.environ.get("PRE", "16"))
b = api.Bases(curve, 1, (dpts.ptr, n), precompute_window_bits=pre, on_device=True)
api.set_kernel_timing(True)
r0 = api.VariableBaseMSM.launch(b, dsc, n).finish()
ts, ks_ = [], []
for _ in range(5):
    t = time.time(); r = api.VariableBaseMSM.launch(b, dsc, n).finish(); ts.append(time.time() - t); ks_.append(api.last_accumulate_ms())
assert (r == r0).all()
print(f"{os.path.basename(os.environ.get('MANTA_LIB','default')):24s} L={os.environ.get('MANTA_MSM_L','auto'):>4s} pre={pre}: accumulate {np.mean(ks_):7.3f} ms  msm {min(ts)*1e3:7.2f} ms  x={int(r[0])&0xffff:04x}", flush=True)
