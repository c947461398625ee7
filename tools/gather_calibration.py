"""Calibration of the accumulate kernel's memory side. Runs the 2^20 BLS12-381 G1 MSM of bench.py
with the normal accumulate kernel and -- in a second process, MANTA_ACC_GATHER_ONLY=1 -- with its gather-only twin
(same sorted stream, same 128 B base-record gathers, no field arithmetic), for several window widths. Prints the
HIP-event duration of that kernel per launch. The twin exists only in a -DMG_CALIBRATION build of the library:
    tools/build_variant.sh calib "-DMG_CALIBRATION" msm_bls381_g1        (here, before gpurun)
which this script selects through MANTA_LIB; the shipped libmantagpu.so does not read MANTA_ACC_GATHER_ONLY.
usage: python tools/gather_calibration.py [c ...]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, json
sys.path.insert(0, %r)
import numpy as np
from manta_rs_amd import api, synth
c = int(sys.argv[1]); n = 1 << 20
api.init(0)
q = synth.FQ_MODULUS[1]
G = synth.to_mont([0x17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb, 0x08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1], q, 6).reshape(-1)
kb = np.zeros((n, 4), dtype=np.uint64); kb[:, 0] = np.uint64(12345) + np.arange(n, dtype=np.uint64) * np.uint64(977)
dp = api.fixed_base_mul(1, 1, G, api.DeviceBuffer.from_numpy(kb), n)
b = api.Bases(1, 1, (dp.ptr, n), precompute_window_bits=c, on_device=True)
d = api.DeviceBuffer.from_numpy(synth.msm_scalars(1, n, "U", seed=5))
api.set_kernel_timing(True)
ms = []
import time
tt = []
for i in range(8):
    t = time.perf_counter(); api.VariableBaseMSM.launch(b, d, n).finish(); tt.append(time.perf_counter() - t); ms.append(api.last_accumulate_ms())
print(json.dumps({"c": c, "gather_only": bool(os.environ.get("MANTA_ACC_GATHER_ONLY")), "kernel_ms": round(float(np.median(ms[2:])), 4),
                  "msm_ms": round(float(np.median(tt[2:])) * 1e3, 4), "table_bytes": b.device_bytes()}))
''' % ROOT
for c in [int(a) for a in sys.argv[1:]] or [16]:
    for g in (0, 1):
        env = dict(os.environ)
        if g:
            lib = os.path.join(ROOT, "manta_rs_amd", "lib", "libmantagpu_calib.so")
            if not os.path.exists(lib):
                sys.exit("no calibration build: run tools/build_variant.sh calib \"-DMG_CALIBRATION\" msm_bls381_g1 first")
            env["MANTA_LIB"] = lib
            env["MANTA_ACC_GATHER_ONLY"] = "1"
        r = subprocess.run([sys.executable, "-c", CHILD, str(c)], env=env, capture_output=True, text=True)
        print(r.stdout.strip() or r.stderr[-500:], flush=True)
