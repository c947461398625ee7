"""Ad-hoc MSM timing on the GPU box (not the bench): python tools/msm_time.py [log_n ...]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from manta_rs_amd import api, synth

api.init(0)
curve = int(os.environ.get("CURVE", "1"))
group = int(os.environ.get("GROUP", "1"))
logs = [int(a) for a in sys.argv[1:]] or [16, 20]
G1 = {0: (1, 2)}
p = synth.FR_MODULUS[curve]
# generator in Montgomery limbs via the library itself: [1]G from a 1-element fixed-base... needs the generator;
# take it from params: use the oracle-free closed form: BN254 G1 = (1,2)
def gen(curve, group):
    q = synth.FQ_MODULUS[curve]
    nl = synth.FQ_LIMBS[curve]
    if curve == 0 and group == 1:
        xs = [1, 2]
    elif curve == 1 and group == 1:
        xs = [0x17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb,
              0x08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1]
    else:
        raise SystemExit("G2 timing: use tests")
    return synth.to_mont(xs, q, nl).reshape(-1)

G = gen(curve, group)
for lg in logs:
    n = 1 << lg
    ks = np.zeros((n, 4), dtype=np.uint64)
    ks[:, 0] = np.arange(n, dtype=np.uint64) * np.uint64(0x9E3779B1) + np.uint64(12345)
    t = time.time()
    dks = api.DeviceBuffer.from_numpy(ks)
    dpts = api.fixed_base_mul(curve, group, G, dks, n)
    api.synchronize()
    print(f"n=2^{lg}: base generation {time.time()-t:.3f}s", flush=True)
    for dist in ("U", "W"):
        sc = synth.msm_scalars(curve, n, dist)
        dsc = api.DeviceBuffer.from_numpy(sc)
        for pre in (0, 13 if lg <= 18 else 16):
            t = time.time()
            b = api.Bases(curve, group, (dpts.ptr, n), precompute_window_bits=pre, on_device=True)
            api.synchronize()
            tb = time.time() - t
            for wb in ([0] if pre else [0, 12, 16]):
                r0 = api.VariableBaseMSM.launch(b, dsc, n, window_bits=wb).finish()
                ts = []
                for _ in range(3):
                    t = time.time()
                    r = api.VariableBaseMSM.launch(b, dsc, n, window_bits=wb).finish()
                    ts.append(time.time() - t)
                assert (r == r0).all()
                best = min(ts)
                print(f"  dist={dist} precompute_c={pre} window={wb}: {best*1e3:8.2f} ms  {n/best/1e6:8.2f} Mscalar/s  (bases {b.device_bytes()/1e6:.0f} MB, setup {tb:.2f}s) x={int(r[0])&0xffff:04x}", flush=True)
            b.close()
