import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from manta_rs_amd import api, synth
import oracle_lib as O, helpers as H
api.init(0)
curve, group = 1, 2
G = O.generator(curve, group)
ks = synth.ints_to_limbs([1, 2, 3, 5, 0xdeadbeef, synth.FR_MODULUS[1]-1], 4)
d = api.fixed_base_mul(curve, group, G, api.DeviceBuffer.from_numpy(ks), len(ks))
got = d.to_numpy(shape=(len(ks), 24))
for i in range(len(ks)):
    w = O.g_mul(curve, group, G, ks[i])
    print('fixed_base', i, (got[i] == w).all())
pts = H.random_points(curve, group, 4, seed=1)
print('host sum', (api.points_sum(curve, group, pts) == O.g_sum(curve, group, pts)).all())
for kk in ([1], [2], [3], [0x10001], [0xdeadbeefcafe]):
    sc = synth.ints_to_limbs(kk, 4)
    got = api.VariableBaseMSM.multi_scalar_mul(api.Bases(curve, group, pts[:1]), sc)
    print('msm n=1 k=', kk, (got == O.g_mul(curve, group, pts[0], sc[0])).all())
