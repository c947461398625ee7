"""dev tool: a few 2^20 BLS12-381 transforms on the device (for rocprofv3 --kernel-trace / --pmc runs) and a PrivateTransfer witness map.
usage: python tools/ntt_loop.py [log_n] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from manta_rs_amd import api, synth
api.init(0)
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
curve = 1
n = 1 << lg
rng = np.random.RandomState(4)
x = rng.randint(0, 1 << 62, size=(n, 4)).astype(np.uint64)
x[:, 3] &= np.uint64((1 << 60) - 1)
d = api.DeviceBuffer.from_numpy(x)
dom = api.Radix2EvaluationDomain(curve, n)
for inv, coset in ((False, False), (True, False), (False, True), (True, True)):
    dom.fft_device(d, inverse=inv, coset=coset)
    api.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        dom.fft_device(d, inverse=inv, coset=coset)
    api.synchronize()
    print(f"2^{lg} inverse={inv} coset={coset}: {(time.perf_counter()-t)/reps*1e3:.4f} ms per call (wall)")
