"""BASELINE config[2]: 2^20 Fr NTT + 2^20 G1/G2 MSMs -> one Groth16 proof on one MI355X (BLS12-381 by default).
Builds a synthetic D = 2^20 circuit, a valid proving key (host scalars + GPU fixed-base multiply), proves, and
checks the proof with the oracle's pairing (tests-side checker). Also times the 2^20 NTT variants in HBM.
usage: python tools/config3_bls_2_20.py [curve=1] [log_d=20]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from manta_rs_amd import api, synth, keygen
curve = int(sys.argv[1]) if len(sys.argv) > 1 else 1
lg = int(sys.argv[2]) if len(sys.argv) > 2 else 20
api.init(0)
p = synth.FR_MODULUS[curve]
D, P = 1 << lg, 16
out = {"curve": "BLS12-381" if curve else "BN254", "log_d": lg}
# ---- NTT micro-benchmark, data resident in HBM
x = np.random.RandomState(1).randint(0, 1 << 62, size=(D, 4), dtype=np.int64).astype(np.uint64); x[:, 3] &= np.uint64((1 << 60) - 1)
d = api.DeviceBuffer.from_numpy(x)
dom = api.Radix2EvaluationDomain(curve, D)
for name, inv, cos in (("fft", 0, 0), ("ifft", 1, 0), ("coset_fft", 0, 1), ("coset_ifft", 1, 1)):
    dom.fft_device(d, inv, cos)
    t = time.perf_counter(); reps = 10
    for _ in range(reps): dom.fft_device(d, inv, cos)
    dt = (time.perf_counter() - t) / reps
    out["ntt_%s_ms" % name] = round(dt * 1e3, 4)
    out["ntt_%s_Melem_per_s" % name] = round(D / dt / 1e6, 1)
    out["ntt_%s_algorithmic_GBps" % name] = round(D * 64 / dt / 1e9, 1)
print(json.dumps(out), flush=True)
# ---- circuit + key + proof
t = time.perf_counter()
c = synth.make_circuit(curve, D - P, D, P, seed=0x4D414E5441_0301)   # V = D = 2^20 variables
out["circuit_s"] = round(time.perf_counter() - t, 1); print(json.dumps(out), flush=True)
rng = synth.XorShift(0x4D414E5441_0302)
t = time.perf_counter()
pk = keygen.generate(c, [rng.field(p) for _ in range(5)])
out["keygen_s"] = round(time.perf_counter() - t, 1); print(json.dumps(out), flush=True)
t = time.perf_counter()
ctx = api.ProvingContext(curve, pk)
r1cs = api.R1CS.from_circuit(c)
ctx.set_r1cs(r1cs)
out["context_s"] = round(time.perf_counter() - t, 1)
rs = synth.to_mont([rng.field(p), rng.field(p)], p, 4)
proof = api.Groth16.prove_with_randomness(ctx, c.z, rs[0], rs[1])
ts = []
for _ in range(5):
    t = time.perf_counter(); pr = api.Groth16.prove_with_randomness(ctx, c.z, rs[0], rs[1]); ts.append(time.perf_counter() - t)
assert pr == proof
out["prove_ms"] = round(min(ts) * 1e3, 2); out["proofs_per_s"] = round(1 / min(ts), 2)
print(json.dumps(out), flush=True)
import oracle_lib as O  # checker only
t = time.perf_counter()
ok = O.groth16_verify(curve, pk, c.z[1:c.P], proof)
out["pairing_verified"] = (ok == 1); out["verify_s"] = round(time.perf_counter() - t, 2)
print(json.dumps(out), flush=True)
assert ok == 1
