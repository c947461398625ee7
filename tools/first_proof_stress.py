"""dev tool (round 6): the first proofs of a fresh process, many processes. One intermittent failure of
tests/test_gpu_ntt_prove.py::test_round5_knobs_do_not_change_results showed a wrong B element (A and C right) in the FIRST proof of a
child process. Spawns N children (the test's script, default environment), compares every PROOF / BATCH / CONC line with the oracle's
bytes and reports which element of which proof differed.   python tools/first_proof_stress.py [N=100] [ENV=VAL ...]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H, oracle_lib as O
from manta_rs_amd import synth, keygen
import test_gpu_ntt_prove as T
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
extra = dict(a.split("=", 1) for a in sys.argv[2:])
c = synth.make_shape(0, "to_public", profile="W")
pk = keygen.generate(c, synth.from_mont(H.toxic(0, seed=6), synth.FR_MODULUS[0]))
rs = H.rand_fr_mont(0, 4, seed=98)
rs[2][:] = 0
z2 = synth.Reassigner(c).assign(0x5EED).z
O.set_threads(O.usable_cpus())
two = [O.groth16_prove(c, pk, rs[0], rs[1]).hex(), O.groth16_prove(c, pk, rs[2], rs[3], z=z2).hex()]
def parts(h): return {"A": h[:64], "B": h[64:192], "C": h[192:]}
bad = 0
env = H.knob_env(extra, strip_prefix="MANTA_")
POLLUTE = "import torch\nxs = [torch.full((1 << 30,), 0x5A5A5A5A if i & 1 else -1, dtype=torch.int32, device='cuda') for i in range(60)]\ntorch.cuda.synchronize()\n"
for it in range(N):
    if os.environ.get("STRESS_POLLUTE"):  # 240 GB of HBM filled with non-zero words, then freed: the child's hipMalloc'ed memory is dirty
        subprocess.run([sys.executable, "-c", POLLUTE], capture_output=True, timeout=600)
    out = subprocess.run([sys.executable, "-c", T._R5_SCRIPT.format(root=ROOT)], env=env, capture_output=True, text=True, timeout=600)
    if out.returncode != 0:
        bad += 1
        print("iteration %d: child failed rc %d: %s" % (it, out.returncode, (out.stdout + out.stderr)[-300:]), flush=True)
        continue
    lines = out.stdout.split("\n")
    got = [ln.split()[1] for ln in lines if ln.startswith("PROOF")]
    for i, g in enumerate(got):
        w = two[i & 1]
        if g != w:
            bad += 1
            print("iteration %d: PROOF %d differs in %s" % (it, i, [k for k in "ABC" if parts(g)[k] != parts(w)[k]]), flush=True)
    for bi, ln in enumerate(l for l in lines if l.startswith("BATCH")):
        for q, g in enumerate(ln.split()[1:]):
            if g != two[q & 1]:
                bad += 1
                print("iteration %d: BATCH %d proof %d differs in %s" % (it, bi, q, [k for k in "ABC" if parts(g)[k] != parts(two[q & 1])[k]]), flush=True)
    for ln in lines:
        if ln.startswith("CONC"):
            _, j, g = ln.split()
            if g != two[int(j)]:
                bad += 1
                print("iteration %d: CONC proof differs in %s" % (it, [k for k in "ABC" if parts(g)[k] != parts(two[int(j)])[k]]), flush=True)
print("first_proof_stress: %d children, %d bad proofs" % (N, bad), flush=True)
