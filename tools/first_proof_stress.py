"""dev tool (round 6): the first proofs of a fresh process, many processes. One intermittent failure of
tests/test_gpu_ntt_prove.py::test_round5_knobs_do_not_change_results showed a wrong B element (A and C right) in the FIRST proof of a
child process. Spawns N children (the test's script, default environment), compares every PROOF / BATCH / CONC line with the oracle's
bytes and reports which element of which proof differed.   python tools/first_proof_stress.py [N=100] [ENV=VAL ...]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H, oracle_lib as O
from manta_rs_amd import synth, keygen
import hashlib, threading
import test_gpu_ntt_prove as T
KEYS = ("alpha_g1", "beta_g1", "delta_g1", "beta_g2", "delta_g2", "a_query", "b_g1_query", "b_g2_query", "h_query", "l_query")
# the test's child script, plus a digest of every array of the key the child generated (is a wrong B the G2 MSM or the key?)
CHILD = T._R5_SCRIPT.replace("ctx = gpu.ProvingContext(0, pk)", "import hashlib\nfor _k in %r: print('KEY', _k, hashlib.sha256(np.ascontiguousarray(getattr(pk, _k)).tobytes()).hexdigest())\nctx = gpu.ProvingContext(0, pk)" % (KEYS,))
assert CHILD != T._R5_SCRIPT
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
extra = dict(a.split("=", 1) for a in sys.argv[2:])
c = synth.make_shape(0, "to_public", profile="W")
pk = keygen.generate(c, synth.from_mont(H.toxic(0, seed=6), synth.FR_MODULUS[0]))
rs = H.rand_fr_mont(0, 4, seed=98)
rs[2][:] = 0
z2 = synth.Reassigner(c).assign(0x5EED).z
O.set_threads(O.usable_cpus())
two = [O.groth16_prove(c, pk, rs[0], rs[1]).hex(), O.groth16_prove(c, pk, rs[2], rs[3], z=z2).hex()]
import numpy as np
key_want = {k: hashlib.sha256(np.ascontiguousarray(getattr(pk, k)).tobytes()).hexdigest() for k in KEYS}
def parts(h): return {"A": h[:64], "B": h[64:192], "C": h[192:]}


bad = [0]
lock = threading.Lock()
env = H.knob_env(extra, strip_prefix="MANTA_")
WORKERS = int(os.environ.get("STRESS_WORKERS", "1"))


def note(msg):
    with lock:
        bad[0] += 1
        print(msg, flush=True)


def one(it):
    if os.environ.get("STRESS_POLLUTE"):
        subprocess.run([sys.executable, "-c", POLLUTE], capture_output=True, timeout=600)
    out = subprocess.run([sys.executable, "-c", CHILD.format(root=ROOT)], env=env, capture_output=True, text=True, timeout=600)
    if out.returncode != 0:
        return note("iteration %d: child failed rc %d: %s" % (it, out.returncode, (out.stdout + out.stderr)[-300:]))
    lines = out.stdout.split("\n")
    for ln in lines:
        if ln.startswith("KEY"):
            _, k, h = ln.split()
            if h != key_want[k]:
                note("iteration %d: the child's KEY array %s differs from the parent's" % (it, k))
    got = [ln.split()[1] for ln in lines if ln.startswith("PROOF")]
    for i, g in enumerate(got):
        if g != two[i & 1]:
            note("iteration %d: PROOF %d differs in %s" % (it, i, [k for k in "ABC" if parts(g)[k] != parts(two[i & 1])[k]]))
    for bi, ln in enumerate(l for l in lines if l.startswith("BATCH")):
        for q, g in enumerate(ln.split()[1:]):
            if g != two[q & 1]:
                note("iteration %d: BATCH %d proof %d differs in %s" % (it, bi, q, [k for k in "ABC" if parts(g)[k] != parts(two[q & 1])[k]]))
    for ln in lines:
        if ln.startswith("CONC"):
            _, j, g = ln.split()
            if g != two[int(j)]:
                note("iteration %d: CONC proof differs in %s" % (it, [k for k in "ABC" if parts(g)[k] != parts(two[int(j)])[k]]))


def worker(wid):
    for it in range(wid, N, WORKERS):
        one(it)


POLLUTE = "import torch\nxs = [torch.full((1 << 30,), 0x5A5A5A5A if i & 1 else -1, dtype=torch.int32, device='cuda') for i in range(60)]\ntorch.cuda.synchronize()\n"
ts = [threading.Thread(target=worker, args=(w,)) for w in range(WORKERS)]
[t.start() for t in ts]
[t.join() for t in ts]
print("first_proof_stress: %d children (%d at a time), %d bad" % (N, WORKERS, bad[0]), flush=True)
