"""dev tool: how the CPU baseline (oracle, arkworks-`parallel`-style decomposition) scales with threads on this host."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O, helpers as H
from manta_rs_amd import synth
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "usable", O.usable_cpus())
for f in ("/sys/fs/cgroup/cpu.max", "/sys/devices/system/cpu/cpu0/topology/thread_siblings_list"):
    try: print(f, open(f).read().strip())
    except OSError as e: print(f, e)
c = synth.make_shape(0, "private_transfer")
pk = O.groth16_setup(c, H.toxic(0))
rs = H.rand_fr_mont(0, 2)
for th in (1, 8, 16, 32, 64, 128, 256):
    O.set_threads(th)
    t = time.time(); O.groth16_prove(c, pk, rs[0], rs[1]); print("threads", th, "proof_s", round(time.time() - t, 3), flush=True)
