"""dev tool (round 5): bench.py's config2 leg alone (one 2^20 BLS12-381 proof, W witness) -- prove_ms, median, phases.
usage: [MANTA_GRAPH=split ...] python tools/config2_ab.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from manta_rs_amd import api
api.init(0)
r = bench.config2_bench(None)
print(json.dumps({k: r[k] for k in ("prove_ms", "prove_ms_median", "phases_ms", "setup_s")}))
