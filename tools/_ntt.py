import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
class E: rank=0; world=1
from manta_rs_amd import api
api.init(0)
print(json.dumps(bench.ntt_bench(E())))
