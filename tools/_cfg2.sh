# config2 (2^20 BLS12-381 proof) with 16- / 17-bit key tables, same box, alternating
for rep in 1 2; do for c in 16 17; do
MANTA_PROVE_C=$c python - <<PY
import sys, json; sys.argv=["bench.py"]
import bench
class E: world=1; rank=0
r = bench.config2_bench(E())
print("c=$c", r["prove_ms"], r["prove_ms_median"], {k: round(v,2) for k,v in r["phases_ms"].items()}, flush=True)
PY
done; done
