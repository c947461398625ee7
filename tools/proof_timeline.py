"""dev tool: per-kernel timeline of the LAST proof in a rocprofv3 --kernel-trace database (rocpd sqlite):
python tools/proof_timeline.py <results.db> [n_kernels_per_proof]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
view = "kernels" if "kernels" in tabs else None
if not view:
    print([t for t in tabs if "kernel" in t.lower()]); sys.exit(1)
cols = [r[1] for r in db.execute(f"pragma table_info({view})")]
rows = list(db.execute(f"select name, start, end, queue_id, stream_id from {view} order by start")) if "stream_id" in cols else \
       list(db.execute(f"select name, start, end, queue_id, 0 from {view} order by start"))
# the last proof = kernels after the last digits_kernel burst: find the last 5 digits_kernel launches
idx = [i for i, r in enumerate(rows) if "digits_kernel" in r[0]]
first = idx[-5]
# walk back to include the witness map of the same proof (spmv kernels just before)
while first > 0 and rows[first][1] - rows[first - 1][2] < 200_000 and "reduce_level1" not in rows[first - 1][0] and "tile_reduce" not in rows[first - 1][0]:
    first -= 1
t0 = rows[first][1]
for name, s, e, q, st in rows[first:]:
    short = name.replace("void mg::", "").split("(")[0][:44]
    print(f"{(s - t0)/1e3:9.1f} {(e - t0)/1e3:9.1f} {(e - s)/1e3:8.1f} us  q{q}  {short}")
