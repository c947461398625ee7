"""dev tool: per-kernel PMC sums from a rocprofv3 --pmc run (rocpd sqlite output).
usage: python tools/pmc_kernels.py <results.db>"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
cand = [t for t in tabs if "pmc" in t.lower() or "counter" in t.lower()]
print("tables:", cand, file=sys.stderr)
view = "counters_collection" if "counters_collection" in tabs else None
if not view:
    print(tabs); sys.exit(1)
cols = [r[1] for r in db.execute(f"pragma table_info({view})")]
print("cols:", cols, file=sys.stderr)
rows = db.execute(f"select kernel_name, counter_name, count(*), sum(value) from {view} group by kernel_name, counter_name order by kernel_name")
for k, c, n, v in rows:
    print(f"{k[:70]:<70} {c:<18} launches={n:<5} sum={v:.4g} per_launch={v/n:.4g}")
