"""dev tool: per-kernel timeline of the LAST MSM in a rocprofv3 --kernel-trace database (rocpd sqlite) -- start, end and
duration of every launch from the digit kernel to the last reduce: where a latency-mode MSM spends its time.
usage: python tools/msm_timeline.py <results.db>"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end from kernels order by start"))
idx = [i for i, r in enumerate(rows) if "digits_kernel" in r[0]]
first = idx[-1]
t0 = rows[first][1]
tot = {}
for name, s, e in rows[first:]:
    short = name.replace("void mg::", "").split("(")[0][:52]
    print(f"{(s - t0)/1e3:9.1f} {(e - t0)/1e3:9.1f} {(e - s)/1e3:8.1f} us  {short}")
    tot[short] = tot.get(short, 0) + (e - s) / 1e3
print("---- per kernel (us):", {k: round(v, 1) for k, v in tot.items()})
print("---- span (us):", round((rows[-1][2] - t0) / 1e3, 1), " sum of durations:", round(sum(tot.values()), 1))
