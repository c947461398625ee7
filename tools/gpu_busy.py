"""dev tool: how busy was the GPU in the steady state of a rocprofv3 --kernel-trace run? Over the last `frac` of the trace:
share of time with at least one kernel running, time-weighted mean number of kernels running, and the same per kernel
family. python tools/gpu_busy.py <results.db> [frac=0.5]"""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = list(db.execute("select name, start, end, queue_id from kernels order by start"))
t_lo, t_hi = rows[0][1], max(r[2] for r in rows)
t0 = t_hi - (t_hi - t_lo) * frac
ev = []
fam = collections.Counter()
for name, s, e, q in rows:
    if e <= t0: continue
    s = max(s, t0)
    ev.append((s, 1)); ev.append((e, -1))
    fam[name.replace("void mg::", "").replace("mg::", "").split("<")[0].split("(")[0]] += e - s
ev.sort()
busy = area = 0; cur = 0; last = t0
for t, d in ev:
    if cur > 0: busy += t - last
    area += cur * (t - last)
    cur += d; last = t
span = t_hi - t0
print(f"window {span/1e6:.1f} ms: GPU busy {100*busy/span:.1f} %, mean kernels in flight {area/span:.2f}, queues seen {len(set(r[3] for r in rows))}")
for k, v in fam.most_common(12): print(f"  {k:28s} {100*v/span:6.1f} % of the window (summed durations)")
