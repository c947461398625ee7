"""dev tool: kernel-by-kernel picture of the PIPELINED steady state in a rocprofv3 --kernel-trace database: every launch of
the last `ms` milliseconds before the end of the last accumulate kernel of the main timed loop, with its queue, start,
duration and the gap since the previous kernel of the same queue ended (a gap = the launch waited: for the host, or for
registers / SIMD slots held by another queue's kernel).  python tools/pipe_timeline.py <results.db> [ms=9] [skip_tail_ms=0]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
win = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 9e6
rows = list(db.execute("select name, start, end, queue_id from kernels order by start"))
acc = [r for r in rows if "accumulate_chunks" in r[0]]
# the pipelined loop comes first in bench.py --quick (then latency mode): take the window ending at 60 % of the accumulate launches
t_hi = acc[int(len(acc) * 0.55)][2]
t_lo = t_hi - win
last_end = {}
q_ids = {}
for name, s, e, q in rows:
    if e < t_lo - 3e6: continue
    gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = e
    if s < t_lo or s > t_hi: continue
    qi = q_ids.setdefault(q, len(q_ids))
    short = name.replace("void mg::", "").replace("mg::", "").split("<")[0].split("(")[0][:22]
    print(f"{(s - t_lo)/1e3:9.1f} us  q{qi}  {'  ' * qi * 6}{short:22s} {(e - s)/1e3:8.1f} us  gap {gap:8.1f}")
