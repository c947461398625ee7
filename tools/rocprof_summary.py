#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats result database (rocpd sqlite) as a text table.
usage: python tools/rocprof_summary.py <results.db> [> profiles/<name>.txt]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
print(f"{'kernel':<78} {'calls':>6} {'total_ms':>12} {'avg_us':>10} {'%':>6}")
for name, calls, total, avg, pct in rows:
    n = name.replace("void ", "")
    if len(n) > 76:
        n = n[:73] + "..."
    print(f"{n:<78} {int(calls):>6} {float(total)/1e3:>12.2f} {float(avg):>10.1f} {float(pct):>6.2f}")
