#!/bin/bash
# dev tool: kernel timeline of ONE mg_groth16_verify (PrivateTransfer shape, BN254) under rocprofv3 --kernel-trace
R=$PWD; python tools/verify_profile.py 50 2>/dev/null | tail -1
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/vv
rocprofv3 --kernel-trace -d /tmp/vv -o v -- python $R/tools/verify_profile.py 3 > /dev/null 2>&1
python - <<PY
import sqlite3, glob
db = sqlite3.connect(glob.glob("/tmp/vv/**/*.db", recursive=True)[0])
rows = list(db.execute("select name, start, end, queue_id from kernels order by start"))
# the last verification: from its prepared-inputs MSM (digits kernel) or the Miller kernel of the two early pairs, whichever came first
idx = [i for i, r in enumerate(rows) if "digits_kernel" in r[0]][-1]
millers = [i for i, r in enumerate(rows) if "miller_kernel" in r[0]]
first = min(idx, millers[-2])
t0 = rows[first][1]
for n, s, e, q in rows[first:]:
    print(f"{(s-t0)/1e3:9.1f} {(e-t0)/1e3:9.1f} {(e-s)/1e3:8.1f} us q{q} {n.replace('void mg::','')[:60]}")
PY
