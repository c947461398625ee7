#!/bin/bash
# dev tool: kernel timeline of ONE mg_groth16_verify_batch (256 proofs) under rocprofv3 --kernel-trace
R=$PWD; python tools/verify_batch_profile.py 10 2>/dev/null | tail -1
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/vb
rocprofv3 --kernel-trace -d /tmp/vb -o v -- python $R/tools/verify_batch_profile.py 2 > /dev/null 2>&1
python - <<PY
import sqlite3, glob
db = sqlite3.connect(glob.glob("/tmp/vb/**/*.db", recursive=True)[0])
rows = list(db.execute("select name, start, end, queue_id from kernels order by start"))
fe = [i for i, r in enumerate(rows) if "final_exp_kernel" in r[0]]
lo, hi = fe[-2] + 1, fe[-1]
t0 = rows[lo][1]
for n, s, e, q in rows[lo:hi + 1]:
    print(f"{(s-t0)/1e3:9.1f} {(e-t0)/1e3:9.1f} {(e-s)/1e3:8.1f} us q{q} {n.replace('void mg::','')[:60]}")
PY
