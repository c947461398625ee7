#!/bin/bash
# dev tool (round 4): kernel timeline of the 2^20 BLS12-381 proof (BASELINE configs[2]) -- the G2 chain
R=$PWD; O=$R/gpurun_out/${1:-r4m}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pc
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/pc -o c -- python $R/tools/config3_bls_2_20.py > $O/config2.txt 2>&1
db=$(find /tmp/pc -name "*.db" | head -1)
python $R/tools/rocprof_summary.py $db > $O/config2_kernel_stats.txt
python - "$db" > $O/config2_g2_timeline.txt <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end, queue_id from kernels order by start"))
# the last proof: kernels after the last spmv3
idx = [i for i, r in enumerate(rows) if "spmv3" in r[0]]
first = idx[-1] - 2
t0 = rows[first][1]
for name, s, e, q in rows[first:]:
    short = name.replace("void mg::", "").replace("mg::", "").split("(")[0][:60]
    if (e - s) > 20000: print(f"{(s - t0)/1e3:9.1f} {(e - t0)/1e3:9.1f} {(e - s)/1e3:8.1f} us  q{q}  {short}")
PY
tail -2 $O/config2.txt | cut -c1-300; head -16 $O/config2_kernel_stats.txt | cut -c1-60,78-130
