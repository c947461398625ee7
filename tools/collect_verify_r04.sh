#!/bin/bash
# dev tool: round-4 verification measurements on the GPU box -> gpurun_out/$1/ (run from the repo root): the driver-style bench line,
# kernel statistics of single / batched verifications, timelines, building blocks
R=$PWD; O=$R/gpurun_out/${1:-r04verify}; mkdir -p $O
timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err
tools/verify_timeline.sh > $O/verify_timeline.txt 2>&1
for i in 1 2 3; do python tools/verify_profile.py 100 2>/dev/null | tail -1; done >> $O/verify_timeline.txt
tools/verify_batch_timeline.sh > $O/verify_batch_timeline.txt 2>&1
[ -x tools/bin/ubench_pairing ] && timeout 100 tools/bin/ubench_pairing > $O/ubench_pairing.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pv; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pv -o v -- python $R/tools/verify_profile.py 50 > $O/verify_profiled.txt 2>> $O/bench.err
python $R/tools/rocprof_summary.py $(find /tmp/pv -name "*.db" | head -1) > $O/verify_kernel_stats.txt
rm -rf /tmp/pw; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pw -o v -- python $R/tools/verify_batch_profile.py 10 > $O/verify_batch_profiled.txt 2>> $O/bench.err
python $R/tools/rocprof_summary.py $(find /tmp/pw -name "*.db" | head -1) > $O/verify_batch_kernel_stats.txt
tail -c 300 $O/bench.err
