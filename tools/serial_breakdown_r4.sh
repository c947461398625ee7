#!/bin/bash
# dev tool (round 4): standalone kernel durations of a 32-proof pass -- every kernel on ONE stream, one pass in flight, no graphs
R=$PWD; O=$R/gpurun_out/${1:-r4f}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for prof in dense W sparse; do
  rm -rf /tmp/ps; PROFILE=$prof MANTA_PROVE_STREAMS=1 MANTA_GRAPH=off timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ps -o q -- python $R/tools/prove_batch_profile.py 32 6 > $O/serial_batch_${prof}.txt 2>&1
  python $R/tools/rocprof_summary.py $(find /tmp/ps -name "*.db" | head -1) > $O/serial_batch32_${prof}_kernel_stats.txt
  grep "k=32" $O/serial_batch_${prof}.txt
done
