#!/bin/bash
# dev tool (round 4): the MSM headline with two builds of the library, alternating on one box
for r in 1 2 3; do
  for lib in "$@"; do
    MANTA_LIB=$PWD/$lib MANTA_BENCH_NO_PMC=1 timeout 600 python bench.py --workload msm --quick --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
l=json.loads(sys.stdin.read()); print('$lib', 'r$r', 'pipelined', l['value'], 'Mscalar/s; step', l['ms_per_step'], 'ms; accumulate alone', l['roofline']['kernel_ms'], 'ms; one at a time', l['config']['latency_mode'])"
  done
done
