#!/bin/bash
# dev tool (round 6): headline (three 2^20 MSMs in flight) per process with K foreign streams created first, under env variants
cd $(dirname $0)/..
ks=$1; shift
for k in $ks; do for v in "$@"; do
  m=$(env $v python tools/precreate_bench.py $k --quick --no-cpu-baseline --workload msm 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f Mscalar/s (%.3f ms/step) one at a time %s kernel %.3f ms' % (d['value'], d['ms_per_step'], d['config'].get('one_at_a_time_Mscalar_s'), d['roofline']['kernel_ms']))")
  echo "K=$k [${v:-default}] $m"
done; done
