#!/bin/bash
# dev tool (round 5): the MSM headline with K foreign streams created first, with / without the side stream of the bucket reduce
cd /root/repo
for k in 0 1 4; do for v in "MANTA_QUEUE_AWARE=0" "MANTA_QUEUE_AWARE=0 MANTA_RED_SIDE=0" "MANTA_QUEUE_AWARE=0 MANTA_RED_S=0"; do
  m=$(env $v python tools/precreate_bench.py $k --quick --no-cpu-baseline --workload msm 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f Mscalar/s (%.3f ms/step) one at a time %s' % (d['value'], d['ms_per_step'], d['config'].get('one_at_a_time_Mscalar_s')))")
  echo "K=$k [$v] $m"
done; done
