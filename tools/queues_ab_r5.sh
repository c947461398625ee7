#!/bin/bash
# dev tool (round 5): the MSM headline (bench.py --quick: 2^20 BLS12-381 G1, three in flight) per PROCESS, K foreign normal-priority
# streams created before the library's, with the queue-aware stream pools on / off (MANTA_QUEUE_AWARE)
cd $(dirname $0)/..
for k in ${@:-0 1 4 5}; do for v in "" "MANTA_QUEUE_AWARE=0"; do
  m=$(env $v python tools/precreate_bench.py $k --quick --no-cpu-baseline --workload msm 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f Mscalar/s (%.3f ms/step) one at a time %s' % (d['value'], d['ms_per_step'], d['config'].get('one_at_a_time_Mscalar_s')))")
  echo "K=$k [${v:-default}] $m"
done; done
