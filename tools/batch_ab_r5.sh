#!/bin/bash
# dev tool (round 5): batched proofs/s (bench.py's batched leg: 256 proofs per call, two host threads, three passes in flight) under
# environment variants. usage: tools/batch_ab_r5.sh "<profiles>" "VAR=val VAR=val" "VAR=val" ...   (one quoted group per variant; "" = default)
cd $(dirname $0)/..
profiles=$1; shift
for rep in 1 2; do
for prof in $profiles; do
  for v in "$@"; do
    r=$(env $v python bench.py --workload prove --child --batched-only --no-cpu-baseline --profile $prof 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.0f (min %.0f max %.0f)' % (d['batched']['proofs_per_s'], d['batched']['min'], d['batched']['max']))")
    echo "rep $rep  $prof  [${v:-default}]  $r proofs/s"
  done
done
done
