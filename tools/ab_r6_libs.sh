#!/bin/bash
# dev tool (round 6): batched proofs/s (bench.py's batched leg) per A/B LIBRARY build (tools/build_variant.sh tags; "" = the shipped one).
# usage: tools/ab_r6_libs.sh "<profiles>" "<tag> <tag> ..." [reps]
cd $(dirname $0)/..
profiles=$1; tags=$2; reps=${3:-2}
for rep in $(seq $reps); do
for prof in $profiles; do
  for t in $tags; do
    lib=manta_rs_amd/lib/libmantagpu_$t.so; [ "$t" = base ] && lib=manta_rs_amd/lib/libmantagpu.so
    r=$(MANTA_LIB=$PWD/$lib python bench.py --workload prove --child --batched-only --no-cpu-baseline --profile $prof 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.0f (min %.0f max %.0f)' % (d['batched']['proofs_per_s'], d['batched']['min'], d['batched']['max']))")
    echo "rep $rep  $prof  [$t]  $r proofs/s"
  done
done
done
