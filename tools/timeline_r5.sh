#!/bin/bash
# dev tool (round 5): kernel timeline of the last single proof of a short run, per witness profile -> gpurun_out/timeline_<profile>.txt
cd $(dirname $0)/..; export TMPDIR=/tmp
for prof in ${1:-W}; do
  d=/tmp/tl_$prof; rm -rf $d
  PROFILE=$prof PROVE_N=6 rocprofv3 --kernel-trace -d $d -o t -- python tools/prove_profile.py > /tmp/tl_$prof.log 2>&1
  db=$(find $d -name "*.db" | head -1)
  python tools/proof_timeline.py $db > gpurun_out/timeline_$prof.txt 2>&1
  tail -3 /tmp/tl_$prof.log | grep sequential
done
