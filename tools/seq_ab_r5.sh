#!/bin/bash
# dev tool (round 5): sequential single-proof latency on the three witness profiles under environment variants, alternating.
# usage: tools/seq_ab_r5.sh "VAR=val ..." "VAR=val" ...   ("" = default)
cd $(dirname $0)/..
for rep in 1 2; do
  for v in "$@"; do
    r=$(env $v SEQ_ONLY=1 CHECK=${CHECK:-0} timeout 300 python tools/profile_proofs.py sparse,W,dense 2>&1 | grep -E "sequential|differ|Error" | awk '{printf "%s ", $2}')
    echo "rep $rep [${v:-default}] sparse / W / dense ms: $r"
  done
done
