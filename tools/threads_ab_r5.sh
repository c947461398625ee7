#!/bin/bash
# dev tool (round 5): single proofs from 1..12 host threads on one context (tools/single_threads_sweep.py) under environment variants,
# alternating; one line per run: threads::proofs/s ...
# usage: tools/threads_ab_r5.sh "VAR=val ..." "VAR=val" ...   ("" = default)
cd $(dirname $0)/..
for rep in 1 2 3; do
  for v in "$@"; do
    r=$(env $v timeout 300 python tools/single_threads_sweep.py 2>&1 | grep -E "host threads|Error|assert" | sed -E 's/.*threads= *([0-9]+): *([0-9.]+) proofs.*/\1::\2/' | tr '\n' ' ')
    echo "rep $rep [${v:-default}] $r"
  done
done
