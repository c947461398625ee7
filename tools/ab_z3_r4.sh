#!/bin/bash
# dev tool (round 4): A/B of the combined MSM without its sort (one digit launch per query) and of the early host fold
# -> gpurun_out/$1/ab_z3.txt (sequential proofs on the three witness profiles, alternating, two rounds)
R=$PWD; O=$R/gpurun_out/${1:-abz3}; mkdir -p $O
: > $O/ab_z3.txt
for round in 1 2; do
  for v in "old:MANTA_Z3_SORT=1 MANTA_Z3_EARLY=0" "nosort:MANTA_Z3_SORT=0 MANTA_Z3_EARLY=0" "early:MANTA_Z3_SORT=1 MANTA_Z3_EARLY=1" "new:MANTA_Z3_SORT=0 MANTA_Z3_EARLY=1"; do
    name=${v%%:*}; envs=${v#*:}
    echo "## $name ($envs) round $round" >> $O/ab_z3.txt
    env $envs SEQ_ONLY=1 CHECK=$([ $round = 1 ] && echo 1 || echo 0) timeout 400 python tools/profile_proofs.py sparse,W,dense 2>&1 | grep -E "==|oracle|sequential|host side|differ|Error|error|Traceback|assert" >> $O/ab_z3.txt
  done
done
cat $O/ab_z3.txt
