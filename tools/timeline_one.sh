#!/bin/bash
# dev tool: kernel timeline of a single proof; usage: tools/timeline_one.sh <outdir> <name> [ENV=VAL ...]
R=$PWD; O=$R/gpurun_out/$1; name=$2; shift 2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pp_$name
env "$@" PROVE_N=30 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pp_$name -o p -- python $R/tools/prove_profile.py > $O/prove_$name.txt 2>&1
db=$(find /tmp/pp_$name -name "*.db" | head -1)
python $R/tools/proof_timeline.py $db > $O/proof_timeline_$name.txt
grep "ms/proof" $O/prove_$name.txt
