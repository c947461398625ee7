#!/bin/bash
# dev tool (round 4): A/B of library builds on the three witness profiles, alternating on one box
# usage: tools/ab_profiles_r4.sh <outdir> <libA.so> <libB.so> [rounds]
out=gpurun_out/$1; mkdir -p $out; A=$2; B=$3; n=${4:-2}
for r in $(seq 1 $n); do
  for lib in $A $B; do
    tag=$(basename $lib .so)
    MANTA_LIB=$PWD/$lib CHECK=0 timeout 300 python tools/profile_proofs.py sparse,W,dense 2>&1 | grep -E "==|sequential|batched|pass of 32" | sed "s/^/$tag r$r /" | tee -a $out/ab.txt | grep -E "sequential|batched" | awk '{printf "%s %s %s %s | ", $1, $2, $3, $4} END {print ""}'
  done
done
