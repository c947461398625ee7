#!/bin/bash
# dev tool (round 4): witness-map NTT pass durations inside a serialised 32-proof pass, per library build / MANTA_NTT_R
R=$PWD
cd /tmp && export TMPDIR=/tmp
for cfg in "libmantagpu -" "libmantagpu 2" "libmantagpu_rrchain -" "libmantagpu_rrchain 0"; do
  set -- $cfg; lib=$1; r=$2
  export MANTA_LIB=$R/manta_rs_amd/lib/$lib.so
  if [ "$r" = "-" ]; then unset MANTA_NTT_R; else export MANTA_NTT_R=$r; fi
  rm -rf /tmp/ps; PROFILE=W MANTA_PROVE_STREAMS=1 MANTA_GRAPH=off timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ps -o q -- python $R/tools/prove_batch_profile.py 32 6 > /tmp/ps.txt 2>&1
  echo "== $lib R=$r $(grep k=32 /tmp/ps.txt)"
  python $R/tools/ntt_pass_times.py $(find /tmp/ps -name "*.db" | head -1) | awk '{s+=$NF; print} END {print "sum of mins", s}' FS="min="
  echo "2^20: $(python $R/tools/ntt_loop.py 20 30 2>&1 | grep '2^20' | awk '{printf "%s ", $5}')"
done
