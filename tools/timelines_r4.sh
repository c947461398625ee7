#!/bin/bash
# dev tool (round 4): kernel timelines of single proofs and 32-proof passes per witness profile
R=$PWD; O=$R/gpurun_out/${1:-r4c}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for prof in dense W; do
  for tab in full bucket; do
    if [ $tab = bucket ]; then export MANTA_FULL_TABLE_GB=0; else unset MANTA_FULL_TABLE_GB; fi
    rm -rf /tmp/pp; PROFILE=$prof PROVE_N=30 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pp -o p -- python $R/tools/prove_profile.py > $O/prove_${prof}_${tab}.txt 2>&1
    db=$(find /tmp/pp -name "*.db" | head -1)
    python $R/tools/rocprof_summary.py $db > $O/prove_${prof}_${tab}_kernel_stats.txt
    python $R/tools/proof_timeline.py $db > $O/proof_timeline_${prof}_${tab}.txt
  done
  unset MANTA_FULL_TABLE_GB
  rm -rf /tmp/pq; PROFILE=$prof timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pq -o q -- python $R/tools/prove_batch_profile.py > $O/prove_batch_${prof}.txt 2>&1
  python $R/tools/rocprof_summary.py $(find /tmp/pq -name "*.db" | head -1) > $O/prove_batch32_${prof}_kernel_stats.txt
done
cd $R; cat $O/prove_*_full.txt $O/prove_*_bucket.txt $O/prove_batch_*.txt | grep "ms"
