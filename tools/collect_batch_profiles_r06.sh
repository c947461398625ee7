R=$PWD; O=$R/gpurun_out/r06batch; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for prof in W sparse dense; do
  rm -rf /tmp/pq; PROFILE=$prof timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pq -o q -- python $R/tools/prove_batch_profile.py > $O/prove_batch_${prof}.txt 2>> $O/err.txt
  python $R/tools/rocprof_summary.py $(find /tmp/pq -name "*.db" | head -1) > $O/prove_batch32_${prof}_kernel_stats.txt
  tail -1 $O/prove_batch_${prof}.txt
done
