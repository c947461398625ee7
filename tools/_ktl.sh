R=$PWD; mkdir -p $R/gpurun_out/r3n; cd /tmp; export TMPDIR=/tmp
for k in 1 2 4; do
  rm -rf /tmp/kt$k
  timeout 300 rocprofv3 --kernel-trace -d /tmp/kt$k -o p -- python $R/tools/prove_batch_profile.py $k 6 > $R/gpurun_out/r3n/k$k.txt 2>/dev/null
  python $R/tools/proof_timeline.py $(find /tmp/kt$k -name "*.db" | head -1) > $R/gpurun_out/r3n/tl_k$k.txt
done
cd $R; tail -1 gpurun_out/r3n/k1.txt gpurun_out/r3n/k2.txt gpurun_out/r3n/k4.txt
