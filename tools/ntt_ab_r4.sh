#!/bin/bash
# dev tool (round 4): NTT pass kernels in a 32-proof PrivateTransfer pass (serialised launches) per MANTA_NTT_R, and the batched rate
R=$PWD; O=$R/gpurun_out/${1:-r4i}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for r in 0 2 3; do
  rm -rf /tmp/ps; MANTA_NTT_R=$r PROFILE=W MANTA_PROVE_STREAMS=1 MANTA_GRAPH=off timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ps -o q -- python $R/tools/prove_batch_profile.py 32 6 > /tmp/ps.txt 2>&1
  echo "== R=$r $(grep k=32 /tmp/ps.txt)"
  python $R/tools/rocprof_summary.py $(find /tmp/ps -name "*.db" | head -1) | grep -E "ntt_pass|spmv|qap"
  python $R/tools/ntt_pass_times.py $(find /tmp/ps -name "*.db" | head -1)
done
cd $R
for r in 0 2 3 0 2 3; do echo "R=$r $(MANTA_NTT_R=$r PROFILE=W python tools/prove_batch_profile.py 32 20 | tail -1)"; done
