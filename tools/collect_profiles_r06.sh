#!/bin/bash
# dev tool: round-6 measurement set on the GPU box -> gpurun_out/$1/ (run from the repo root)
R=$PWD; O=$R/gpurun_out/${1:-r06set}; mkdir -p $O
timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err
cp $R/gpurun_out/bench_detail_n1.json $O/bench_detail.json 2>/dev/null  # (the --quick runs below write the same file again)
cd /tmp && export TMPDIR=/tmp
# the bench command itself under the profiler (pipelined headline, 3 MSMs in flight) and with one MSM at a time
MANTA_BENCH_NO_PMC=1 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pb -o b -- python $R/bench.py --workload msm --quick --no-cpu-baseline > $O/bench_profiled.json 2>> $O/bench.err
python $R/tools/rocprof_summary.py $(find /tmp/pb -name "*.db" | head -1) > $O/bench_kernel_stats.txt
MANTA_BENCH_DEPTH=1 MANTA_BENCH_NO_PMC=1 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pb1 -o b -- python $R/bench.py --workload msm --quick --no-cpu-baseline > $O/bench_depth1_profiled.json 2>> $O/bench.err
python $R/tools/rocprof_summary.py $(find /tmp/pb1 -name "*.db" | head -1) > $O/bench_depth1_kernel_stats.txt
python $R/tools/msm_timeline.py $(find /tmp/pb1 -name "*.db" | head -1) > $O/msm_latency_timeline.txt
# proofs on the three witness profiles: single proofs (kernel stats + timeline of the last proof) and 32-proof passes
for prof in W sparse dense; do
  rm -rf /tmp/pp; PROFILE=$prof PROVE_N=30 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pp -o p -- python $R/tools/prove_profile.py > $O/prove_${prof}.txt 2>> $O/bench.err
  python $R/tools/rocprof_summary.py $(find /tmp/pp -name "*.db" | head -1) > $O/prove_${prof}_kernel_stats.txt
  python $R/tools/proof_timeline.py $(find /tmp/pp -name "*.db" | head -1) > $O/proof_timeline_${prof}.txt
  rm -rf /tmp/pq; PROFILE=$prof timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pq -o q -- python $R/tools/prove_batch_profile.py > $O/prove_batch_${prof}.txt 2>> $O/bench.err
  python $R/tools/rocprof_summary.py $(find /tmp/pq -name "*.db" | head -1) > $O/prove_batch32_${prof}_kernel_stats.txt
done
# PMC passes of the accumulate kernel: VALU instructions and waves (FETCH / WRITE are measured by bench.py itself)
: > $O/pmc_kernels.txt
for c in SQ_INSTS_VALU SQ_WAVES; do
  rm -rf /tmp/pmc_$c; MANTA_BENCH_NO_PMC=1 MANTA_BENCH_DEPTH=1 MANTA_MSM_DEDICATED_QUEUES=0 timeout 600 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_$c -o m -- python $R/bench.py --workload msm --quick --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2>> $O/bench.err
  python $R/tools/pmc_kernels.py $(find /tmp/pmc_$c -name "*.db" | head -1) 2>/dev/null | grep -E "accumulate_chunks|digits_kernel" >> $O/pmc_kernels.txt
done
cd $R
CHECK=0 timeout 400 python tools/profile_proofs.py sparse,W,dense > $O/profile_proofs.txt 2>&1
tail -c 600 $O/bench.err
