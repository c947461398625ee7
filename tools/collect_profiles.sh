#!/bin/bash
# dev tool: collect the round's measurement set on the GPU box into gpurun_out/$1/ (run from the repo root)
set -x
R=$PWD; O=$R/gpurun_out/${1:-set}; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
for sh in to_private to_public; do python bench.py --workload prove --shape $sh --no-cpu-baseline >> $O/prove_shapes.jsonl 2>> $O/bench.err; done
cd /tmp && export TMPDIR=/tmp
# pipelined headline (3 MSMs in flight) and latency mode (one at a time): kernel statistics + the timeline of one MSM
rocprofv3 --kernel-trace --stats -d /tmp/pb -o b -- python $R/bench.py --quick --no-cpu-baseline > $O/bench_profiled.json 2>> $O/bench.err
python $R/tools/rocprof_summary.py $(find /tmp/pb -name "*.db" | head -1) > $O/bench_kernel_stats.txt
MANTA_BENCH_DEPTH=1 rocprofv3 --kernel-trace --stats -d /tmp/pb1 -o b -- python $R/bench.py --quick --no-cpu-baseline > $O/bench_depth1_profiled.json 2>> $O/bench.err
python $R/tools/rocprof_summary.py $(find /tmp/pb1 -name "*.db" | head -1) > $O/bench_depth1_kernel_stats.txt
python $R/tools/msm_timeline.py $(find /tmp/pb1 -name "*.db" | head -1) > $O/msm_latency_timeline.txt
rocprofv3 --kernel-trace --stats -d /tmp/pp -o p -- python $R/tools/prove_profile.py > $O/prove_profiled.txt 2>> $O/bench.err
python $R/tools/rocprof_summary.py $(find /tmp/pp -name "*.db" | head -1) > $O/prove_kernel_stats.txt
python $R/tools/proof_timeline.py $(find /tmp/pp -name "*.db" | head -1) > $O/proof_timeline.txt
rocprofv3 --kernel-trace --stats -d /tmp/pq -o q -- python $R/tools/prove_batch_profile.py > $O/prove_batch_profiled.txt 2>> $O/bench.err
python $R/tools/rocprof_summary.py $(find /tmp/pq -name "*.db" | head -1) > $O/prove_batch32_kernel_stats.txt
# PMC passes (own runs, --kernel-trace only): HBM traffic and VALU mix of the accumulate kernel
for c in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU SQ_WAVES; do
  rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_$c -o m -- python $R/bench.py --quick --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2>> $O/bench.err
  python $R/tools/pmc_kernels.py $(find /tmp/pmc_$c -name "*.db" | head -1) 2>/dev/null | grep -E "accumulate_chunks|digits_kernel|gather_only" >> $O/pmc_kernels.txt
done
MANTA_ACC_GATHER_ONLY=1 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmc_g -o m -- python $R/tools/gather_calibration.py 16 > /dev/null 2>> $O/bench.err
for db in $(find /tmp/pmc_g -name "*.db"); do python $R/tools/pmc_kernels.py $db 2>/dev/null | grep -E "gather_only|accumulate_chunks" >> $O/pmc_gather_only.txt; done
python $R/tools/gather_calibration.py 16 > $O/gather_calibration.jsonl 2>> $O/bench.err
python $R/tools/hbm_bw.py > $O/hbm_bw.txt 2>> $O/bench.err
# single verification + a batch (wave-cooperative pairing), and its building blocks
rocprofv3 --kernel-trace --stats -d /tmp/pv -o v -- python $R/tools/verify_profile.py > $O/verify_profiled.txt 2>> $O/bench.err
python $R/tools/rocprof_summary.py $(find /tmp/pv -name "*.db" | head -1) | grep -E "kernel  |miller|final_exp|prepare|f12" > $O/verify_kernel_stats.txt
[ -x $R/tools/ubench_pairing ] && $R/tools/ubench_pairing > $O/ubench_pairing.txt
python $R/tools/batch_threads_sweep.py > $O/batch_threads_sweep.txt 2>> $O/bench.err
python $R/tools/config3_bls_2_20.py 1 20 > $O/config3_bls12_381_2_20.jsonl 2>> $O/bench.err
