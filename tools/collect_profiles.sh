#!/bin/bash
# dev tool: collect the round's measurement set on the GPU box into gpurun_out/set/ (run from the repo root)
set -x
R=$PWD; O=$R/gpurun_out/set; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
P="python bench.py --workload prove"
$P --threads 1 --steps 100 > $O/prove.jsonl 2>> $O/bench.err
$P --threads 2 --steps 300 --no-cpu-baseline >> $O/prove.jsonl 2>> $O/bench.err
$P --threads 2 --batch 8 --steps 640 --no-cpu-baseline >> $O/prove.jsonl 2>> $O/bench.err
$P --threads 2 --batch 32 --steps 1280 --no-cpu-baseline >> $O/prove.jsonl 2>> $O/bench.err
for sh in to_private to_public; do
  $P --shape $sh --threads 2 --steps 300 --no-cpu-baseline >> $O/prove.jsonl 2>> $O/bench.err
  $P --shape $sh --threads 2 --batch 32 --steps 1280 --no-cpu-baseline >> $O/prove.jsonl 2>> $O/bench.err
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/pb -o b -- python $R/bench.py --no-cpu-baseline > $O/bench_profiled.json 2>> $O/bench.err
python $R/tools/rocprof_summary.py $(find /tmp/pb -name "*.db" | head -1) > $O/bench_kernel_stats.txt
rocprofv3 --kernel-trace --stats -d /tmp/pp -o p -- python $R/tools/prove_profile.py > $O/prove_profiled.txt 2>> $O/bench.err
python $R/tools/rocprof_summary.py $(find /tmp/pp -name "*.db" | head -1) > $O/prove_kernel_stats.txt
