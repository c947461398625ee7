#!/bin/bash
# dev tool (round 4): PMC counters of EVERY kernel of (1) the MSM headline command with one MSM in flight and (2) 32-proof
# PrivateTransfer passes on the W profile -- the tail kernels (sort, merges, bucket reduce, witness map) next to the accumulate
# kernel. One counter group per run (separate --pmc passes, --kernel-trace only). -> gpurun_out/$1/pmc_tails_{msm,batch}.txt
R=$PWD; O=$R/gpurun_out/${1:-r5t}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
GROUPS_=("SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_WAVES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "FETCH_SIZE" "WRITE_SIZE")
: > $O/pmc_tails_msm.txt; : > $O/pmc_tails_batch.txt
for grp in "${GROUPS_[@]}"; do
  rm -rf /tmp/pm; MANTA_BENCH_DEPTH=1 MANTA_BENCH_NO_PMC=1 timeout 300 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pm -o m -- python $R/bench.py --workload msm --quick --no-cpu-baseline --steps 3 --warmup 1 > /tmp/pm.log 2>&1
  db=$(find /tmp/pm -name "*.db" | head -1)
  if [ -n "$db" ]; then python $R/tools/pmc_kernels.py $db 2>/dev/null | grep -v -E "precompute|fixed_base|xyzz_to_affine|clock_probe|bases_to_internal|rocclr" >> $O/pmc_tails_msm.txt; else echo "no db for $grp: $(tail -2 /tmp/pm.log)" >> $O/pmc_tails_msm.txt; fi
  rm -rf /tmp/pq; PROFILE=W timeout 300 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pq -o q -- python $R/tools/prove_batch_profile.py 32 3 > /tmp/pq.log 2>&1
  db=$(find /tmp/pq -name "*.db" | head -1)
  if [ -n "$db" ]; then python $R/tools/pmc_kernels.py $db 2>/dev/null | grep -v -E "precompute|fixed_base|xyzz_to_affine|full_table_chain|bases_to_internal|rocclr|std_to_rr|powers_kernel|permute_bitrev" >> $O/pmc_tails_batch.txt; else echo "no db for $grp: $(tail -2 /tmp/pq.log)" >> $O/pmc_tails_batch.txt; fi
done
# durations of the same launches (kernel trace only)
rm -rf /tmp/pk; MANTA_BENCH_DEPTH=1 MANTA_BENCH_NO_PMC=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pk -o k -- python $R/bench.py --workload msm --quick --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2>&1
python $R/tools/rocprof_summary.py $(find /tmp/pk -name "*.db" | head -1) > $O/pmc_tails_msm_durations.txt
rm -rf /tmp/pk; PROFILE=W timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pk -o k -- python $R/tools/prove_batch_profile.py 32 3 > /dev/null 2>&1
python $R/tools/rocprof_summary.py $(find /tmp/pk -name "*.db" | head -1) > $O/pmc_tails_batch_durations.txt
wc -l $O/pmc_tails_*.txt
