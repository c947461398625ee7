#!/bin/bash
# dev tool (round 4): PMC counters of the NTT pass kernel (and of the tail kernels of an MSM) -- one counter group per run
R=$PWD; O=$R/gpurun_out/${1:-r4h}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
: > $O/pmc_ntt.txt
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_WAVES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  rm -rf /tmp/pm; timeout 300 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pm -o m -- python $R/tools/ntt_loop.py 20 5 > /tmp/pm.log 2>&1
  db=$(find /tmp/pm -name "*.db" | head -1)
  if [ -n "$db" ]; then python $R/tools/pmc_kernels.py $db 2>/dev/null | grep -E "ntt_pass" >> $O/pmc_ntt.txt; else echo "no db for $grp: $(tail -2 /tmp/pm.log)" >> $O/pmc_ntt.txt; fi
done
rm -rf /tmp/pk; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pk -o k -- python $R/tools/ntt_loop.py 20 5 > $O/ntt_loop.txt 2>&1
python $R/tools/ntt_pass_times.py $(find /tmp/pk -name "*.db" | head -1) >> $O/pmc_ntt.txt
cat $O/pmc_ntt.txt
