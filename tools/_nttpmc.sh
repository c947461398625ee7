R=$PWD; mkdir -p $R/gpurun_out/r3ntt; cd /tmp; export TMPDIR=/tmp
for c in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE"; do
  n=$(echo $c | tr ' ' '_'); rm -rf /tmp/np_$n
  timeout 200 rocprofv3 --kernel-trace --pmc $c -d /tmp/np_$n -o m -- python $R/tools/_ntt.py > /dev/null 2>&1
  python $R/tools/pmc_kernels.py $(find /tmp/np_$n -name "*.db" | head -1) 2>/dev/null | grep -E "ntt_pass" >> $R/gpurun_out/r3ntt/pmc.txt
done
cd $R; cat gpurun_out/r3ntt/pmc.txt | cut -c1-40,70-200
