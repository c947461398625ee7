source tools/red_sweep.sh gpurun_out/r3f
for rep in 1 2; do
run c16_scan_$rep MANTA_BENCH_C=16 MANTA_RED_S=0
run c16_side_$rep MANTA_BENCH_C=16
run c16_noside_$rep MANTA_BENCH_C=16 MANTA_RED_SIDE=0
run c16_noside_sp4_$rep MANTA_BENCH_C=16 MANTA_RED_SIDE=0 MANTA_RED_SP=4
run c16_noside_min4k_$rep MANTA_BENCH_C=16 MANTA_RED_SIDE=0 MANTA_RED_MIN=4096
run c16_scan_q8_$rep MANTA_BENCH_C=16 MANTA_RED_S=0 GPU_MAX_HW_QUEUES=8
run c16_side_q8_$rep MANTA_BENCH_C=16 GPU_MAX_HW_QUEUES=8
run c20_side_q8_$rep MANTA_BENCH_C=20 GPU_MAX_HW_QUEUES=8
done
