source tools/red_sweep.sh gpurun_out/r3ad
for rep in 1 2; do
run c16_$rep MANTA_BENCH_C=16
run c16_s4_$rep MANTA_BENCH_C=16 MANTA_RED_S=4
run c17_s4_$rep MANTA_BENCH_C=17 MANTA_RED_S=4
run c17_s5_$rep MANTA_BENCH_C=17 MANTA_RED_S=5
run c17_s4_sp4_$rep MANTA_BENCH_C=17 MANTA_RED_S=4 MANTA_RED_SP=4
run c17_s4_d4_$rep MANTA_BENCH_C=17 MANTA_RED_S=4 MANTA_BENCH_DEPTH=4
done
