source tools/red_sweep.sh gpurun_out/r3aa
for rep in 1 2; do
run c16_$rep MANTA_BENCH_C=16
run c18_$rep MANTA_BENCH_C=18
run c20_$rep MANTA_BENCH_C=20
run c20_min4k_$rep MANTA_BENCH_C=20 MANTA_RED_MIN=4096
run c20_d4_$rep MANTA_BENCH_C=20 MANTA_BENCH_DEPTH=4
done
