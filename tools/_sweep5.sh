source tools/red_sweep.sh gpurun_out/r3h
for rep in 1 2; do
run c16_scan_$rep MANTA_BENCH_C=16 MANTA_RED_S=0
run c16_scan_d4_$rep MANTA_BENCH_C=16 MANTA_RED_S=0 MANTA_BENCH_DEPTH=4
run c16_side_$rep MANTA_BENCH_C=16
run c20_side_$rep MANTA_BENCH_C=20
run c20_side_d4_$rep MANTA_BENCH_C=20 MANTA_BENCH_DEPTH=4
run c20_noside_$rep MANTA_BENCH_C=20 MANTA_RED_SIDE=0
done
