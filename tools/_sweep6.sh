source tools/red_sweep.sh gpurun_out/r3i
A=$PWD/manta_rs_amd/lib/libmantagpu_slimA.so
B=$PWD/manta_rs_amd/lib/libmantagpu_slimB.so
for rep in 1 2; do
run c16_scan_$rep MANTA_BENCH_C=16 MANTA_RED_S=0
run c16_scan_A_$rep MANTA_BENCH_C=16 MANTA_RED_S=0 MANTA_LIB=$A
run c16_scan_A_nocoop_$rep MANTA_BENCH_C=16 MANTA_RED_S=0 MANTA_LIB=$A MANTA_COOP_TILES=0 MANTA_COOP_WAVES=0
run c16_scan_B_$rep MANTA_BENCH_C=16 MANTA_RED_S=0 MANTA_LIB=$B
run c16_scan_A_d4_$rep MANTA_BENCH_C=16 MANTA_RED_S=0 MANTA_LIB=$A MANTA_BENCH_DEPTH=4
run c20_A_$rep MANTA_BENCH_C=20 MANTA_LIB=$A
run c20_A_noside_$rep MANTA_BENCH_C=20 MANTA_LIB=$A MANTA_RED_SIDE=0
run c20_B_noside_$rep MANTA_BENCH_C=20 MANTA_LIB=$B MANTA_RED_SIDE=0
run c20_A_noside_d4_$rep MANTA_BENCH_C=20 MANTA_LIB=$A MANTA_RED_SIDE=0 MANTA_BENCH_DEPTH=4
done
