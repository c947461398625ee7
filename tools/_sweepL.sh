source tools/red_sweep.sh gpurun_out/r3r
for rep in 1 2; do
run L128_$rep MANTA_MSM_L=128
run L86_$rep MANTA_MSM_L=86
run L64_$rep MANTA_MSM_L=64
run L43_$rep MANTA_MSM_L=43
done
