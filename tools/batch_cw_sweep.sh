# dev tool: batched proving throughput against the window width of the a / b / l tables (MANTA_PROVE_CW) and of h (MANTA_PROVE_CH)
for rep in 1 2; do
for cw in 10 11 12 9; do echo "cw=$cw"; MANTA_PROVE_CW=$cw python tools/batch_threads_sweep.py 256 2>&1 | grep "K= 256 host threads=2"; done
for ch in 12 13 15; do echo "ch=$ch"; MANTA_PROVE_CH=$ch python tools/batch_threads_sweep.py 256 2>&1 | grep "K= 256 host threads=2"; done
done
