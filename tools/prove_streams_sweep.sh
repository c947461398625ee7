for rep in 1 2; do for st in 6 3 1; do echo "MANTA_PROVE_STREAMS=$st"; for K in 256 1024; do MANTA_PROVE_STREAMS=$st python tools/batch_threads_sweep.py $K 2>&1 | grep "K="; done; done; done
for fl in 4 5 6; do echo "streams=1 inflight=$fl"; MANTA_BATCH_INFLIGHT=$fl MANTA_PROVE_STREAMS=1 python tools/batch_threads_sweep.py 1024 2>&1 | grep "K="; done
