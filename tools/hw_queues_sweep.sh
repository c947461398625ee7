# dev tool: batched / sequential proving against the HIP runtime's hardware-queue count
for q in ${@:-4 6}; do echo "GPU_MAX_HW_QUEUES=$q"; for K in 256 1024; do GPU_MAX_HW_QUEUES=$q python tools/batch_threads_sweep.py $K 2>&1 | grep "K="; done; GPU_MAX_HW_QUEUES=$q python tools/prove_profile.py 2>&1 | tail -1; done
