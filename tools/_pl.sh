run() { echo "== $@"; env "$@" MANTA_COALESCE=0 timeout 200 python tools/pass_latency.py 1 2 4 8 2>&1 | grep "k="; }
run X=1
run MANTA_PROVE_STREAMS=3
run MANTA_PROVE_STREAMS=1
run MANTA_GRAPH=off
run MANTA_GRAPH=split
run GPU_MAX_HW_QUEUES=8
run MANTA_WIDE_MIN=16
