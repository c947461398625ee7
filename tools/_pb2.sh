run() { echo "== $@"; for i in 1 2; do env "$@" timeout 120 python tools/prove_batch_profile.py 32 12 2>&1 | grep "k="; done; env "$@" timeout 200 python tools/batch_threads_sweep.py 1024 2>/dev/null | grep -E "K=|proofs/s" | tail -3; }
run MANTA_GRAPH=off
run MANTA_GRAPH=off MANTA_RED_S=3 MANTA_RED_MIN=4096
run MANTA_GRAPH=off MANTA_RED_S=2 MANTA_RED_MIN=4096
run MANTA_GRAPH=split
run MANTA_GRAPH=split MANTA_RED_S=3 MANTA_RED_MIN=4096
run X=1
