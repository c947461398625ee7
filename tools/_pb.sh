run() { echo "== $@"; env "$@" MANTA_BENCH_DISTINCT=0 timeout 300 python bench.py --workload prove --child --batched-only --no-cpu-baseline --gpus 1 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print(d['batched'])"; }
for rep in 1 2; do
run X=1
run MANTA_RED_S=3 MANTA_RED_MIN=4096
run MANTA_RED_S=2 MANTA_RED_MIN=4096
run MANTA_RED_S=3 MANTA_RED_MIN=1024
done
