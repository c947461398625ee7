for rep in 1 2 3; do for e in "X=1" "MANTA_RED_S=0" "MANTA_RED_S=3"; do echo -n "$e: "; env $e MANTA_BENCH_NO_PMC=1 timeout 300 python bench.py --workload msm --quick --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print(d['value'], d['ms_per_step'], d['config']['latency_mode']['ms_per_msm'], d['roofline']['kernel_ms'])"; done; done
