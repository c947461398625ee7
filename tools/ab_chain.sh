# dev tool: A/B of two library builds (manta_rs_amd/lib/libmantagpu_{base,chain}.so): 2^20 MSM and batched PrivateTransfer proofs
R=$PWD
for rep in 1 2 3; do for v in base chain; do echo -n "$v: "; MANTA_LIB=$R/manta_rs_amd/lib/libmantagpu_$v.so python bench.py --workload msm --quick --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print(d['value'], d['config']['latency_mode']['ms_per_msm'], d['roofline']['kernel_ms'])"; done; done
cd /tmp; export TMPDIR=/tmp
for rep in 1 2; do for v in base chain; do echo "== $v"; rm -rf /tmp/ab; MANTA_LIB=$R/manta_rs_amd/lib/libmantagpu_$v.so rocprofv3 --kernel-trace --stats -d /tmp/ab -o a -- python $R/tools/prove_batch_profile.py 32 8 2>/dev/null | grep "per pass"; python $R/tools/rocprof_summary.py $(find /tmp/ab -name "*.db" | head -1) | grep accumulate_chunks | cut -c1-130; MANTA_LIB=$R/manta_rs_amd/lib/libmantagpu_$v.so python $R/tools/batch_threads_sweep.py 1024 2>/dev/null | grep "K="; done; done
