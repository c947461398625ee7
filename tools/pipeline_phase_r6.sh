#!/bin/bash
# dev tool (round 6): rocprofv3 kernel timeline of the pipelined headline (three 2^20 MSMs in flight) in a FAST and a SLOW process
# (K foreign streams created first: tools/precreate_bench.py); usage: tools/pipeline_phase_r6.sh "0 4" [ENV=VAL ...]
R=$PWD; O=$R/gpurun_out/r06_phase; mkdir -p $O
ks=$1; shift
cd /tmp && export TMPDIR=/tmp
for k in $ks; do
  rm -rf /tmp/ph_$k
  env "$@" timeout 600 rocprofv3 --kernel-trace -d /tmp/ph_$k -o p -- python $R/tools/precreate_bench.py $k --quick --no-cpu-baseline --workload msm > $O/bench_K$k.txt 2>&1
  tail -1 $O/bench_K$k.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('K=$k %.1f Mscalar/s (%.3f ms/step) one at a time %s' % (d['value'], d['ms_per_step'], d['config'].get('one_at_a_time_Mscalar_s')))"
  db=$(find /tmp/ph_$k -name "*.db" | head -1)
  python $R/tools/pipe_timeline.py $db 12 > $O/pipe_K$k.txt
done
