R=$PWD; name=$1; shift
mkdir -p $R/gpurun_out/r3g
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pp_$name
env "$@" rocprofv3 --kernel-trace -d /tmp/pp_$name -o b -- python $R/bench.py --quick --no-cpu-baseline --steps 12 --warmup 3 > $R/gpurun_out/r3g/$name.json 2> $R/gpurun_out/r3g/$name.err
python $R/tools/pipe_timeline.py $(find /tmp/pp_$name -name "*.db" | head -1) 9 > $R/gpurun_out/r3g/pipe_$name.txt
python $R/tools/gpu_busy.py $(find /tmp/pp_$name -name "*.db" | head -1) 0.3 > $R/gpurun_out/r3g/busy_$name.txt
cd $R
