# timeline of one latency-mode MSM for a given window width / reduce knobs: tools/_tl.sh <name> VAR=val...
R=$PWD; name=$1; shift
mkdir -p $R/gpurun_out/r3c
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl_$name
env "$@" MANTA_BENCH_DEPTH=1 rocprofv3 --kernel-trace -d /tmp/tl_$name -o b -- python $R/bench.py --quick --no-cpu-baseline --steps 4 --warmup 1 > /dev/null 2> $R/gpurun_out/r3c/$name.err
python $R/tools/msm_timeline.py $(find /tmp/tl_$name -name "*.db" | head -1) > $R/gpurun_out/r3c/tl_$name.txt
cd $R
