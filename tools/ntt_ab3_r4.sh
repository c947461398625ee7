#!/bin/bash
# dev tool (round 4): 2^20 / 2^16 transform times per library build (load/store-phase product coding, waves_per_eu)
R=$PWD
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
for lib in libmantagpu_base libmantagpu libmantagpu_w4; do
  export MANTA_LIB=$R/manta_rs_amd/lib/$lib.so
  echo "$lib: 2^20 $(python $R/tools/ntt_loop.py 20 40 2>&1 | grep '2^20' | tr '\n' ' ')"
  echo "$lib: 2^16 $(python $R/tools/ntt_loop.py 16 40 2>&1 | grep '2^16' | tr '\n' ' ')"
done; done
