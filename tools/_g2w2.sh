#!/bin/bash
R=$PWD
for i in 1 2 3; do
for lib in manta_rs_amd/lib/libmantagpu.so manta_rs_amd/lib/libmantagpu_g2w2.so; do
  echo "== $lib"
  MANTA_LIB=$R/$lib python $R/tools/prove_profile.py | tail -1
  MANTA_LIB=$R/$lib python $R/tools/prove_batch_profile.py 32 12 | tail -1
done; done
