#!/bin/bash
# dev tool: the same profiled runs against several builds of the library (MANTA_LIB), one box, back to back
R=$PWD; cd /tmp; export TMPDIR=/tmp
for lib in "$@"; do
  echo "== $lib"
  for tool in prove_profile prove_batch_profile; do
    rm -rf /tmp/ab; MANTA_LIB=$R/$lib rocprofv3 --kernel-trace --stats -d /tmp/ab -o a -- python $R/tools/$tool.py > /tmp/ab.txt 2>/dev/null
    grep -E "ms/proof|ms per pass" /tmp/ab.txt
    python $R/tools/rocprof_summary.py $(find /tmp/ab -name "*.db" | head -1) | grep -E "ntt|digits|spmv|qap"
  done
  for i in 1 2; do MANTA_LIB=$R/$lib python $R/tools/prove_batch_profile.py 32 12 | tail -1; MANTA_LIB=$R/$lib python $R/tools/prove_profile.py | tail -1; done
done
