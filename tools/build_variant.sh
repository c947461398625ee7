#!/bin/bash
# dev tool: an A/B build of the library -- tools/build_variant.sh <tag> "<extra flags>" unit [unit ...]
# recompiles the named units (e.g. msm_bls381_g1) with the extra flags and links manta_rs_amd/lib/libmantagpu_<tag>.so
# from them plus the regular objects of everything else; select it at run time with MANTA_LIB=<path>
set -e
tag=$1; flags=$2; shift 2
cd $(dirname $0)/../manta_rs_amd/csrc
make -s -j8
mkdir -p ../lib/obj_$tag
objs=""
for o in ../lib/obj/*.o; do
  [ "$(basename $o)" = "prover_diag.o" ] && continue
  b=$(basename $o .o); use=$o
  for u in "$@"; do
    if [ "$u" = "$b" ]; then
      src=$u.hip; [ -f $src ] || src=$u.cpp
      /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -Wno-unused-value $flags -x hip -c $src -o ../lib/obj_$tag/$b.o 2>/dev/null
      use=../lib/obj_$tag/$b.o
    fi
  done
  objs="$objs $use"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libmantagpu_$tag.so $objs -ldl -Wl,-rpath,/opt/rocm/lib
echo built manta_rs_amd/lib/libmantagpu_$tag.so
