#!/bin/bash
# A variant of libepa_dev.so for same-box A/Bs: one source recompiled with extra -D switches, the other objects of the
# regular build linked in.   bash profiles/build_variant.sh <tag> <source.hip> -DX=1 ...   -> exp/libepa_dev_<tag>.so
# (run with EPA_DEV_SO=exp/libepa_dev_<tag>.so)
tag=$1; src=$2; shift 2
R=$(cd $(dirname $0)/.. && pwd)
mkdir -p $R/exp
extra=""
case $src in thorough_dna.hip|thorough_aa_mfma.hip) extra="-mllvm -amdgpu-sched-strategy=iterative-ilp";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I $R/include -I $R/epa_ng_amd/csrc -Wno-unused-function -Wno-unused-result \
  $extra "$@" -c $R/epa_ng_amd/csrc/$src -o $R/exp/${src%.hip}_$tag.o || exit 1
objs=""
for o in $R/epa_ng_amd/build/*.o; do
  if [ "$(basename $o)" = "${src%.hip}.o" ]; then objs="$objs $R/exp/${src%.hip}_$tag.o"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/exp/libepa_dev_$tag.so $objs && echo built exp/libepa_dev_$tag.so
