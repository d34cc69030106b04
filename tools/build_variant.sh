#!/bin/bash
# usage: tools/build_variant.sh NAME "EXTRA_FLAGS" [files to recompile with the flags ...]   (default: rf_k_col_gen)
# builds tools/bin/lib_NAME.so = the product library with some translation units recompiled under extra -D flags
# (the configuration selectors of rf_configs.h); the other objects are reused from randomfield_amd/csrc/*.o
set -e
name=$1; extra=$2; shift 2 || true
files=${@:-rf_k_col_gen}
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/randomfield_amd/csrc
out=/tmp/rf_variant_$name
mkdir -p $out $root/tools/bin
objs=""
for f in rf_k_col_plain rf_k_col_direct rf_k_col_gen rf_k_col_gen64 rf_k_row rf_k_row_c2c rf_k_yz rf_k_misc rf_k_mt rf_k_generic rf_capi rf_capi_mt rf_capi_slab; do
  fl=""; [[ $f == rf_k_col_gen || $f == rf_k_mt ]] && fl="-mllvm -amdgpu-sched-strategy=max-ilp"       # (the Makefile's per-file flag)
  if [[ " $files " == *" $f "* ]]; then
    hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -Wno-unused-function -Wno-unused-variable $fl $extra -I$src -c $src/$f.hip -o $out/$f.o &
    objs="$objs $out/$f.o"
  else
    objs="$objs $src/$f.o"
  fi
done
wait
hipcc -shared -fPIC --offload-arch=gfx950 -o $root/tools/bin/lib_$name.so $objs -ldl
echo built tools/bin/lib_$name.so
