#!/bin/bash
# Experiment helper: liboemgpu_<name>.so = the product objects with ONE source recompiled with extra flags.
#   tools/build_variant.sh nt gram.hip '-DOEM_GLDS_POLICY=" nt"'
# Load it with OEMGPU_LIB=oem_amd/liboemgpu_<name>.so.  Never the product.
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; shift 2
python -m oem_amd.build > /dev/null
B=oem_amd/_build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -fvisibility=hidden -Wno-unused-function "$@" -c oem_amd/csrc/$src -o $B/$src.$name.o
objs=""
for o in $B/*.hip.o; do
  if [ "$(basename $o)" = "$src.o" ]; then objs="$objs $B/$src.$name.o"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o oem_amd/liboemgpu_$name.so $objs
echo oem_amd/liboemgpu_$name.so
