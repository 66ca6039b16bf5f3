#!/bin/bash
# A/B of library variants on config 4, alternately on ONE box: tools/ab_c4.sh <reps> <lib or "-"> ...   ("-" = the product library)
reps=$1; shift
for i in $(seq $reps); do
  for L in "$@"; do
    if [ "$L" = "-" ]; then echo -n "product: "; python tools/c4_time.py "" 2>/dev/null | grep -v amdgpu.ids | cut -c1-90
    else echo -n "$L: "; OEMGPU_LIB=$L python tools/c4_time.py "" 2>/dev/null | grep -v amdgpu.ids | cut -c1-90; fi
  done
done
