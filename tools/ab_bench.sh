#!/bin/bash
# A/B of library variants on ONE box: tools/ab_bench.sh <reps> <lib or "-"> ...   ("-" = the product library)
reps=$1; shift
run() { python bench.py --no-c5 --no-host --no-live-pmc --steps 30 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['value'],1), 'gram_ms', round(d['roofline']['kernel_ms'],5), 'frac', round(d['roofline']['frac'],4), 'path_cycles', round(d['path_kernel_cycles']), 'path_ms', round(d['stage_ms']['eigen_plus_path'],4), 'clk', round(d['path_kernel_clock_GHz'],3))"; }
for i in $(seq $reps); do
  for L in "$@"; do
    if [ "$L" = "-" ]; then run product; else OEMGPU_LIB=$L run $L; fi
  done
done
