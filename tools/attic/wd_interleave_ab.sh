#!/bin/bash
# A/B on one box: gram_wd_kernel's workgroups of a row chunk one behind the other on one XCD (product) against "all diagonal units, then all
# off-diagonal blocks" (oem_amd/liboemgpu_wdseq.so = tools/build_variant.sh wdseq gram_wd.hip -DOEM_WD_INTERLEAVE=0): time and HBM traffic at p = 512 / 1024
cd "$(dirname "$0")/.."; export TMPDIR=/tmp; o=gpurun_out
for i in 1 2; do
  for L in "" oem_amd/liboemgpu_wdseq.so; do
    export OEMGPU_LIB=$L; [ -z "$L" ] && unset OEMGPU_LIB
    echo -n "${L:-product}: "; python3 tools/gram_c3_time.py 1000000 512 2>/dev/null | grep -v amdgpu.ids
  done
done
for L in "" oem_amd/liboemgpu_wdseq.so; do
  export OEMGPU_LIB=$L; [ -z "$L" ] && unset OEMGPU_LIB
  echo -n "${L:-product}: "; python3 tools/gram_c3_time.py 400000 1024 2>/dev/null | grep -v amdgpu.ids
  tag=$( [ -z "$L" ] && echo il || echo seq )
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $o/wd_${tag}_$c -o x -- python3 tools/run_c3.py c3 2 > /dev/null 2> $o/wd_${tag}_$c.err
  done
  python3 tools/pmc_summary.py gram_wd_kernel $o/wd_${tag}_pmc.json "$(find $o/wd_${tag}_FETCH_SIZE -name '*counter_collection.csv' | head -1)" "$(find $o/wd_${tag}_WRITE_SIZE -name '*counter_collection.csv' | head -1)" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('  HBM bytes per launch', d['hbm_bytes_per_dispatch'] / 1e9, 'GB (algorithmic 4.10)')"
  rm -rf $o/wd_${tag}_FETCH_SIZE $o/wd_${tag}_WRITE_SIZE
done
