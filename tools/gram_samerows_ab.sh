#!/bin/bash
# What is the HBM traffic of the moment kernel worth?  liboemgpu_samerows.so = the product with gram_sb.hip built -DOEM_SB_EXP_SAMEROWS:
# every workgroup of gram_sb_kernel multiplies the rows of its XCD's FIRST chunk -- the same MFMA work, (almost) no HBM traffic, wrong
# results (an experiment, never the product).  Build the variant here (CPU), run this on the GPU box:
#   bash tools/build_variant.sh samerows gram_sb.hip -DOEM_SB_EXP_SAMEROWS
# p = 256 runs on gram_wd.hip since round 5: OEM_NO_GRAM_WD=1 below keeps the comparison on gram_sb_kernel (profiles/r5_gram_one_read.txt).
export OEM_NO_GRAM_WD=1
for v in "" samerows "" samerows; do
  [ -n "$v" ] && export OEMGPU_LIB=oem_amd/liboemgpu_$v.so || unset OEMGPU_LIB
  echo "== ${v:-product}"
  python tools/gram_time.py 12500000 256 10 2>&1 | grep -v amdgpu.ids
  python tools/gram_time.py 1000000 512 10 2>&1 | grep -v amdgpu.ids
  python tools/gram_time.py 4000000 128 10 2>&1 | grep -v amdgpu.ids
done
