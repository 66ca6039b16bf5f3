# A/B of the delayed first poll sweep in path_coop.hip (build: tools/build_variant.sh csl6 path_coop.hip -DOEM_XCHG_SLEEP=6, ... csl12)
for v in "" csl6 csl12; do
  echo "== path_coop.hip sleep variant: ${v:-0}"
  [ -n "$v" ] && export OEMGPU_LIB=oem_amd/liboemgpu_$v.so || unset OEMGPU_LIB
  python tools/coop_time.py 2>&1 | grep -v amdgpu.ids | grep "coop" | grep -v stride | cut -c1-200
done
