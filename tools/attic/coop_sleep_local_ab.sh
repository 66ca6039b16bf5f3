for v in "" csl0 csl2 csl8 csl12; do
  echo "== OEM_XCHG_SLEEP_LOCAL variant: ${v:-5 (product)}"
  [ -n "$v" ] && export OEMGPU_LIB=oem_amd/liboemgpu_$v.so || unset OEMGPU_LIB
  for r in 1 2; do python tools/coop_time.py 2>&1 | grep ": coop  " | cut -c1-100; done
done
