for v in "" wss1 wss3 wss6; do
  echo "== path_wcoop.hip sleep between sweeps: ${v:-0}"
  [ -n "$v" ] && export OEMGPU_LIB=oem_amd/liboemgpu_$v.so || unset OEMGPU_LIB
  python tools/wres_time.py 3 2>&1 | grep "resident\]"
  python tools/wcoop_time.py 1 2>&1 | grep "\[coop\]"
done
