for v in "" wsl6 wsl12 wsl18; do
  echo "== path_wcoop.hip sleep variant: ${v:-0}"
  [ -n "$v" ] && export OEMGPU_LIB=oem_amd/liboemgpu_$v.so || unset OEMGPU_LIB
  python tools/wres_time.py 3 2>&1 | grep "resident\]"
  python tools/wcoop_time.py 1 2>&1 | grep "coop"
  python tools/wstream_time.py 2 2>&1 | grep "streamed\]"
done
for v in "" ssl6 ssl12 ssl18; do
  echo "== path_symcoop.hip sleep variant: ${v:-0}"
  [ -n "$v" ] && export OEMGPU_LIB=oem_amd/liboemgpu_$v.so || unset OEMGPU_LIB
  python tools/c4_time.py 2>&1 | grep blocks | cut -c1-120
  python tools/symcoop_check.py 1088 2048 3000 2>&1 | grep -v amdgpu | cut -c90-260
done
