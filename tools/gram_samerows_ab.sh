for v in "" samerows "" samerows; do
  [ -n "$v" ] && export OEMGPU_LIB=oem_amd/liboemgpu_$v.so || unset OEMGPU_LIB
  echo "== ${v:-product}"
  python tools/gram_time.py 12500000 256 10 2>&1 | grep -v amdgpu.ids
  python tools/gram_time.py 1000000 512 10 2>&1 | grep -v amdgpu.ids
  python tools/gram_time.py 4000000 128 10 2>&1 | grep -v amdgpu.ids
done
