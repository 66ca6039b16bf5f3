python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q -x -k "shared_slab or moment_kernels or one_read" 2>&1 | tail -3
for v in "" 1 "" 1; do
  [ -n "$v" ] && export OEM_NO_GRAM_WD=1 || unset OEM_NO_GRAM_WD
  echo "== OEM_NO_GRAM_WD=${v:-unset}"
  for np in "500000 192" "500000 176" "500000 161" "4000000 192" "100000 192"; do set -- $np; python tools/gram_time.py $1 $2 10 2>&1 | grep -v amdgpu.ids; done
done
