# the moment kernels below the shared-slab form (p + 2 <= 112): HBM-bound at small p, MFMA-bound towards p = 110
for np in "8000000 4" "8000000 10" "8000000 14" "6000000 20" "6000000 30" "4000000 40" "4000000 46" "4000000 50" "3000000 62" "3000000 64" "2000000 80" "2000000 94" "2000000 100" "2000000 110"; do set -- $np; python tools/gram_time.py $1 $2 10 2>/dev/null | grep -v amdgpu.ids | python -c "
import sys,re
l=sys.stdin.read().strip(); m=re.search(r'n=(\d+) p=(\d+).*median ([0-9.]+)', l)
n,p,us=int(m.group(1)),int(m.group(2)),float(m.group(3))
fl=n*(p+2.0)*(p+3.0)
print(f'n={n} p={p}: moment kernel(s) {us/1e3:.3f} ms = {fl/us/1e6:.1f} TF = {fl/us/1e6/78.6:.3f} of the FP64-MFMA peak; {8.0*n*(p+1)/us/1e6:.2f} TB/s at one read of X and y')"; done
