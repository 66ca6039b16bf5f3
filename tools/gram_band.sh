for np in "1000000 100" "2000000 128" "500000 160" "500000 192" "2000000 256" "400000 300" "500000 384" "1000000 512" "200000 700" "200000 1024" "100000 2048" "60000 3000" "40000 5000" "30000 8192" "20000 12000"; do set -- $np; python tools/gram_time.py $1 $2 5 2>/dev/null | grep -v amdgpu.ids | python -c "
import sys,re
l=sys.stdin.read().strip(); m=re.search(r'n=(\d+) p=(\d+).*median ([0-9.]+)', l)
n,p,us=int(m.group(1)),int(m.group(2)),float(m.group(3))
fl=n*p*(p+1.0)+2.0*n*p
print(f'n={n} p={p}: moment kernel(s) {us/1e3:.2f} ms = {fl/us/1e6:.1f} TF = {fl/us/1e6/78.6:.3f} of the FP64-MFMA peak; {8.0*n*p/us/1e6:.2f} TB/s at one read of X')"; done
