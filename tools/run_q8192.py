"""oem.xtx at q = 8,192 (the HBM-bound GEMV loop: the packed lower triangle streamed once per product), for rocprofv3:
python3 tools/run_q8192.py [q] [penalty | full]   (full: dense-vector products only, for the counter passes)"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, oem_amd
p = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
pen = sys.argv[2] if len(sys.argv) > 2 else "lasso"
g = torch.Generator(device="cuda"); g.manual_seed(p)
n = p + p // 2
x = torch.randn((n, p), generator=g, device="cuda", dtype=torch.float64)
b = torch.zeros(p, dtype=torch.float64, device="cuda"); b[:25] = 1.0
y = x @ b + torch.randn(n, generator=g, device="cuda", dtype=torch.float64)
xtx = (x.t() @ x) / n; xty = ((x.t() @ y) / n).cpu().numpy()
del x
if pen == "full":
    # FULL products only (a dense vector: no block is skipped) -- what the counter passes behind bench.py's q8192_ms roofline profile
    import ctypes as C
    from oem_amd import _lib as L
    v = torch.randn(p, generator=g, device="cuda", dtype=torch.float64); o = torch.empty_like(v)
    us = C.c_double(0.0)
    L.check(L.lib().oemgpu_selftest_sympk_gemv(oem_amd.context(), xtx.data_ptr(), p, v.data_ptr(), o.data_ptr(), 60, C.byref(us)))
    print("full products:", us.value, "us each")
    sys.exit(0)
grp = np.arange(p) // 8 + 1 if pen.startswith("grp") else ()
for _ in range(2):
    fit = oem_amd.oem_xtx(xtx, xty, penalty=pen, groups=grp, nlambda=20, tol=1e-8, lambda_min_ratio=0.01)
print(int(fit["niter"][0].sum()), oem_amd.last_path_engine()[0])
