import sys, os, ctypes as C, numpy as np
sys.path.insert(0, "/root/repo")
import torch, oem_amd
from oem_amd import _lib as L
dev = "cuda"
q8, n8 = 8192, 16 * 8192
g8 = torch.Generator(device=dev); g8.manual_seed(8192)
xtx8 = torch.zeros((q8, q8), device=dev, dtype=torch.float64)
b8 = torch.zeros(q8, dtype=torch.float64, device=dev); b8[:25] = 2.0 * torch.rand(25, generator=g8, device=dev, dtype=torch.float64) - 1.0
xty8 = torch.zeros(q8, device=dev, dtype=torch.float64)
for _ in range(8):
    xb = torch.randn((n8 // 8, q8), generator=g8, device=dev, dtype=torch.float64)
    yb = xb @ b8 + torch.randn(n8 // 8, generator=g8, device=dev, dtype=torch.float64)
    xtx8 += xb.t() @ xb; xty8 += xb.t() @ yb
    del xb, yb
xtx8 /= n8; xty8h = (xty8 / n8).cpu().numpy()
ctx8 = oem_amd.context(); L.check(L.lib().oemgpu_set_timing(ctx8, 1))
for _ in range(2):
    f8 = oem_amd.oem_xtx(xtx8, xty8h, penalty="lasso", nlambda=100, tol=1e-10); torch.cuda.synchronize()
    ms8 = (C.c_double * L.NTIMERS)(); L.check(L.lib().oemgpu_last_timings(ctx8, ms8))
    it = int(np.sum(f8["niter"][0]))
    print("q8192 c4-recipe: eigen+path", ms8[L.T_EIGPATH], "ms,", it, "iterations; nnz per lambda (every 10th):", [(int((np.asarray(f8["beta"][0])[:, i] != 0).sum()), int(f8["niter"][0][i])) for i in range(0, 100, 10)])
