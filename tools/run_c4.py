import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, oem_amd
rng = np.random.default_rng(123)
p, n = 4096, 65536
x = rng.normal(size=(n, p)); b = np.zeros(p); b[:25] = rng.uniform(-1, 1, 25); y = x @ b + rng.normal(size=n)
xtx, xty = x.T @ x / n, x.T @ y / n
xtxd = torch.as_tensor(xtx, device="cuda")
for _ in range(2):
    fit = oem_amd.oem_xtx(xtxd, xty, penalty="lasso", nlambda=100, tol=1e-10)
print(int(fit["niter"][0].sum()))
