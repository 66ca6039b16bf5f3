#!/usr/bin/env python3
"""One oem() call with p >= n on the persistent cooperating-workgroup engine (n = 500, p = 2,000, 50-lambda lasso, X resident), for
rocprofv3: the whole eigenvalue + path is ONE launch of path_wcoop_kernel (plus the standardisation kernels and the buffer clears)."""
import os, sys, warnings, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, oem_amd
warnings.simplefilter("ignore")
rng = np.random.default_rng(5)
n, p = 500, 2000
x = np.asfortranarray(rng.normal(size=(n, p))); y = x[:, :10] @ rng.uniform(0.5, 1.5, 10) + rng.normal(size=n)
xd = torch.as_tensor(np.ascontiguousarray(x.T), device="cuda").t()
for _ in range(3):
    fit = oem_amd.oem(xd, y, penalty="lasso", nlambda=50, tol=1e-7)
torch.cuda.synchronize()
print("iterations", int(fit["niter"][0].sum()))
