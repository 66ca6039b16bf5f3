import os, sys, warnings, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, oem_amd
warnings.simplefilter("ignore")
rng = np.random.default_rng(5)
n, p = 500, 20000                    # the library's own choice at this size: path_wres_kernel<8>, ONE launch (tools/run_wide.py: the launches it replaced)
x = np.asfortranarray(rng.normal(size=(n, p))); y = x[:, :10] @ rng.uniform(0.5, 1.5, 10) + rng.normal(size=n)
xd = torch.as_tensor(np.ascontiguousarray(x.T), device="cuda").t()
for _ in range(2):
    fit = oem_amd.oem(xd, y, penalty="lasso", nlambda=50, tol=1e-7)
print(int(fit["niter"][0].sum()), oem_amd.last_path_engine())
