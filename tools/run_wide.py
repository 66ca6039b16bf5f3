import os, sys, warnings, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, oem_amd
warnings.simplefilter("ignore")
os.environ["OEM_NO_WRES"] = "1"      # the launch-per-iteration wide engine (wide_cols_kernel) is what this profile is of: since round 4 the library
                                     # itself runs this size from registers (path_wres_kernel; tools/wres_time.py)
rng = np.random.default_rng(5)
n, p = 500, 20000
x = np.asfortranarray(rng.normal(size=(n, p))); y = x[:, :10] @ rng.uniform(0.5, 1.5, 10) + rng.normal(size=n)
xd = torch.as_tensor(np.ascontiguousarray(x.T), device="cuda").t()
for _ in range(2):
    fit = oem_amd.oem(xd, y, penalty="lasso", nlambda=50, tol=1e-7)
print(int(fit["niter"][0].sum()))
