import os, sys, time, warnings, numpy as np
sys.path.insert(0, "/root/repo")
import torch, oem_amd
warnings.simplefilter("ignore")
rng = np.random.default_rng(5)
n, p = 500, 20000
x = np.asfortranarray(rng.normal(size=(n, p))); y = x[:, :10] @ rng.uniform(0.5, 1.5, 10) + rng.normal(size=n)
xd = torch.as_tensor(np.ascontiguousarray(x.T), device="cuda").t()
kw = dict(penalty=["lasso", "mcp", "grp.lasso"], groups=np.arange(p) // 10 + 1, nlambda=20, tol=1e-7)
for mode in ("two parts", "one call"):
    os.environ.pop("OEM_NO_PENALTY_SPLIT", None)
    if mode == "one call": os.environ["OEM_NO_PENALTY_SPLIT"] = "1"
    best = 1e9
    for _ in range(2):
        t0 = time.perf_counter(); f = oem_amd.oem(xd, y, **kw); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    print(f"500 x 20,000, lasso + mcp + grp.lasso, 20 lambdas [{mode}]: {1e3 * best:.1f} ms, iterations {[int(np.sum(v)) for v in f['niter']]}", flush=True)
