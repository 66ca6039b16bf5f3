#!/usr/bin/env python3
"""xval.oem at p + 1 > 288 (n = 2e5, p = 400, 10 folds, 50 lambdas): the K + 1 fits as one cooperating-engine launch vs a host
thread per fold on the launch-per-iteration engines."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oem_amd
g = torch.Generator(device="cuda"); g.manual_seed(5)
n, p = 200_000, 400
xt = torch.randn((p, n), generator=g, device="cuda", dtype=torch.float64)
bb = torch.zeros(p, dtype=torch.float64, device="cuda"); bb[:20] = torch.rand(20, generator=g, device="cuda", dtype=torch.float64) - 0.5
y = (xt.t() @ bb + torch.randn(n, generator=g, device="cuda", dtype=torch.float64)).cpu().numpy()
foldid = np.random.default_rng(1).permutation(np.resize(np.arange(1, 11), n))
kw = dict(foldid=foldid, penalty=["lasso"], nlambda=50, tol=1e-9)
res = {}
for name, env in (("one coop launch", {}), ("thread per fold", {"OEM_NO_COOP": "1"})):
    for k, v in env.items(): os.environ[k] = v
    f = oem_amd.xval_oem(xt.t(), y, **kw); torch.cuda.synchronize()
    t0 = time.perf_counter(); f = oem_amd.xval_oem(xt.t(), y, **kw); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    for k in env: del os.environ[k]
    res[name] = (f, dt)
    print(f"{name}: {1e3 * dt:.2f} ms; lambda.min {f['lambda.min']:.6g}")
a, b = res["one coop launch"][0], res["thread per fold"][0]
print("max |cvm diff| rel", float(np.max(np.abs(a["cvm"][0] - b["cvm"][0]) / np.abs(b["cvm"][0]))))
