import ctypes as C, os, sys, time, warnings, numpy as np
sys.path.insert(0, "/root/repo")
import torch, oem_amd
from oem_amd import _lib as L
warnings.simplefilter("ignore")
lib = L.lib()
for (n, p, nlam, pen) in ((500, 2000, 50, "lasso"), (1000, 2048, 20, "elastic.net"), (300, 1200, 30, "lasso"), (900, 1500, 30, "mcp")):
    gw = torch.Generator(device="cuda"); gw.manual_seed(7)
    xw = torch.randn((p, n), generator=gw, device="cuda", dtype=torch.float64)
    bw = torch.zeros(p, dtype=torch.float64, device="cuda"); bw[:10] = 1.0
    yw = (xw.t() @ bw + torch.randn(n, generator=gw, device="cuda", dtype=torch.float64)).cpu().numpy()
    res = {}
    for mode in ("default(rowcoop Gram)", "OEM_WIDE=1 (wcoop)"):
        os.environ.pop("OEM_WIDE", None)
        __import__('oem_amd')._lib.reload_switches()      # (the library parses its switches once)
        if mode.startswith("OEM_WIDE"): os.environ["OEM_WIDE"] = "1"
        __import__('oem_amd')._lib.reload_switches()      # (the library parses its switches once)
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); fit = oem_amd.oem(xw.t(), yw, penalty=pen, nlambda=nlam, tol=1e-7); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        it = int(np.sum(fit["niter"][0])); res[mode] = fit
        print(n, p, pen, mode, f"{1e3*best:.1f} ms wall, {it} iterations, {1e6*best/it:.2f} us/it", flush=True)
    a, b = list(res.values())
    print("   beta diff", np.abs(a["beta"][0] - b["beta"][0]).max(), "niter diff", np.abs(a["niter"][0].astype(int) - b["niter"][0].astype(int)).max())
