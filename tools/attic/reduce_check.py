import sys, time, ctypes as C, numpy as np, torch
sys.path.insert(0, '/root/repo')
import oem_amd
from oem_amd import _lib as L
lib = L.lib()
ctx = oem_amd.context()
for (n, p) in [(200000, 512), (1000000, 512), (300000, 256), (50000, 300), (70000, 1000), (1000000, 100)]:
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    xt = torch.randn((p, n), generator=g, device="cuda", dtype=torch.float64)
    y = torch.randn(n, generator=g, device="cuda", dtype=torch.float64)
    mom = torch.zeros((p + 2) * (p + 2), device="cuda", dtype=torch.float64)
    L.check(lib.oemgpu_set_timing(ctx, 1))
    for _ in range(3):
        L.check(lib.oemgpu_moments_dev(ctx, xt.data_ptr(), n, n, p, y.data_ptr(), None, mom.data_ptr()))
    ms = (C.c_double * 8)(); L.check(lib.oemgpu_last_timings(ctx, ms))
    L.check(lib.oemgpu_set_timing(ctx, 0))
    M = mom.reshape(p + 2, p + 2).cpu().numpy()
    ref = (xt[:, :].double() @ xt.t()).cpu().numpy()
    err = np.abs(np.tril(M[:p, :p]) - np.tril(ref)).max() / np.abs(ref).max()
    xy = (xt @ y).cpu().numpy()
    err2 = np.abs(M[p, :p] - xy).max() / np.abs(xy).max()
    a = mom.clone()
    L.check(lib.oemgpu_moments_dev(ctx, xt.data_ptr(), n, n, p, y.data_ptr(), None, mom.data_ptr()))
    torch.cuda.synchronize()
    print(f"n={n} p={p}: moments {ms[1]:.3f} ms, gram kernel {ms[4]:.3f} ms, reduce ~{ms[1]-ms[4]:.3f} ms; rel err {err:.1e} {err2:.1e}; bitwise repeat {bool(torch.equal(a, mom))}")
    del xt, y
