#!/usr/bin/env python3
"""1024 < q <= 4096 with groups that are NOT runs of neighbouring coordinates -- of about 25 members, and 60 scattered groups of q / 60 (more than the
32 coordinates of an owner's slice: norms summed over several owners) -- on the register-resident engine, and the same calls / the second attempt after a
timed-out persistent launch (OEM_NO_SYMCOOP=1) on the launch-per-iteration engines: us per iteration.  python tools/scattered_groups_time.py [q ...]"""
import os, sys, ctypes as C
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
import oem_amd
from oem_amd import _lib as L
for p in [int(a) for a in sys.argv[1:]] or [1536, 2048, 3000, 4096]:
    g = torch.Generator(device="cuda"); g.manual_seed(p)
    n = 2 * p
    x = torch.randn((n, p), generator=g, device="cuda", dtype=torch.float64)
    b = torch.zeros(p, dtype=torch.float64, device="cuda"); b[:25] = 1.0
    y = x @ b + torch.randn(n, generator=g, device="cuda", dtype=torch.float64)
    xtx = (x.t() @ x) / n; xty = ((x.t() @ y) / n).cpu().numpy()
    del x
    lib = L.lib(); ctx = oem_amd.context()
    L.check(lib.oemgpu_set_timing(ctx, 1))
    scattered = np.arange(p) % 60 + 1
    small = np.arange(p) % (p // 25) + 1                 # <= 32 members each: reordered into runs, the register-resident engine takes them
    for label, env, kw in (("grp.lasso, scattered groups of 25", {}, dict(penalty=["grp.lasso"], groups=small)),
                           ("grp.lasso, 60 scattered groups", {}, dict(penalty=["grp.lasso"], groups=scattered)),
                           ("grp.lasso, 60 scattered groups, launches (OEM_NO_SYMCOOP=1)", {"OEM_NO_SYMCOOP": "1"}, dict(penalty=["grp.lasso"], groups=scattered)),
                           ("lasso, launches (OEM_NO_SYMCOOP=1)", {"OEM_NO_SYMCOOP": "1"}, dict(penalty=["lasso"])),
                           ("lasso, row-streaming launches (OEM_NO_SYMCOOP=1 OEM_NO_SYM=1)", {"OEM_NO_SYMCOOP": "1", "OEM_NO_SYM": "1"}, dict(penalty=["lasso"])),
                           ("grp.lasso scattered, row-streaming launches (OEM_NO_SYMCOOP=1 OEM_NO_SYM=1)", {"OEM_NO_SYMCOOP": "1", "OEM_NO_SYM": "1"}, dict(penalty=["grp.lasso"], groups=scattered))):
        for k, v in env.items(): os.environ[k] = v
        best = 1e9
        for _ in range(2):
            fit = oem_amd.oem_xtx(xtx, xty, nlambda=20, tol=1e-8, lambda_min_ratio=0.01, **kw); torch.cuda.synchronize()
            ms = (C.c_double * L.NTIMERS)(); L.check(lib.oemgpu_last_timings(ctx, ms)); best = min(best, ms[L.T_EIGPATH])
        for k in env: del os.environ[k]
        st, cp = C.c_int32(-1), C.c_int32(-1); lib.oemgpu_last_eigen_info(ctx, C.byref(st), C.byref(cp))
        it = int(np.sum(fit["niter"][0])) + int(st.value)
        print(f"q={p} {label}: engine {oem_amd.last_path_engine()[0]}, eigen+path {best:.1f} ms, {it} products: {1e3 * best / it:.1f} us each", flush=True)
    del xtx; torch.cuda.empty_cache()
