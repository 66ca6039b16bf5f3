#!/usr/bin/env python3
"""Two (or more) PROCESSES on one GPU, each making cooperating-engine calls (209 <= q <= 512: one XCD per instance) in a loop: every
result must equal the process's first one; prints calls, persistent fallbacks (timeouts) and placements seen.

    python tools/two_process_coop.py [nproc] [seconds]
"""
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
if len(sys.argv) > 1 and sys.argv[1] == "--worker":
    sys.path.insert(0, str(ROOT))
    import numpy as np
    import torch
    import oem_amd as oa
    k, secs = int(sys.argv[2]), float(sys.argv[3])
    rng = np.random.default_rng(100 + k)
    q = (300, 512, 256, 430)[k % 4]
    x = rng.normal(size=(4000, q)); y = x[:, :8] @ np.ones(8) + rng.normal(size=4000)
    xd = torch.as_tensor(np.asfortranarray(x), device="cuda"); yd = torch.as_tensor(y, device="cuda")
    kw = dict(penalty=["lasso", "mcp"] if k % 2 else "lasso", nlambda=40, tol=1e-9)
    first = oa.oem(xd, yd, **kw)
    t0, calls, places, bad = time.time(), 0, {}, 0
    while time.time() - t0 < secs:
        f = oa.oem(xd, yd, **kw)
        calls += 1
        pl = oa.api.last_placement(); places[pl] = places.get(pl, 0) + 1
        bad += not all(np.array_equal(np.asarray(f["beta"][i]), np.asarray(first["beta"][i])) for i in range(len(f["beta"])))
    print(f"worker {k} (q = {q}, {kw['penalty']}): {calls} calls in {secs:.0f} s, engine {oa.last_path_engine()[0]}, persistent fallbacks {oa.last_path_engine()[1]}, "
          f"placements {places}, results that differ from the first: {bad}")
    sys.exit(1 if bad else 0)
nproc = int(sys.argv[1]) if len(sys.argv) > 1 else 2
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
ps = [subprocess.Popen([sys.executable, __file__, "--worker", str(k), str(secs)]) for k in range(nproc)]
rc = [p.wait() for p in ps]
print("exit codes", rc)
sys.exit(max(rc))
