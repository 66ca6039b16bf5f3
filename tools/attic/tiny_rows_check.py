#!/usr/bin/env python3
"""Moment kernels on a handful of rows (1 .. 129) with garbage behind the last row, against numpy: the shapes a 1-row shard of
big.oem or a tiny cross-validation fold produce."""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from oem_amd import _lib as L  # noqa: E402
from oem_amd.distributed import HipBackend  # noqa: E402

be = HipBackend(0)
rng = np.random.default_rng(0)
bad = 0
for p in (5, 65, 100, 110, 111, 120, 300, 520):
    for n in (1, 2, 3, 5, 7, 8, 9, 17, 63, 64, 65, 127, 129):
        for pad in (0, 2):
            ld = (n + 1) // 2 * 2 + pad
            print("case p", p, "n", n, "ld", ld, flush=True)
            x = rng.normal(size=(n, p)); y = rng.normal(size=n)
            buf = torch.full((p, ld), 7.5, device="cuda", dtype=torch.float64)       # garbage beyond row n
            buf[:, :n] = torch.as_tensor(np.ascontiguousarray(x.T), device="cuda")
            yd = torch.full((ld + 8,), -3.25, device="cuda", dtype=torch.float64); yd[:n] = torch.as_tensor(y, device="cuda")
            with be.section():
                mom = be.new_buffer(L.moments_len(p))
                be.moments(buf[:, :n].t(), n, ld, p, yd, None, mom)
            torch.cuda.synchronize()
            M = mom.cpu().numpy().reshape(p + 2, p + 2)
            z = np.column_stack([x, y, np.ones(n)])
            ref = z.T @ z
            err = np.abs(M - ref).max() / max(1.0, np.abs(ref).max())
            if not (err < 1e-12):
                bad += 1; print("MISMATCH p", p, "n", n, "ld", ld, "err", err, flush=True)
print("bad", bad)
