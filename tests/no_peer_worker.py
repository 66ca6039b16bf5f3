"""Worker of tests/test_gpu_host.py::test_handover_staged_through_the_host (a child process: OEMGPU_NO_PEER is read once).
With OEMGPU_NO_PEER=1 every cross-device hand-over of the in-library multi-GPU path (hoststream.hip: hand_over) takes the route
for devices that cannot access each other -- device -> pinned host buffer -> device -- also between two contexts of ONE device,
so the fallback runs on a one-GPU box.  Same numbers as the one-device call."""
import os
import sys

import numpy as np

os.environ["OEMGPU_NO_PEER"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oem_amd  # noqa: E402
from oem_amd import _lib as L  # noqa: E402

rng = np.random.default_rng(5)
ok = True
n, p = 30_011, 60
x = np.asfortranarray(rng.normal(size=(n, p)) * 2.0 + 0.3)
y = x[:, :4] @ np.array([1.0, -1.0, 0.5, 0.25]) + rng.normal(size=n) + 0.7
kw = dict(penalty=["lasso", "mcp"], nlambda=12, tol=1e-10)
one = oem_amd.oem(x, y, **kw)
ok &= L.host_stats()["host_staged_handovers"] == 0
for devices in ([0, 0], [0, 0, 0]):
    many = oem_amd.oem(x, y, devices=devices, **kw)
    st = L.host_stats()
    ok &= st["host_staged_handovers"] == len(devices) - 1 and st["devices"] == len(devices)
    for k in range(2):
        ok &= float(np.abs(many["beta"][k] - one["beta"][k]).max()) < 1e-10 and bool(np.array_equal(many["niter"][k], one["niter"][k]))
# a shifted redo (sums summed and broadcast as well) and the penalty split (moments broadcast to every device)
xs = x + 500.0
one = oem_amd.oem(xs, y, **kw)
many = oem_amd.oem(xs, y, devices=[0, 0], **kw)
ok &= L.host_stats()["host_staged_handovers"] >= 3
for k in range(2):
    ok &= float(np.abs(many["beta"][k] - one["beta"][k]).max()) < 1e-9
p3, n3 = 320, 3000
x3 = np.asfortranarray(rng.normal(size=(n3, p3))); y3 = x3[:, :3] @ np.array([1.0, -1.0, 0.5]) + rng.normal(size=n3)
kw3 = dict(penalty=["lasso", "mcp", "scad"], nlambda=5, tol=1e-9, maxit=1000)
one = oem_amd.oem(x3, y3, **kw3)
many = oem_amd.oem(x3, y3, devices=[0, 0], **kw3)
for k in range(3):
    ok &= float(np.abs(many["beta"][k] - one["beta"][k]).max()) < 1e-10
# xval.oem over devices: fold moments to the first device, fold coefficients back
K = 5
fid = rng.permutation(np.resize(np.arange(1, K + 1), n))
kx = dict(penalty=["lasso"], nlambda=8, tol=1e-9, maxit=2000, foldid=fid)
one = oem_amd.xval_oem(x, y, **kx)
many = oem_amd.xval_oem(x, y, devices=[0, 0], **kx)
ok &= float(np.abs(many["beta"][0] - one["beta"][0]).max()) < 1e-10 and bool(np.allclose(many["cvm"][0], one["cvm"][0], rtol=1e-10))
print("NO_PEER_OK" if ok else "NO_PEER_MISMATCH", flush=True)
