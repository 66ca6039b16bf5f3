"""A second PROCESS that holds most of the GPU's CUs for a few seconds (tests/test_gpu_host.py: the persistent engines' fallback
under real contention).  usage: hold_cus_worker.py <blocks> <milliseconds> <flag file>"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import oem_amd
from oem_amd import _lib as L
blocks, ms, flag = int(sys.argv[1]), float(sys.argv[2]), Path(sys.argv[3])
lib = L.lib(); ctx = oem_amd.context()
L.check(lib.oemgpu_selftest_hold_cus(ctx, blocks, ms))
flag.write_text("holding")                       # the kernel is enqueued; it starts within microseconds
L.check(lib.oemgpu_synchronize(ctx))
flag.write_text("released")
print("HOLD_DONE", flush=True)
