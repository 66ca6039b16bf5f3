"""Row-sharded solve with two and three processes on a GPU box (the N > 1 path of bench.py), through the C ABI."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.gpu
@pytest.mark.parametrize("nproc", [2, 3])              # 3: the row split leaves a remainder on the last rank
def test_ranks_share_one_gpu(nproc):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(29575 + nproc), str(ROOT / "tests" / "dist_gpu_worker.py")]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert "DIST_GPU_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])
