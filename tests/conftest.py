import json
import os
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


# the oracle's two products per iteration on a few threads: bit-identical results (oracle/oem_oracle.c: orc_threads), and most of the
# GPU suite's wall clock is that loop
os.environ.setdefault("ORC_THREADS", str(max(1, min(16, os.cpu_count() or 1))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def doc_kats():
    return json.loads((ROOT / "tests" / "golden" / "doc_kats.json").read_text())
