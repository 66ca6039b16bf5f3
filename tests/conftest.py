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


def _reload_switches():
    """liboemgpu parses its OEM_* / OEMGPU_* environment switches ONCE (oem_amd/csrc/switches.hpp); tests that flip one tell it"""
    try:
        from oem_amd import _lib as L
    except Exception:
        return
    if L._lib is not None:
        L.reload_switches()


@pytest.fixture(autouse=True)
def _switches_follow_the_environment(monkeypatch):
    """every monkeypatch.setenv / delenv of a test is followed by oemgpu_reload_switches(), and so are the start of a test (the previous
    one's undo) and its end"""
    _reload_switches()
    setenv, delenv = monkeypatch.setenv, monkeypatch.delenv

    def setenv_and_reload(*a, **k):
        setenv(*a, **k)
        _reload_switches()

    def delenv_and_reload(*a, **k):
        delenv(*a, **k)
        _reload_switches()
    monkeypatch.setenv, monkeypatch.delenv = setenv_and_reload, delenv_and_reload
    yield
    monkeypatch.setenv, monkeypatch.delenv = setenv, delenv
