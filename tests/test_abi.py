"""The C ABI library loads on a CPU-only box, exports every symbol include/oemgpu.h declares, and refuses to
compute without a GPU (no CPU fallback)."""
import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def _declared():
    h = (ROOT / "include" / "oemgpu.h").read_text()
    h = re.sub(r"/\*.*?\*/", "", h, flags=re.S)
    names = set(re.findall(r"\b(oemgpu_[a-z_0-9]+)\s*\(", h))
    names.discard("oemgpu_moments_len")          # static inline
    names.discard("oemgpu_sums_len")             # static inline
    return sorted(names)


def test_header_symbols_are_exported():
    import oem_amd
    L = oem_amd.lib()
    decl = _declared()
    assert len(decl) >= 17
    for name in decl:
        assert hasattr(L, name), name
    assert sorted(oem_amd.EXPORTS) == decl
    assert L.oemgpu_version().decode().startswith("oemgpu")


def test_no_internal_cxx_symbols_leak():
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", str(ROOT / "oem_amd" / "liboemgpu.so")], capture_output=True, text=True).stdout
    syms = [l.split()[-1] for l in out.splitlines() if " T " in l]
    assert syms and all(s.startswith("oemgpu_") for s in syms), [s for s in syms if not s.startswith("oemgpu_")][:5]


def test_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import oem_amd
    L = oem_amd.lib()
    assert L.oemgpu_device_count() == 0
    assert not L.oemgpu_create(0, None)
    assert b"no HIP device" in L.oemgpu_last_error()
    x = np.asfortranarray(np.random.default_rng(0).normal(size=(50, 4)))
    with pytest.raises(oem_amd.OemgpuError) as e:
        oem_amd.oem(x, x[:, 0])
    assert e.value.code == -2
    with pytest.raises(oem_amd.OemgpuError):
        oem_amd.oem_xtx(x.T @ x, x.T @ x[:, 0])
    with pytest.raises(oem_amd.OemgpuError):
        oem_amd.big_oem(x, x[:, 0], penalty="lasso")


def test_product_never_imports_the_oracle():
    for f in list((ROOT / "oem_amd").rglob("*.py")) + list((ROOT / "oem_amd" / "csrc").rglob("*.*")):
        txt = f.read_text(errors="ignore")
        assert "oracle" not in txt.replace("# oracle", ""), f
