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


def test_opts_struct_matches_the_header():
    """the ctypes mirror of oemgpu_opts has the header's size and field offsets (checked with the C compiler)"""
    import subprocess
    import tempfile
    from oem_amd import _lib as L
    fields = [f[0] for f in L.OemgpuOpts._fields_]
    prog = '#include <stddef.h>\n#include <stdio.h>\n#include "oemgpu.h"\nint main(void){printf("%zu", sizeof(oemgpu_opts));\n' + \
        "".join(f'printf(" %zu", offsetof(oemgpu_opts, {f}));\n' for f in fields) + "return 0;}\n"
    with tempfile.TemporaryDirectory() as td:
        src = Path(td) / "t.c"
        src.write_text(prog)
        subprocess.run(["gcc", "-I", str(ROOT / "include"), str(src), "-o", str(Path(td) / "t")], check=True)
        out = subprocess.run([str(Path(td) / "t")], capture_output=True, text=True, check=True).stdout.split()
    assert int(out[0]) == C.sizeof(L.OemgpuOpts)
    for f, off in zip(fields, out[1:]):
        assert int(off) == getattr(L.OemgpuOpts, f).offset, f


def test_row_split_is_the_references_block_rule():
    """floor(n / G) rows per device, the remainder on the last (ref src/oem_dense.h:328,343) -- pure host arithmetic"""
    import oem_amd
    from oem_amd.distributed import row_partition
    L = oem_amd.lib()
    for n in (0, 1, 7, 8, 1000003, 10 ** 8):
        for G in (1, 2, 3, 8):
            got = []
            for g in range(G):
                a, b = C.c_int64(), C.c_int64()
                L.oemgpu_row_split(n, G, g, C.byref(a), C.byref(b))
                got.append((a.value, b.value))
            assert got == row_partition(n, G)
            assert got[0][0] == 0 and got[-1][1] == n and all(got[i][1] == got[i + 1][0] for i in range(G - 1))
            assert all(b - a == n // G for a, b in got[:-1])
