"""The R binding (r/oem_shim.c, r/oem_shim_big.cpp) read by a compiler and RUN -- without R.

tests/r_api_stub/ holds hand-written stand-ins for exactly the R C-API names the shim uses (Rinternals.h ...), a toy runtime behind
them that collects on every allocation (R under gctorture(TRUE)), a recording fake of liboemgpu, and a driver that makes the `.Call`s
of R/oem.R:556-575, R/oem_xtx.R:389-406, R/big_oem.R:447-491, R/oem_xval.R:497-521, R/oem.R:534-553.

What this shows: the shim parses under -Wall -Wextra -Werror (C99 / C++), its calls into include/oemgpu.h type-check, the 19 / 16 / 22
SEXP arguments reach the oemgpu_opts fields they belong to without a copy of x, the returned list has the names, storage modes and
dimensions of ref src/oem_dense.cpp:280-307 ("ols" as a plain vector), PROTECT / UNPROTECT balance on every exit path, errors carry
oemgpu_last_error() and an interrupt is re-raised with Rf_onintr.  What it does not show: anything about R itself or about numbers --
it pins no parity and is not an oracle/_ref build (the stand-ins exist only under tests/)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB = os.path.join(ROOT, "tests", "r_api_stub")
FLAGS = ["-Wall", "-Wextra", "-Werror", "-I", STUB, "-I", os.path.join(ROOT, "include")]


def _build(tmp, shim_c, name):
    objs = []
    for src, cc in ((shim_c, ["gcc", "-std=c99"]), (os.path.join(STUB, "r_stub_runtime.c"), ["gcc", "-std=c99"]),
                    (os.path.join(STUB, "fake_oemgpu.c"), ["gcc", "-std=c99"]), (os.path.join(STUB, "shim_driver.c"), ["gcc", "-std=c99"]),
                    (os.path.join(ROOT, "r", "oem_shim_big.cpp"), ["g++"]), (os.path.join(STUB, "big_matrix_maker.cpp"), ["g++"])):
        obj = os.path.join(tmp, f"{name}_{os.path.basename(src)}.o")
        subprocess.run(cc + ["-g", "-O0"] + FLAGS + ["-c", src, "-o", obj], check=True, capture_output=True, text=True)
        objs.append(obj)
    exe = os.path.join(tmp, name)
    subprocess.run(["g++", "-o", exe] + objs, check=True, capture_output=True, text=True)
    return exe


def test_the_shim_parses():
    """gcc -std=c99 -Wall -Wextra -Werror -fsyntax-only on the C shim, g++ on the big.matrix translation unit"""
    for cc, src in ((["gcc", "-std=c99"], "oem_shim.c"), (["g++"], "oem_shim_big.cpp")):
        r = subprocess.run(cc + FLAGS + ["-fsyntax-only", os.path.join(ROOT, "r", src)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr


def test_the_shim_marshals_and_balances_its_protects(tmp_path):
    exe = _build(str(tmp_path), os.path.join(ROOT, "r", "oem_shim.c"), "drv")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    assert "checks passed" in r.stdout and int(r.stdout.split()[2]) > 500


@pytest.mark.parametrize("mutation", ["a list that is not protected", "one UNPROTECT too few"])
def test_the_stand_in_runtime_notices_protect_bugs(tmp_path, mutation):
    """the test of the test: pack() with one PROTECT removed dies in the collector emulation; with an unbalanced count the driver's
    depth check fails"""
    src = open(os.path.join(ROOT, "r", "oem_shim.c")).read()
    good = "SEXP lb = PROTECT(Rf_allocVector(VECSXP, o->npen)), ll"
    assert good in src and "UNPROTECT(6);" in src
    if mutation.startswith("a list"):
        src = src.replace(good, "SEXP lb = Rf_allocVector(VECSXP, o->npen), ll").replace("UNPROTECT(6);", "UNPROTECT(5);")
    else:
        src = src.replace("UNPROTECT(6);", "UNPROTECT(5);")
    mut = tmp_path / "oem_shim_mutated.c"
    mut.write_text(src)
    exe = _build(str(tmp_path), str(mut), "drvmut")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert r.returncode != 0
    assert ("a PROTECT is missing" in r.stderr) if mutation.startswith("a list") else ("stub_protect_depth() == 0" in r.stderr)
