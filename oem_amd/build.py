"""Builds liboemgpu.so (hipcc, gfx950 only) in-tree and audits the generated ISA.

    python -m oem_amd.build [--force]

The shared object lands in oem_amd/liboemgpu.so (git-ignored, but it travels with the repo
snapshot to the GPU box).  hipcc cross-compiles without a GPU.
"""
import os
import re
import subprocess
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
CSRC = HERE / "csrc"
OUT = HERE / "liboemgpu.so"
SOURCES = ["api.hip", "hoststream.hip", "gram.hip", "gram_sb.hip", "gram_wd.hip", "path_small.hip", "path_coop.hip", "path_symcoop.hip", "path_wcoop.hip", "path_wres.hip", "path_large.hip", "xval.hip", "sparse.hip", "wide.hip", "weighted.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-fvisibility=hidden", "-Wall", "-Wno-unused-function"]


def _newer(a, b):
    return (not b.exists()) or a.stat().st_mtime > b.stat().st_mtime


def _deps():
    # every header / include any source may pull in: an edited penalty_ops.hpp must rebuild path_*.hip
    return [CSRC / s for s in SOURCES] + sorted(CSRC.glob("*.hpp")) + sorted((CSRC / "gen").glob("*.inc")) + \
        [HERE.parent / "include" / "oemgpu.h"]


def audit_gram_isa(asm_text):
    """The Gram kernels own AGPRs a0..a223 through literal register names in inline asm
    (csrc/gen/acc_tiles.inc).  That is only sound if hipcc itself never touches the accumulator
    file and never spills in those kernels: check both in the emitted ISA."""
    problems = []
    found = 0
    for m in re.finditer(r"^(_ZN6oemgpu1[4567]gram_(?:tri|blk|ring|sb|wd3?)_kernel\w+):[^\n]*\n(.*?)\n\.Lfunc_end", asm_text, re.S | re.M):
        name, body = m.group(1), m.group(2)
        found += 1
        in_asm = False
        for line in body.splitlines():
            if "#ASMSTART" in line:
                in_asm = True
            elif "#ASMEND" in line:
                in_asm = False
            elif not in_asm and "v_accvgpr" in line:
                problems.append(f"{name}: compiler-emitted {line.strip()}")
            if "scratch_" in line:
                problems.append(f"{name}: scratch access {line.strip()}")
            if "flat_load" in line:
                problems.append(f"{name}: flat_load (drains the prefetch) {line.strip()}")
    if found == 0:
        problems.append("no Gram kernel found in the ISA listing (the audit pattern is stale)")
    return problems


def audit_dpp_hazards(asm_text):
    """path_small.hip issues v_fmac_f64_dpp from inline asm, for which hipcc pads no hazards: a VALU write of the DPP
    source needs two wait states before the DPP instruction reads it.  dpp_hazard_fence() ties an s_nop to the
    registers; this check proves it on the emitted ISA (every instruction is one wait state, s_nop N is N + 1)."""
    problems, nfma = [], 0
    lines = [l.strip() for l in asm_text.splitlines()]
    ins = [l for l in lines if l and not l.startswith((";", ".", "#")) and not l.endswith(":")]

    def regs(tok):
        m = re.match(r"v\[(\d+):(\d+)\]", tok)
        if m:
            return set(range(int(m.group(1)), int(m.group(2)) + 1))
        m = re.match(r"v(\d+)$", tok)
        return {int(m.group(1))} if m else set()

    for i, l in enumerate(ins):
        if not l.startswith("v_fmac_f64_dpp"):
            continue
        nfma += 1
        ops = [t.strip() for t in l.split(None, 1)[1].split(",")]
        src = regs(ops[1])
        waited, k = 0, i - 1
        while k >= 0 and waited < 2:
            p = ins[k]
            if p.startswith("s_nop"):
                waited += int(p.split()[1]) + 1
            else:
                if p.startswith("v_") and not p.startswith("v_fmac_f64_dpp"):
                    dst = regs(p.split(None, 1)[1].split(",")[0].strip())
                    if dst & src:
                        problems.append(f"DPP hazard: '{p}' {waited} wait state(s) before '{l}'")
                waited += 1
            k -= 1
    if nfma == 0:
        problems.append("no v_fmac_f64_dpp found in path_small.hip (the audit pattern is stale)")
    return problems


def audit_round_spills(asm_text, limit=16):
    """The eight-wave forms of the row-split path kernel keep 176 of their 256 VGPRs for the matrix, and hipcc's allocation of
    such a kernel is one edit away from spilling into its hot loop (docs/history.md section 3.2: config 2's SCAD round once went from
    6 to 22 ms that way, with every result right).  iterate_rows_t marks its rounds in the listing; between the marks of an
    element-wise operator without the accelerate option there may be at most `limit` scratch instructions (a handful is what
    the allocator leaves there today; a cliff is a hundred)."""
    problems, found = [], 0
    for m in re.finditer(r"^(_ZN6oemgpu\S*path_rows_kernel\w+):[^\n]*\n(.*?)\n\.Lfunc_end", asm_text, re.S | re.M):
        name, body = m.group(1), m.group(2)
        kind, count = None, 0
        for line in body.splitlines():
            b = re.search(r"; oem-round-begin (\d+)", line)
            if b:
                kind, count = int(b.group(1)), 0
                found += 1
            elif "; oem-round-end" in line:
                if kind is not None and kind < 8 and kind % 2 == 0 and count > limit:      # K_SOFT .. K_OLS, accelerate off
                    problems.append(f"{name}: {count} scratch instructions in a round of operator kind {kind // 2}")
                kind = None
            elif kind is not None and "scratch_" in line:
                count += 1
    if found == 0:
        problems.append("no round markers found in path_small.hip (the audit pattern is stale)")
    return problems


def audit_wres_isa(asm_text):
    """path_wres.hip: path_wres_kernel keeps two to eight column sets of every wave in AGPRs a0..a255 that only its inline asm names
    (as path_symcoop.hip's kernels): hipcc itself must not touch the accumulator file there, nor spill to scratch."""
    problems, found = [], 0
    for m in re.finditer(r"^(_ZN6oemgpu\S*path_wres_kernel\w+):[^\n]*\n(.*?)\n\.Lfunc_end", asm_text, re.S | re.M):
        name, body = m.group(1), m.group(2)
        found += 1
        in_asm = False
        for line in body.splitlines():
            if "#ASMSTART" in line:
                in_asm = True
            elif "#ASMEND" in line:
                in_asm = False
            elif not in_asm and ("v_accvgpr" in line or re.search(r"\ba\[?\d+", line.split(";")[0])):
                problems.append(f"{name}: compiler-emitted {line.strip()}")
            if "scratch_" in line:
                problems.append(f"{name}: scratch access {line.strip()}")
    if found < 8:
        problems.append(f"only {found} of the 8 path_wres kernels found in the ISA listing (the audit pattern is stale)")
    return problems


def audit_symcoop_isa(asm_text):
    """path_symcoop.hip keeps two tiles of every wave in AGPRs a0..a255 that only its inline asm names.  That is only sound if
    hipcc itself never uses the accumulator file in those kernels (its own values must fit the architectural VGPRs) and never
    spills to scratch: check both on the emitted ISA."""
    problems, found = [], 0
    for m in re.finditer(r"^(_ZN6oemgpu\S*path_(?:symcoop|rowcoop)_kernel\w+):[^\n]*\n(.*?)\n\.Lfunc_end", asm_text, re.S | re.M):
        name, body = m.group(1), m.group(2)
        found += 1
        in_asm = False
        for line in body.splitlines():
            if "#ASMSTART" in line:
                in_asm = True
            elif "#ASMEND" in line:
                in_asm = False
            elif not in_asm and ("v_accvgpr" in line or re.search(r"\ba\[?\d+", line.split(";")[0])):
                problems.append(f"{name}: compiler-emitted {line.strip()}")
            if "scratch_" in line:
                problems.append(f"{name}: scratch access {line.strip()}")
    if found == 0:
        problems.append("no path_symcoop kernel found in the ISA listing (the audit pattern is stale)")
    elif found < 7:
        problems.append(f"only {found} of the 7 register-resident kernels found in the ISA listing (the audit pattern is stale)")
    return problems


def build_diag():
    """liboemgpu_diag.so: the same library with -DOEM_PATH_DIAG (stamped round segments); never the product."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    out = HERE / "liboemgpu_diag.so"
    subprocess.run([hipcc, *FLAGS, "-DOEM_PATH_DIAG", "-DOEM_GRAM_DIAG", "-shared", "-o", str(out)] + [str(CSRC / s) for s in SOURCES], check=True)
    return out


def build(force=False, verbose=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    gen = CSRC / "gen" / "acc_tiles.inc"
    if force or _newer(CSRC / "gen" / "gen_acc_tiles.py", gen):
        subprocess.run([sys.executable, str(CSRC / "gen" / "gen_acc_tiles.py")], check=True, stdout=subprocess.DEVNULL)
    if not force and OUT.exists() and all(not _newer(d, OUT) for d in _deps()):
        return OUT
    objs = []
    bdir = HERE / "_build"
    bdir.mkdir(exist_ok=True)
    # the objects are independent hipcc runs: side by side
    from concurrent.futures import ThreadPoolExecutor
    jobs = []
    AUDITED = ("gram.hip", "gram_sb.hip", "gram_wd.hip", "path_small.hip", "path_coop.hip", "path_wcoop.hip", "path_wres.hip", "path_symcoop.hip")
    # the audited sources are compiled ONCE: -save-temps=obj leaves the device ISA of the very object that is linked next to it (until
    # round 5 each of the five largest translation units was compiled twice, once for the object and once more with -S for the audit:
    # the cold build's critical path)
    def isa_path(src):
        return bdir / (src + ".tmp") / (src[:-4] + "-hip-amdgcn-amd-amdhsa-gfx950.s")
    with ThreadPoolExecutor(max_workers=int(os.environ.get("OEM_BUILD_JOBS", str(min(8, os.cpu_count() or 1))))) as ex:
        for s in SOURCES:
            o = bdir / (s + ".o")
            stale = force or _newer(CSRC / s, o) or any(_newer(d, o) for d in _deps()[len(SOURCES):]) or (s in AUDITED and not isa_path(s).exists())
            if stale:
                cmd = [hipcc, *FLAGS, "-c", str(CSRC / s), "-o", str(o)]
                if s in AUDITED:
                    tmp = bdir / (s + ".tmp")
                    tmp.mkdir(exist_ok=True)
                    cmd = [hipcc, *FLAGS, "-save-temps=obj", "-c", str(CSRC / s), "-o", str(tmp / (s[:-4] + ".o"))]
                if verbose:
                    print(" ".join(cmd))
                jobs.append((s, ex.submit(subprocess.run, cmd, capture_output=s in AUDITED, text=True)))
            objs.append(str(o))
        for s, j in jobs:
            r = j.result()
            if r.returncode != 0:
                raise RuntimeError(f"hipcc failed on {s}:\n" + ((r.stderr or "")[-4000:]))
            if s in AUDITED:
                import shutil
                shutil.copyfile(bdir / (s + ".tmp") / (s[:-4] + ".o"), bdir / (s + ".o"))
                for f in (bdir / (s + ".tmp")).iterdir():             # (keep the ISA listing only: the other temporaries are 100 MB)
                    if f != isa_path(s):
                        f.unlink()

        class _Listing:                                  # (the shape the audits below were written against)
            def __init__(self, path):
                self.stdout = Path(path).read_text()

            def result(self):
                return self
        listing = {src: _Listing(isa_path(src)) for src in AUDITED}
        for src in ("gram.hip", "gram_sb.hip", "gram_wd.hip"):
            problems = audit_gram_isa(listing[src].result().stdout)
            if problems:
                raise RuntimeError(src + " ISA audit failed:\n  " + "\n  ".join(problems[:20]))
        for src in ("path_small.hip", "path_coop.hip", "path_wcoop.hip", "path_wres.hip"):
            problems = audit_dpp_hazards(listing[src].result().stdout)
            if problems:
                raise RuntimeError(src + " ISA audit failed:\n  " + "\n  ".join(problems[:20]))
        problems = audit_wres_isa(listing["path_wres.hip"].result().stdout)
        if problems:
            raise RuntimeError("path_wres.hip (path_wres_kernel) ISA audit failed:\n  " + "\n  ".join(problems[:20]))
        problems = audit_symcoop_isa(listing["path_symcoop.hip"].result().stdout)
        if problems:
            raise RuntimeError("path_symcoop.hip ISA audit failed:\n  " + "\n  ".join(problems[:20]))
        problems = audit_round_spills(listing["path_small.hip"].result().stdout)
        if problems:
            raise RuntimeError("path_small.hip round-spill audit failed:\n  " + "\n  ".join(problems[:20]))
    subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(OUT), *objs], check=True)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    if "--diag" in sys.argv:
        print(build_diag())
