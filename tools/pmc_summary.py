#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter CSVs for one kernel.

    python tools/pmc_summary.py <kernel-substring> <out.json> <counter_collection.csv> [...]

Per counter: mean value per dispatch of the kernels whose name contains the substring.  For FETCH_SIZE / WRITE_SIZE
(KB) it also writes hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 -- the gfx950 correction for 16-byte-per-lane
streaming reads prescribed by /opt/skills/guides/MI355X_MICROARCH.md (HBM section).
"""
import csv
import json
import sys
from collections import defaultdict

sub, out = sys.argv[1], sys.argv[2]
acc = defaultdict(list)
names = set()
for path in sys.argv[3:]:
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if sub in row["Kernel_Name"]:
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
                names.add(row["Kernel_Name"].split("(")[0][:120])
res = {"kernel_filter": sub, "kernels": sorted(names),
       "counters": {k: {"mean_per_dispatch": sum(v) / len(v), "dispatches": len(v)} for k, v in acc.items()}}
c = res["counters"]
if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
    res["fetch_bytes_corrected"] = 2.0 * c["FETCH_SIZE"]["mean_per_dispatch"] * 1024.0
    res["write_bytes"] = c["WRITE_SIZE"]["mean_per_dispatch"] * 1024.0
    res["hbm_bytes_per_dispatch"] = res["fetch_bytes_corrected"] + res["write_bytes"]
    res["correction"] = "FETCH_SIZE x2 (gfx950, 16 B/lane streaming reads), WRITE_SIZE as read; KB -> bytes"
if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
    # rocprofv3's own derived metric (rocprofv3 -L): MfmaUtil = sum(SQ_VALU_MFMA_BUSY_CYCLES) / (max(GRBM_GUI_ACTIVE) * SIMD_NUM).  The CSV holds
    # GRBM_GUI_ACTIVE summed over the 8 XCDs of an MI355X (each counts the dispatch's cycles), so max = sum / 8; SIMD_NUM = 256 CUs x 4.
    cyc = c["GRBM_GUI_ACTIVE"]["mean_per_dispatch"] / 8.0
    res["gpu_cycles_per_dispatch"] = cyc
    res["mfma_util"] = c["SQ_VALU_MFMA_BUSY_CYCLES"]["mean_per_dispatch"] / (cyc * 1024.0)
    res["mfma_util_note"] = "fraction of SIMD-cycles in which the matrix pipe is busy, at the clock the chip actually held (counter-based; the bench line's roofline.frac divides by the 2.4 GHz peak)"
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
